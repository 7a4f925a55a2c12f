// Probe (development aid): does a stream created with hipExtStreamCreateWithCUMask keep a kernel's workgroups on the CUs of its mask on
// this box, and how are the mask's bits numbered?  Launches 512 workgroups of 512 threads that note (XCC id, SE id, CU id) of the CU they
// ran on, on the null stream and on streams with a mask of every other bit / the low half of the bits.
// build + run:  hipcc --offload-arch=gfx950 -O2 -o /tmp/cu_mask_probe profiles/cu_mask_probe.hip && /tmp/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>

__global__ void k(unsigned *out)
{
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 15u;      // HW_REG_XCC_ID
        const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));               // HW_REG_HW_ID
        out[blockIdx.x] = (xcc << 16) | ((hw >> 8) & 0xFu) | (((hw >> 13) & 0x7u) << 4) | (((hw >> 12) & 1u) << 7);      // cu_id | se_id << 4 | sh_id << 7
    }
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(20);       // (long enough that the workgroups have to be spread over the CUs)
}

static int distinct(hipStream_t st, unsigned *dev, const char *name)
{
    const int n = 512;
    hipMemsetAsync(dev, 0xFF, n * 4, st);
    hipLaunchKernelGGL(k, dim3(n), dim3(512), 0, st, dev);
    hipStreamSynchronize(st);
    std::vector<unsigned> h(n);
    hipMemcpy(h.data(), dev, n * 4, hipMemcpyDeviceToHost);
    std::set<unsigned> cus(h.begin(), h.end());
    std::set<unsigned> xccs;
    for (unsigned v : h) xccs.insert(v >> 16);
    printf("%-36s %3zu distinct CUs on %zu XCCs\n", name, cus.size(), xccs.size());
    return (int)cus.size();
}

int main()
{
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    printf("multiProcessorCount %d\n", pr.multiProcessorCount);
    unsigned *dev;
    hipMalloc(&dev, 4096 * 4);
    distinct(nullptr, dev, "null stream");
    const int words = (pr.multiProcessorCount + 31) / 32;
    std::vector<unsigned> m(words);
    hipStream_t s1, s2;
    for (int w = 0; w < words; ++w) m[w] = 0x55555555u;
    hipError_t e = hipExtStreamCreateWithCUMask(&s1, words, m.data());
    printf("hipExtStreamCreateWithCUMask(every other bit): %s\n", hipGetErrorString(e));
    if (e == hipSuccess) distinct(s1, dev, "mask 0x55555555...");
    for (int w = 0; w < words; ++w) m[w] = w < words / 2 ? 0xFFFFFFFFu : 0u;
    e = hipExtStreamCreateWithCUMask(&s2, words, m.data());
    if (e == hipSuccess) distinct(s2, dev, "mask low half of the bits");
    return 0;
}
