// Probe (development aid): does `global_load_lds_dword` (the LDS-DMA form of a global load: no VGPR destination) put lane i's dword at
// LDS[M0 + 4 i] on gfx950 - also for M0 beyond 64 KB - and nothing anywhere else?  The window phase of k_run uses it for table-line touches
// whose results nobody reads (sf_win_kernels.h): the answer decides where the dump area may sit.
// build + run:  hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_dma_probe profiles/lds_dma_probe.hip && /tmp/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k(const unsigned *src, unsigned *out, unsigned dump_off, int n_dw)
{
    extern __shared__ unsigned dyn[];
    for (int i = threadIdx.x; i < n_dw; i += blockDim.x) dyn[i] = 0xA5000000u | (unsigned)i;
    __syncthreads();
    if (threadIdx.x < 64 && (threadIdx.x & 1)) {      // odd lanes only: inactive lanes must not write
        unsigned m0s;
        const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned *)dyn + dump_off;
        const unsigned *q = src + threadIdx.x * 16;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, off\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                     : "=&s"(m0s) : "s"(__builtin_amdgcn_readfirstlane(base)), "v"(q) : "memory");
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_dw; i += blockDim.x) out[i] = dyn[i];
}

int main()
{
    const int n_dw = 150 * 1024 / 4;
    unsigned *src, *out;
    hipMalloc(&src, 64 * 16 * 4); hipMalloc(&out, n_dw * 4);
    std::vector<unsigned> h(64 * 16), o(n_dw);
    for (int i = 0; i < 64 * 16; ++i) h[i] = 0xC0DE0000u | (unsigned)i;
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, n_dw * 4);
    int bad_total = 0;
    for (unsigned off : {0u, 1024u, 65536u + 512u, 100u * 1024u, 140u * 1024u}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), n_dw * 4, 0, src, out, off, n_dw);
        hipMemcpy(o.data(), out, n_dw * 4, hipMemcpyDeviceToHost);
        int hit = 0, bad = 0;
        for (int i = 0; i < n_dw; ++i) {
            const int lane = i - (int)(off / 4);
            const bool target = lane >= 0 && lane < 64 && (lane & 1);
            const unsigned want = target ? (0xC0DE0000u | (unsigned)(lane * 16)) : (0xA5000000u | (unsigned)i);
            if (o[i] == want) hit += target; else { if (bad < 4) printf("  off %u: dword %d = %08x, expected %08x\n", off, i, o[i], want); ++bad; }
        }
        printf("dump at LDS byte %6u: %d of 32 lanes' dwords where expected, %d dwords wrong\n", off, hit, bad);
        bad_total += bad;
    }
    printf(bad_total ? "PROBE FAILED\n" : "probe ok: lane i's dword lands at M0 + 4 i, inactive lanes write nothing, M0 reaches beyond 64 KB\n");
    return bad_total != 0;
}
