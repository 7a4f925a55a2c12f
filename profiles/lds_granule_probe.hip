// Development aid: how much dynamic LDS does a launch that asks for N bytes really get?  (gfx950, ROCm 7.2)
// Every thread of one workgroup writes a tag to the word at `probe` bytes and reads it back; out-of-range LDS stores are dropped, loads give 0.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int probe_words, unsigned *out)
{
    extern __shared__ unsigned lds[];
    if (threadIdx.x == 0) lds[probe_words] = 0xABCD0000u + probe_words;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = lds[probe_words];
}
int main()
{
    unsigned *out;
    hipMalloc(&out, 4);
    const int asks[] = {20416, 20480, 20512, 21024, 21504, 22528, 30000, 40960, 65536};
    for (int threads : {128, 192, 1024})
        for (int ask : asks) {
            // highest word that can be written
            int lo = 0, hi = ask / 4 + 4096;
            while (lo + 1 < hi) {
                int mid = (lo + hi) / 2;
                hipMemset(out, 0, 4);
                if (ask > 65536 - 1) hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, ask);
                hipLaunchKernelGGL(k, dim3(1), dim3(threads), ask, 0, mid, out);
                unsigned v = 0;
                hipMemcpy(&v, out, 4, hipMemcpyDeviceToHost);
                if (v == 0xABCD0000u + (unsigned)mid) lo = mid; else hi = mid;
            }
            printf("threads %4d asked %6d bytes -> writable up to byte %6d\n", threads, ask, (lo + 1) * 4);
        }
    return 0;
}
