// Development aid: shader clock during short kernels.  A dependent chain of N v_add (4 clocks each on a
// wave64) timed with HIP events gives the clock; s_memtime deltas give the tick rate of the phase profile.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain(int n, unsigned *out, unsigned long long *ticks)
{
    unsigned v = threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) v = v * 3u + 1u;      // v_mad_u32_u24 / v_mul_lo + add: dependent
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) *ticks = t1 - t0;
}
int main()
{
    unsigned *out; unsigned long long *ticks, h;
    hipMalloc(&out, 256); hipMalloc(&ticks, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep)
        for (int n : {1000, 10000, 100000, 1000000}) {
            hipEventRecord(a, 0);
            hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, n, out, ticks);
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
            printf("n=%7d  %.1f us  memtime ticks=%llu  -> %.1f ticks/us, %.2f ticks/iter\n", n, ms * 1e3, h, h / (ms * 1e3), (double)h / n);
        }
    return 0;
}
