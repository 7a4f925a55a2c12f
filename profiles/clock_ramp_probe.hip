// Probe (development aid): at what shader clock does a SHORT kernel run?  k_run's fixed part (fire found, window loaded, written back, result
// block) is ~14 k shader clocks by the kernel's own counter, yet a 1-update launch takes ~13 us: if the chip takes a launch at a low clock and
// ramps, clocks and microseconds part ways at the head of every launch.  Every wave times a dependent integer chain with both counters: the
// shader clock (s_memtime) and the constant 100 MHz clock (s_memrealtime); lane 0 of workgroup 0 reports.  Launched after an idle gap (the
// bench's situation: a reset, a synchronize, then the timed launch) and back to back, on one CU and on all 256.
// build + run:  hipcc --offload-arch=gfx950 -O2 -w -o /tmp/clock_ramp_probe profiles/clock_ramp_probe.hip && /tmp/clock_ramp_probe
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <cstdio>

__global__ __launch_bounds__(1024) void chain(int n, unsigned *out, unsigned long long *rep)
{
    unsigned v = threadIdx.x;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) v = v * 3u + 1u;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
    if (threadIdx.x == 0 && blockIdx.x == 0) { rep[0] = t1 - t0; rep[1] = r1 - r0; }
}

int main()
{
    unsigned *out; unsigned long long *rep, h[2];
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&rep, 16);
    for (int grid : {1, 256})
        for (int gap_us : {0, 200, 5000})
            for (int n : {300, 1000, 3000, 10000, 30000, 100000}) {
                double ghz = 0, us = 0;
                const int reps = 8;
                for (int i = 0; i < reps + 2; ++i) {
                    hipDeviceSynchronize();
                    if (gap_us) usleep(gap_us);
                    hipLaunchKernelGGL(chain, dim3(grid), dim3(1024), 0, 0, n, out, rep);
                    hipMemcpy(h, rep, 16, hipMemcpyDeviceToHost);
                    if (i >= 2) { ghz += (double)h[0] / ((double)h[1] * 10.0); us += h[1] * 0.01; }
                }
                printf("grid %3d x 1024, idle gap %4d us, chain %6d: %.1f us in the kernel, shader clock %.2f GHz\n", grid, gap_us, n, us / reps, ghz / reps);
            }
    return 0;
}
