// Probe (development aid): what does a launch of k_run's SHAPE cost before and after the work - 256 workgroups of 1024 threads, ~132 KB of
// dynamic LDS each, a ~1 KB argument block by value - with a body that does next to nothing?  The resident launch of the driver's window
// (20 updates, ~47 us) carries ~16 us that do not scale with the updates; the window phase's own prologue + epilogue are ~14 k clocks (~6.5 us)
// of them by the kernel's own clock.  The rest is what this probe measures: dispatch, wave launch, the argument block, the end-of-kernel
// release.  Event-timed per launch (what sf_step_timed reports) and, under rocprofv3 --kernel-trace, the kernel's own duration.
// build + run:  hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_floor_probe profiles/launch_floor_probe.hip && /tmp/launch_floor_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

struct Args { unsigned long long w[120]; unsigned *out; int n; };        // 976 bytes, like StepArgs

template <int TOUCH>
__global__ __launch_bounds__(1024) void k_shape(Args a)
{
    extern __shared__ unsigned lds[];
    // TOUCH 0: one dword per workgroup; 1: every thread loads a dword it depends on and stores one (a round trip to memory and back, the least
    // a launch that hands state over does); 2: as 1, twice in a row (state -> cells)
    if (TOUCH == 0) { if (threadIdx.x == 0) a.out[blockIdx.x] = (unsigned)a.w[blockIdx.x % 120] + (unsigned)a.n; return; }
    unsigned v = a.out[blockIdx.x * 1024 + threadIdx.x];
    if (TOUCH == 2) v = a.out[(v & 0xFFFFu) * 4 + (threadIdx.x & 3)];
    lds[threadIdx.x] = v;
    __syncthreads();
    a.out[blockIdx.x * 1024 + threadIdx.x] = lds[threadIdx.x ^ 1] + (unsigned)a.n;
}

template <typename F>
static void measure(const char *what, F launch)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ms;
    for (int i = 0; i < 60; ++i) {
        hipEventRecord(e0, 0);
        launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float t = 0; hipEventElapsedTime(&t, e0, e1);
        if (i >= 10) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    printf("%-58s event-timed: median %.2f us, min %.2f us\n", what, ms[ms.size() / 2] * 1e3, ms[0] * 1e3);
}

int main()
{
    unsigned *out = nullptr;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMemset(out, 0, 256 * 1024 * 4);
    Args a = {};
    a.out = out; a.n = 1;
    const int big = 132 * 1024, small = 8 * 1024;
    hipFuncSetAttribute((const void *)k_shape<0>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    hipFuncSetAttribute((const void *)k_shape<1>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    hipFuncSetAttribute((const void *)k_shape<2>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    measure("empty: 256 x 1024 threads, 132 KB LDS", [&] { hipLaunchKernelGGL(k_shape<0>, dim3(256), dim3(1024), big, 0, a); });
    measure("empty: 256 x 1024 threads,   8 KB LDS", [&] { hipLaunchKernelGGL(k_shape<0>, dim3(256), dim3(1024), small, 0, a); });
    measure("empty: 256 x  512 threads, 132 KB LDS", [&] { hipLaunchKernelGGL(k_shape<0>, dim3(256), dim3(512), big, 0, a); });
    measure("empty: 256 x   64 threads,   8 KB LDS", [&] { hipLaunchKernelGGL(k_shape<0>, dim3(256), dim3(64), small, 0, a); });
    measure("empty:   1 x   64 threads,   8 KB LDS", [&] { hipLaunchKernelGGL(k_shape<0>, dim3(1), dim3(64), small, 0, a); });
    measure("one round trip: 256 x 1024 threads, 132 KB LDS", [&] { hipLaunchKernelGGL(k_shape<1>, dim3(256), dim3(1024), big, 0, a); });
    measure("two round trips in a row: 256 x 1024, 132 KB LDS", [&] { hipLaunchKernelGGL(k_shape<2>, dim3(256), dim3(1024), big, 0, a); });
    measure("no kernel between the events", [&] {});
    return 0;
}
