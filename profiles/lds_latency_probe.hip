// Probe (development aid): what the dependent chain of a window step is made of, in shader clocks on one CU with a 16-wave workgroup
// resident (the window phase's situation): a dependent LDS read, a returning LDS atomic, a workgroup barrier (all 16 waves arriving
// together; one wave arriving late by a fixed amount of work), a dependent VALU instruction.  These are the terms of
// roofline.chain_bound_clocks (bench.py, NOTEBOOK.md 5.9).
// build + run:  hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_latency_probe profiles/lds_latency_probe.hip && /tmp/lds_latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(1024) void k(unsigned long long *out, int iters)
{
    __shared__ unsigned chain[4096];
    __shared__ unsigned ctr;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < 4096; i += 1024) chain[i] = (unsigned)((i * 1103515245u + 12345u) & 4095u);
    if (tid == 0) ctr = 0;
    __syncthreads();
    unsigned long long t0, t1;
    // (1) dependent LDS reads, wave 0 only (the others wait at the barrier below)
    unsigned v = (unsigned)lane;
    if (wave == 0) {
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) v = chain[v];
        t1 = __builtin_readcyclecounter();
        if (lane == 0) out[0] = (t1 - t0) / iters;
    }
    __syncthreads();
    // (2) returning LDS atomic of one lane, dependent
    if (wave == 0) {
        unsigned a = 1;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) { if (lane == 0) a = atomicAdd(&ctr, a & 1u) | 1u; a = __builtin_amdgcn_readfirstlane(a); }
        t1 = __builtin_readcyclecounter();
        if (lane == 0) out[1] = (t1 - t0) / iters;
        v += a;
    }
    __syncthreads();
    // (3) workgroup barrier, all 16 waves together
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); }
    t1 = __builtin_readcyclecounter();
    if (tid == 0) out[2] = (t1 - t0) / iters;
    // (4) dependent VALU instructions in one wave alone on its SIMD
    __syncthreads();
    if (wave == 0) {
        unsigned x = v;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) x = (x ^ (x >> 3)) + 0x9E3779B9u;      // 3 dependent VALU instructions
        }
        t1 = __builtin_readcyclecounter();
        if (lane == 0) out[3] = (t1 - t0) * 100 / (iters * 48ull);             // clocks x 100 per instruction
        v += x;
    }
    __syncthreads();
    // (5) the same VALU chain in FOUR waves of one SIMD at once (waves 0, 4, 8, 12)
    if ((wave & 3) == 0) {
        unsigned x = v;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) x = (x ^ (x >> 3)) + 0x9E3779B9u;
        }
        t1 = __builtin_readcyclecounter();
        if (tid == 0) out[4] = (t1 - t0) * 100 / (iters * 48ull);
        v += x;
    }
    __syncthreads();
    // (6) ... and in waves 0, 1, 2, 3 at once: whichever of (5) / (6) is the slower one says how a workgroup's waves are dealt to the SIMDs
    if (wave < 4) {
        unsigned x = v;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) x = (x ^ (x >> 3)) + 0x9E3779B9u;
        }
        t1 = __builtin_readcyclecounter();
        if (tid == 0) out[5] = (t1 - t0) * 100 / (iters * 48ull);
        v += x;
    }
    __syncthreads();
    // (7) all 16 waves at once
    {
        unsigned x = v;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) x = (x ^ (x >> 3)) + 0x9E3779B9u;
        }
        t1 = __builtin_readcyclecounter();
        if (tid == 0) out[6] = (t1 - t0) * 100 / (iters * 48ull);
        v += x;
    }
    if (v == 0xFFFFFFFFu) out[7] = v;
}

int main()
{
    unsigned long long *d, h[8] = {};
    hipMalloc(&d, 64);
    hipMemset(d, 0, 64);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(1024), 0, 0, d, 2000);
    hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("dependent LDS read                          %4llu clocks\n", h[0]);
    printf("returning LDS atomic of one lane            %4llu clocks\n", h[1]);
    printf("workgroup barrier, 16 waves, LDS fences     %4llu clocks\n", h[2]);
    printf("dependent VALU instruction, one wave / SIMD %4.2f clocks\n", h[3] / 100.0);
    printf("... waves 0, 4, 8, 12 at once               %4.2f clocks per instruction of each wave\n", h[4] / 100.0);
    printf("... waves 0, 1, 2, 3 at once                %4.2f\n", h[5] / 100.0);
    printf("... all 16 waves at once                    %4.2f\n", h[6] / 100.0);
    return 0;
}
