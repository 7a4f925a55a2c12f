// Development aid: dependent-load latency of one wave on MI355X, in shader clocks (s_memtime), by working-set size.
// Calibrates the "memory round trip" figures of NOTEBOOK.md 5.x.   hipcc --offload-arch=gfx950 -O3 -o /tmp/latency_probe profiles/latency_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

__device__ __forceinline__ uint32_t vload(const uint32_t *p)          // a plain VECTOR load (a uniform address would go to the scalar cache)
{
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ uint32_t vload_sc1(const uint32_t *p)      // agent scope: skips the L1
{
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// warm = laps over the first `n_warm` hops before the timed `n` hops (which start at hop `skip` of the chain)
__global__ void chase(const uint32_t *next, uint32_t start, int n_warm, int n, unsigned long long *out, int sc1)
{
    uint32_t p = start;
    for (int i = 0; i < n_warm; ++i) p = vload(next + p);
    if (n_warm) p = start;
    unsigned long long t0 = __builtin_readcyclecounter();
    if (sc1) for (int i = 0; i < n; ++i) p = vload_sc1(next + p);
    else for (int i = 0; i < n; ++i) p = vload(next + p);
    unsigned long long t1 = __builtin_readcyclecounter();
    out[0] = t1 - t0;
    out[1] = p;
}

// a store to the line, then a load of it by the same lane (what a step does to the rows it reads again one step later)
__global__ void store_then_load(uint32_t *buf, int n_lines, int reps, unsigned long long *out)
{
    unsigned long long acc = 0;
    uint32_t v = 0;
    for (int r = 0; r < reps; ++r) {
        uint32_t *p = buf + (size_t)((r * 7919) % n_lines) * 32 + threadIdx.x % 4;
        *p = r + v;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long t0 = __builtin_readcyclecounter();
        v = vload(p);
        acc += __builtin_readcyclecounter() - t0;
    }
    if (threadIdx.x == 0) { out[0] = acc; out[1] = v; }
}

int main()
{
    unsigned long long *out;
    hipMalloc(&out, 16);
    std::mt19937 rng(1);
    struct Case { size_t bytes; int warm_all; const char *what; };
    const Case cases[] = {{size_t(8) << 10, 1, "8 KB, walked before (L1)"}, {size_t(1) << 20, 1, "1 MB, walked before (L2)"},
                          {size_t(64) << 20, 1, "64 MB, walked before (past the 4 MB L2: memory-side cache)"},
                          {size_t(2) << 30, 0, "2 GB, never touched (HBM + address translation)"}};
    for (const Case &c : cases) {
        const size_t stride = 32;                  // one 128-byte line per hop
        const size_t n = c.bytes / 4 / stride;
        std::vector<uint32_t> perm(n);
        std::iota(perm.begin(), perm.end(), 0u);
        std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<uint32_t> h(c.bytes / 4, 0);
        for (size_t i = 0; i < n; ++i) h[perm[i] * stride] = perm[(i + 1) % n] * stride;
        uint32_t *d;
        hipMalloc(&d, c.bytes);
        hipMemcpy(d, h.data(), c.bytes, hipMemcpyHostToDevice);
        const int hops = (int)std::min<size_t>(n, 2048);
        for (int sc1 = 0; sc1 < 2; ++sc1) {
            // (the second pass over the never-touched set starts 2048 hops further along the chain: again lines nobody has touched)
            chase<<<1, 1>>>(d, c.warm_all ? perm[0] * stride : perm[sc1 * 2048] * stride, c.warm_all ? (int)n : 0, hops, out, sc1);
            unsigned long long r[2];
            hipMemcpy(r, out, 16, hipMemcpyDeviceToHost);
            printf("%-66s %s %7.1f clocks per dependent load\n", c.what, sc1 ? "sc1  " : "plain", (double)r[0] / hops);
        }
        hipFree(d);
    }
    uint32_t *buf;
    hipMalloc(&buf, 4 << 20);
    hipMemset(buf, 0, 4 << 20);
    store_then_load<<<1, 64>>>(buf, (4 << 20) / 128, 2000, out);
    unsigned long long r[2];
    hipMemcpy(r, out, 16, hipMemcpyDeviceToHost);
    printf("store, wait, load of the same line by the same wave: %7.1f clocks\n", (double)r[0] / 2000);
    return 0;
}
