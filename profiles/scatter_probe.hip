// Development aid: how many scattered (one cache line each) loads per clock ONE workgroup on ONE CU gets through -
// the budget of an environment-resident kernel (k_run / k_front), whose environment is served by a single CU.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/scatter_probe profiles/scatter_probe.hip && /tmp/scatter_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int MODE, int BYTES, int INFLIGHT>
__global__ __launch_bounds__(1024) void probe(const uint8_t *buf, size_t span, int iters, unsigned long long *out, unsigned long long *clk)
{
    unsigned long long acc = 0;
    uint32_t h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        unsigned long long v[INFLIGHT];
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) {
            h = h * 1664525u + 1013904223u;
            const size_t off = ((size_t)(h >> 4) % (span / 128)) * 128 + ((h & 15u) * 8u);
            if (BYTES == 8) {
                const unsigned long long *p = reinterpret_cast<const unsigned long long *>(buf + off);
                v[j] = MODE == 0 ? *p : __builtin_nontemporal_load(p);
            } else if (BYTES == 4) {
                const uint32_t *p = reinterpret_cast<const uint32_t *>(buf + off);
                v[j] = MODE == 0 ? *p : __builtin_nontemporal_load(p);
            } else {
                v[j] = MODE == 0 ? buf[off] : __builtin_nontemporal_load(buf + off);
            }
        }
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) acc += v[j];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

// scattered stores (one line each); MODE 0 plain, 1 followed by a load round trip every iteration (what a step of a resident kernel does)
template <int BYTES, int INFLIGHT, int MIX>
__global__ __launch_bounds__(1024) void probe_st(uint8_t *buf, size_t span, int iters, unsigned long long *out, unsigned long long *clk)
{
    unsigned long long acc = 0;
    uint32_t h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) {
            h = h * 1664525u + 1013904223u;
            const size_t off = ((size_t)(h >> 4) % (span / 128)) * 128 + ((h & 15u) * 8u);
            if (BYTES == 8) *reinterpret_cast<unsigned long long *>(buf + off) = h;
            else buf[off] = (uint8_t)h;
        }
        if (MIX) {
            h = h * 1664525u + 1013904223u;
            const size_t off = ((size_t)(h >> 4) % (span / 128)) * 128;
            acc += buf[off];
            asm volatile("" : "+v"(acc));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int BYTES, int INFLIGHT, int MIX>
void run_st(const char *name, uint8_t *buf, size_t span, int blocks, int threads, unsigned long long *out, unsigned long long *clk)
{
    const int iters = 64;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((probe_st<BYTES, INFLIGHT, MIX>), dim3(blocks), dim3(threads), 0, 0, buf, span, iters, out, clk);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> c(blocks);
    hipMemcpy(c.data(), clk, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : c) mean += (double)v; mean /= blocks;
    const double stores = (double)threads * iters * INFLIGHT;
    printf("%-34s span %6zu MB  blocks %4d x %4d thr: %8.0f clocks, %6.2f clocks per wave-store, %7.3f lines/clock/CU, per iteration %6.0f clocks\n",
           name, span >> 20, blocks, threads, mean, mean / (stores / 64), stores / mean, mean / iters);
}

template <int MODE, int BYTES, int INFLIGHT>
void run(const char *name, const uint8_t *buf, size_t span, int blocks, int threads, unsigned long long *out, unsigned long long *clk)
{
    const int iters = 64;
    hipLaunchKernelGGL((probe<MODE, BYTES, INFLIGHT>), dim3(blocks), dim3(threads), 0, 0, buf, span, iters, out, clk);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((probe<MODE, BYTES, INFLIGHT>), dim3(blocks), dim3(threads), 0, 0, buf, span, iters, out, clk);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(blocks);
    hipMemcpy(c.data(), clk, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : c) mean += (double)v; mean /= blocks;
    const double loads = (double)threads * iters * INFLIGHT;
    printf("%-34s span %6zu MB  blocks %4d x %4d thr: %8.0f clocks, %6.2f clocks per wave-load, %7.3f lines/clock/CU, round trip per batch %6.0f clocks\n",
           name, span >> 20, blocks, threads, mean, mean / (loads / 64), loads / mean, mean / iters);
}

int main()
{
    const size_t big = 1ull << 30;
    uint8_t *buf; unsigned long long *out, *clk;
    hipMalloc(&buf, big); hipMemset(buf, 1, big);
    hipMalloc(&out, 1024 * 1024 * 8); hipMalloc(&clk, 4096 * 8);
    for (size_t span : {(size_t)1 << 20, (size_t)64 << 20, big}) {
        for (int blocks : {1, 256}) {
            run<0, 8, 1>("plain 8 B, 1 in flight", buf, span, blocks, 1024, out, clk);
            run<0, 8, 4>("plain 8 B, 4 in flight", buf, span, blocks, 1024, out, clk);
            run<0, 8, 8>("plain 8 B, 8 in flight", buf, span, blocks, 1024, out, clk);
            run<1, 8, 8>("nontemporal 8 B, 8 in flight", buf, span, blocks, 1024, out, clk);
            run<0, 4, 8>("plain 4 B, 8 in flight", buf, span, blocks, 1024, out, clk);
            run<0, 1, 8>("plain 1 B, 8 in flight", buf, span, blocks, 1024, out, clk);
            run<0, 8, 8>("plain 8 B, 8 in flight, 256 thr", buf, span, blocks, 256, out, clk);
            run_st<1, 4, 0>("store 1 B x 4", buf, span, blocks, 1024, out, clk);
            run_st<8, 4, 0>("store 8 B x 4", buf, span, blocks, 1024, out, clk);
            run_st<1, 4, 1>("store 1 B x 4 then a load", buf, span, blocks, 1024, out, clk);
            run_st<1, 1, 1>("store 1 B x 1 then a load", buf, span, blocks, 1024, out, clk);
        }
    }
    return 0;
}
