// Litmus test: does a 16-byte store followed (program order, same wave, no s_waitcnt between) by a 1-byte store from ANOTHER
// LANE of the same wave to an address inside those 16 bytes always leave the byte store's value in memory?
// k_run's walk relies on it only together with an explicit s_waitcnt vmcnt(0); this probe asks whether the wait is needed.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/sop profiles/store_order_probe.hip && /tmp/sop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void probe(uint8_t *buf, int rounds, unsigned long long *bad, int wait)
{
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    // every wave owns 64 vectors of 16 bytes, spread over many cache lines (scattered like the cell planes)
    uint8_t *mine = buf + ((size_t)wave * 64 + lane) * 4096;
    unsigned long long wrong = 0;
    for (int r = 0; r < rounds; ++r) {
        const uint32_t tag = (uint32_t)(r * 2654435761u) | 0x01010101u;
        *reinterpret_cast<uint4 *>(mine) = make_uint4(tag, tag, tag, tag);              // vector store by the owner lane
        // byte store into the NEIGHBOUR lane's vector (lane ^ 1), later in program order
        uint8_t *other = buf + ((size_t)wave * 64 + (lane ^ 1)) * 4096 + (r & 15);
        if (wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *other = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const uint8_t got = __builtin_nontemporal_load(mine + (r & 15));                 // my vector, byte written by lane ^ 1
        if (got != 0) wrong++;
        __builtin_amdgcn_wave_barrier();
    }
    if (wrong) atomicAdd(bad, wrong);
}

int main()
{
    const int blocks = 1024, threads = 256, rounds = 20000;
    const size_t bytes = (size_t)blocks * threads * 4096;
    uint8_t *buf; unsigned long long *bad, h = 0;
    hipMalloc(&buf, bytes); hipMalloc(&bad, 8);
    for (int wait = 0; wait < 2; ++wait) {
        hipMemset(bad, 0, 8);
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, buf, rounds, bad, wait);
        hipDeviceSynchronize();
        hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
        printf("wait=%d: %llu wrong of %llu\n", wait, h, (unsigned long long)blocks * threads * rounds);
    }
    return 0;
}
