// Probe (development aid): can the HOST store straight into DEVICE memory (through the PCIe BAR) on this box, and what does a doorbell cost
// that way?  The closed loop (sf_loop_step) rings its doorbell in HOST memory today: the GPU's relay workgroup polls it with PCIe reads
// (~2 us a look) and fetches the step's points with another round trip.  A doorbell + points the host WRITES into device memory would be
// one posted write, polled by every workgroup at device-memory cost.
// For each kind of allocation: is the pointer host-accessible (a forked child tries the store, so that a fault is survivable), does a kernel
// see the host's store, and the round trip host store -> kernel sees it -> kernel's store to pinned host memory -> host sees it.
// build + run:  hipcc --offload-arch=gfx950 -O2 -o /tmp/bar_write_probe profiles/bar_write_probe.hip && /tmp/bar_write_probe
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstring>

__global__ void k_read(const volatile unsigned *p, unsigned *out) { out[0] = p[0]; }

// one workgroup: waits for *bell == n (agent-scope load that skips the caches), answers by writing n to *ack (host memory), n = 1 .. rounds
__global__ void k_pingpong(const unsigned *bell, unsigned *ack, int rounds, unsigned long long *spins_out)
{
    unsigned long long spins = 0;
    for (int n = 1; n <= rounds; ++n) {
        while (__hip_atomic_load(bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (unsigned)n) ++spins;
        __hip_atomic_store(ack, (unsigned)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    *spins_out = spins;
}

static bool host_can_store(void *p)
{
    const pid_t c = fork();
    if (c == 0) { *(volatile unsigned *)p = 0xBEEFu; _exit(0); }
    int st = 0;
    waitpid(c, &st, 0);
    return WIFEXITED(st) && WEXITSTATUS(st) == 0;
}

int main()
{
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    printf("device: %s, canMapHostMemory %d, managedMemory %d, pageableMemoryAccess %d\n", pr.name, pr.canMapHostMemory, pr.managedMemory, pr.pageableMemoryAccess);
    unsigned *out, *ack_h, *ack_d;
    unsigned long long *spins;
    hipMalloc(&out, 64);
    hipMalloc(&spins, 8);
    hipHostMalloc(&ack_h, 64, hipHostMallocMapped | hipHostMallocCoherent);
    hipHostGetDevicePointer((void **)&ack_d, ack_h, 0);
    struct { const char *name; unsigned flags; int kind; } kinds[] = {
        {"hipMalloc", 0, 0}, {"hipExtMallocWithFlags(Finegrained)", hipDeviceMallocFinegrained, 1}, {"hipExtMallocWithFlags(Uncached)", hipDeviceMallocUncached, 1},
        {"hipMallocManaged", 0, 2}, {"hipHostMalloc(Mapped|Coherent) [today's doorbell]", 0, 3}};
    for (auto &kd : kinds) {
        unsigned *p = nullptr, *pd = nullptr;
        hipError_t e = hipSuccess;
        if (kd.kind == 0) e = hipMalloc(&p, 4096);
        else if (kd.kind == 1) e = hipExtMallocWithFlags((void **)&p, 4096, kd.flags);
        else if (kd.kind == 2) e = hipMallocManaged(&p, 4096);
        else { e = hipHostMalloc(&p, 4096, hipHostMallocMapped | hipHostMallocCoherent); }
        if (e != hipSuccess) { printf("%-52s allocation failed: %s\n", kd.name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        pd = p;
        if (kd.kind == 3) hipHostGetDevicePointer((void **)&pd, p, 0);
        hipMemset(pd, 0, 4096);
        hipDeviceSynchronize();
        const bool ok = host_can_store(p);
        printf("%-52s host store: %s", kd.name, ok ? "ok" : "FAULT");
        if (!ok) { printf("\n"); continue; }
        *(volatile unsigned *)p = 1234u;
        __sync_synchronize();
        hipLaunchKernelGGL(k_read, dim3(1), dim3(1), 0, 0, pd, out);
        unsigned seen = 0;
        hipMemcpy(&seen, out, 4, hipMemcpyDeviceToHost);
        printf("; kernel sees the host's store: %s", seen == 1234u ? "yes" : "NO");
        if (seen != 1234u) { printf(" (%u)\n", seen); continue; }
        // ping-pong
        const int rounds = 2000;
        *(volatile unsigned *)p = 0; ack_h[0] = 0;
        __sync_synchronize();
        hipLaunchKernelGGL(k_pingpong, dim3(1), dim3(64), 0, 0, pd, ack_d, rounds, spins);
        usleep(20000);
        const auto t0 = std::chrono::steady_clock::now();
        bool stuck = false;
        for (int n = 1; n <= rounds && !stuck; ++n) {
            *(volatile unsigned *)p = (unsigned)n;
            __sync_synchronize();
            const auto w0 = std::chrono::steady_clock::now();
            while (*(volatile unsigned *)ack_h != (unsigned)n)
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() > 2.0) { stuck = true; break; }
        }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
        if (stuck) { printf("; ping-pong STUCK (the kernel never saw a later store)\n"); *(volatile unsigned *)p = 0; hipDeviceReset(); return 1; }
        hipDeviceSynchronize();
        unsigned long long sp = 0;
        hipMemcpy(&sp, spins, 8, hipMemcpyDeviceToHost);
        printf("; round trip %.2f us (%.0f polls per round)\n", us, (double)sp / rounds);
    }
    return 0;
}
