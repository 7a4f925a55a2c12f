// Development aid: what do scattered dirty lines cost at the end of a kernel?  N lanes each touch one
// 8-byte word in its own 128 B line of a 2 GB buffer: read-only vs read-modify-write vs write-only.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void touch(double *buf, long long stride_words, int n, int mode, double *sink)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // scatter: a multiplicative hash keeps neighbours far apart
    const long long line = ((long long)i * 2654435761ll) % (1ll << 24);
    double *p = buf + line * stride_words;
    if (mode == 0) { if (*p == 12345.0) *sink = 1.0; }
    else if (mode == 1) *p = *p + 1.0;
    else *p = 1.0;
}
int main()
{
    double *buf, *sink;
    const long long words = (1ll << 24) * 16;           // 2^24 lines of 128 B = 2 GB
    hipMalloc(&buf, words * 8); hipMalloc(&sink, 8);
    hipMemset(buf, 0, words * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const char *names[3] = {"read", "read-modify-write", "write"};
    for (int n : {20000, 118000, 500000})
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 8; ++rep) {
                hipEventRecord(a, 0);
                hipLaunchKernelGGL(touch, dim3((n + 255) / 256), dim3(256), 0, 0, buf, 16ll, n, mode, sink);
                hipEventRecord(b, 0); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (rep > 1 && ms < best) best = ms;
            }
            printf("n=%7d lines  %-18s %.2f us\n", n, names[mode], best * 1e3);
        }
    return 0;
}
