// How long after a kernel's end does the host learn of it?  One 14-us-ish kernel per iteration, then one of:
//  (a) hipStreamSynchronize   (b) hipStreamWriteValue32 into pinned memory + poll   (c) a one-lane marker kernel writing the
//  pinned word + poll   (d) hipEventRecord + hipEventSynchronize   (e) the work kernel's own last store into pinned memory + poll
// hipcc --offload-arch=gfx950 -O2 latency.hip -o latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include <immintrin.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void work(const uint4 *s, uint4 *d, size_t n, volatile unsigned *flag, unsigned v)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
    if (flag && blockIdx.x == 0 && threadIdx.x == 0) *flag = v;   // (only meaningful as "a store from a kernel"; not ordered behind the others)
}
__global__ void marker(volatile unsigned *flag, unsigned v) { *flag = v; }
int main()
{
    const size_t S = 3840ull * 2160 * 4, n = S / 16;
    uint4 *s, *d;
    CK(hipMalloc(&s, S)); CK(hipMalloc(&d, S)); CK(hipMemset(s, 1, S));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned *flag; CK(hipHostMalloc((void **)&flag, 64, hipHostMallocDefault));
    unsigned *dflag; CK(hipHostGetDevicePointer((void **)&dflag, flag, 0));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    *flag = 0;
    unsigned seq = 0;
    auto poll = [&](unsigned v) { while (*(volatile unsigned *)flag != v) _mm_pause(); };
    const char *names[] = {"kernel + hipStreamSynchronize", "kernel + hipStreamWriteValue32 + poll", "kernel + marker kernel + poll",
                           "kernel + hipEventRecord + hipEventSynchronize", "kernel (own store) + poll", "kernel + hipEventRecord + hipEventQuery spin",
                           "kernel + hipStreamQuery spin"};
    for (int mode = 0; mode < 7; mode++) {
        std::vector<double> ts;
        for (int it = 0; it < 600; it++) {
            const double a = now();
            seq++;
            hipLaunchKernelGGL(work, dim3(2048), dim3(256), 0, st, s, d, n, mode == 4 ? dflag : nullptr, seq);
            switch (mode) {
            case 0: CK(hipStreamSynchronize(st)); break;
            case 1: { hipError_t e = hipStreamWriteValue32(st, dflag, seq, 0); if (e != hipSuccess) { printf("%-48s unsupported (%s)\n", names[mode], hipGetErrorString(e)); goto next; } poll(seq); break; }
            case 2: hipLaunchKernelGGL(marker, dim3(1), dim3(1), 0, st, dflag, seq); poll(seq); break;
            case 3: CK(hipEventRecord(ev, st)); CK(hipEventSynchronize(ev)); break;
            case 4: poll(seq); break;
            case 5: CK(hipEventRecord(ev, st)); while (hipEventQuery(ev) == hipErrorNotReady) _mm_pause(); break;
            case 6: while (hipStreamQuery(st) == hipErrorNotReady) _mm_pause(); break;
            }
            if (it >= 100) ts.push_back(now() - a);
            if (mode == 4) CK(hipStreamSynchronize(st));
        }
        std::sort(ts.begin(), ts.end());
        printf("%-48s median %6.1f us  min %6.1f  p90 %6.1f\n", names[mode], 1e6 * ts[ts.size() / 2], 1e6 * ts[0], 1e6 * ts[ts.size() * 9 / 10]);
    next:;
    }
    return 0;
}
