// membw: read-bandwidth ceiling of this MI355X for the access patterns the kernels use
// (development tool).  Reads `bytes` once per launch with 16-byte loads; variants differ in
// loads in flight per lane, workgroups, and temporal hint.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int U, bool NT>
__global__ __launch_bounds__(256) void rd(const u32x4 *p, size_t n16, uint32_t *out)
{
    // each workgroup owns a contiguous span; lanes stride by 256 within it, U loads in flight
    size_t per = (n16 + gridDim.x - 1) / gridDim.x;
    size_t b = per * blockIdx.x, e = b + per < n16 ? b + per : n16;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = b + threadIdx.x; i < e; i += 256 * U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            size_t j = i + (size_t)u * 256;
            if (j < e) v[u] = NT ? __builtin_nontemporal_load(p + j) : p[j]; else v[u] = (u32x4){0,0,0,0};
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *out = 1;
}

// the box-downsample pattern: workgroup (segment, row group) reads ROWS row pieces of 4 KB at
// stride `pitch` (one 16-byte chunk per lane per row), all loads in flight, then exits
template <int ROWS, bool NT>
__global__ __launch_bounds__(256) void rd2d(const uint8_t *p, int pitch, int segs, uint32_t *out)
{
    const int seg = blockIdx.x % segs;
    const size_t rg = blockIdx.x / segs;
    const uint8_t *base = p + rg * ROWS * (size_t)pitch + (size_t)seg * 4096 + threadIdx.x * 16;
    u32x4 v[ROWS];
#pragma unroll
    for (int u = 0; u < ROWS; u++) {
        const u32x4 *q = (const u32x4 *)(base + (size_t)u * pitch);
        v[u] = NT ? __builtin_nontemporal_load(q) : *q;
    }
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < ROWS; u++) acc += v[u];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *out = 1;
}

template <int ROWS, bool NT>
static void run2d(const char *name, const uint8_t *d, size_t bytes, int pitch, uint32_t *out)
{
    const int segs = pitch / 4096;                 // whole 4 KB segments per row
    const size_t rows = bytes / pitch / ROWS * ROWS;
    const int blocks = (int)(rows / ROWS) * segs;
    const double moved = (double)blocks * ROWS * 4096;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((rd2d<ROWS, NT>), dim3(blocks), dim3(256), 0, 0, d, pitch, segs, out);
    CK(hipDeviceSynchronize());
    float tot = 0; const int it = 10;
    for (int i = 0; i < it; i++) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL((rd2d<ROWS, NT>), dim3(blocks), dim3(256), 0, 0, d, pitch, segs, out); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms;
    }
    printf("%-30s pitch=%6d blocks=%6d  avg %.3f ms  %.0f GB/s\n", name, pitch, blocks, tot / it, moved / (tot / it * 1e-3) / 1e9);
}

template <int U, bool NT>
static void run(const char *name, const u32x4 *d, size_t bytes, int blocks, uint32_t *out)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((rd<U, NT>), dim3(blocks), dim3(256), 0, 0, d, bytes / 16, out);
    CK(hipDeviceSynchronize());
    float best = 1e30f, tot = 0; const int it = 10;
    for (int i = 0; i < it; i++) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL((rd<U, NT>), dim3(blocks), dim3(256), 0, 0, d, bytes / 16, out); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms; if (ms < best) best = ms;
    }
    printf("%-22s blocks=%6d  avg %.3f ms  %.0f GB/s   best %.0f GB/s\n", name, blocks, tot / it, bytes / (tot / it * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e9);
}

int main(int argc, char **argv)
{
    size_t bytes = (size_t)(argc > 1 ? atof(argv[1]) : 2.0) * (1ull << 30);
    u32x4 *d; uint32_t *out;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&out, 4)); CK(hipMemset(d, 1, bytes));
    for (int pitch : {15360, 16384, 12288, 30720}) {
        run2d<8, false>("2d rows=8", (const uint8_t *)d, bytes, pitch, out);
        run2d<8, true>("2d rows=8 nt", (const uint8_t *)d, bytes, pitch, out);
        run2d<16, true>("2d rows=16 nt", (const uint8_t *)d, bytes, pitch, out);
    }
    for (int blocks : {4096, 65536}) {
        run<1, false>("U=1", d, bytes, blocks, out);
        run<4, false>("U=4", d, bytes, blocks, out);
        run<8, false>("U=8", d, bytes, blocks, out);
        run<8, true>("U=8 nt", d, bytes, blocks, out);
    }
    return 0;
}
