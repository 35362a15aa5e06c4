// applyPalette's inner loop, variants timed side by side at 4K with a 256-entry palette (what bounds apply_palette_kernel?).
//   A  the library's loop: dot4 + lshl_add + min per entry and pixel, 4 px per lane, 8 entries per trip
//   B  the same with v_min3_u32 (two entries per min)
//   C  B with 8 px per lane
//   D  dot4 only (rate probe), E  lshl_add + min only (rate probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstdlib>
struct Args { const uint32_t *src; uint32_t *out; int n; int npad; const uint32_t *comp, *konst; };

template <int V, int PX>
__global__ __launch_bounds__(256) void k(Args a)
{
    const int i0 = (blockIdx.x * 256 + threadIdx.x) * PX;
    if (i0 >= a.n) return;
    uint32_t c[PX], best[PX];
#pragma unroll
    for (int e = 0; e < PX; e++) { c[e] = a.src[i0 + e] & 0x00ffffffu; best[e] = 0xffffffffu; }
    const uint32_t *__restrict__ comp = a.comp;
    const uint32_t *__restrict__ konst = a.konst;
    for (int j0 = 0; j0 < a.npad; j0 += 8) {
        uint32_t p[8], kk[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { p[j] = comp[j0 + j]; kk[j] = konst[j0 + j]; }
        if (V == 0) {
#pragma unroll
            for (int j = 0; j < 8; j++)
#pragma unroll
                for (int e = 0; e < PX; e++) best[e] = min(best[e], (__builtin_amdgcn_udot4(c[e], p[j], 0u, false) << 9) + kk[j]);
        } else if (V == 1) {
#pragma unroll
            for (int j = 0; j < 8; j += 2)
#pragma unroll
                for (int e = 0; e < PX; e++) {
                    const uint32_t k0 = (__builtin_amdgcn_udot4(c[e], p[j], 0u, false) << 9) + kk[j];
                    const uint32_t k1 = (__builtin_amdgcn_udot4(c[e], p[j + 1], 0u, false) << 9) + kk[j + 1];
                    uint32_t r;
                    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(best[e]), "v"(k0), "v"(k1));
                    best[e] = r;
                }
        } else if (V == 2) {
#pragma unroll
            for (int j = 0; j < 8; j++)
#pragma unroll
                for (int e = 0; e < PX; e++) best[e] += __builtin_amdgcn_udot4(c[e], p[j], 0u, false);
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++)
#pragma unroll
                for (int e = 0; e < PX; e++) best[e] = min(best[e], (c[e] << 9) + kk[j]), c[e] ^= p[j];
        }
    }
#pragma unroll
    for (int e = 0; e < PX; e++) a.out[i0 + e] = best[e];
}

template <int V, int PX>
float run(Args a, const char *name, std::vector<uint32_t> *res)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = (a.n / PX + 255) / 256;
    for (int i = 0; i < 60; i++) hipLaunchKernelGGL((k<V, PX>), dim3(blocks), dim3(256), 0, 0, a);
    hipEventRecord(e0);
    for (int i = 0; i < 40; i++) hipLaunchKernelGGL((k<V, PX>), dim3(blocks), dim3(256), 0, 0, a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (res) { res->resize(a.n); hipMemcpy(res->data(), a.out, 4ull * a.n, hipMemcpyDeviceToHost); }
    printf("%-40s %8.1f us per 4K image\n", name, 1000.0f * ms / 40);
    return ms;
}

int main()
{
    const int n = 3840 * 2160, np = 256;
    std::vector<uint32_t> src(n), comp(np), konst(np);
    srand(1);
    for (auto &v : src) v = (uint32_t)rand() ^ ((uint32_t)rand() << 16);
    for (int i = 0; i < np; i++) {
        const uint32_t r = rand() & 255, g = rand() & 255, b = rand() & 255;
        comp[i] = (255 - r) | ((255 - g) << 8) | ((255 - b) << 16);
        konst[i] = (r * r + g * g + b * b) * 256 + i;
    }
    Args a{};
    uint32_t *ds, *dout, *dc, *dk;
    hipMalloc(&ds, 4ull * n); hipMalloc(&dout, 4ull * n); hipMalloc(&dc, 4 * np); hipMalloc(&dk, 4 * np);
    hipMemcpy(ds, src.data(), 4ull * n, hipMemcpyHostToDevice);
    hipMemcpy(dc, comp.data(), 4 * np, hipMemcpyHostToDevice);
    hipMemcpy(dk, konst.data(), 4 * np, hipMemcpyHostToDevice);
    a.src = ds; a.out = dout; a.n = n; a.npad = np; a.comp = dc; a.konst = dk;
    std::vector<uint32_t> ra, rb, rc;
    run<0, 4>(a, "A dot4 + lshl_add + min, 4 px", &ra);
    run<1, 4>(a, "B min3, 4 px", &rb);
    run<1, 8>(a, "C min3, 8 px", &rc);
    run<0, 8>(a, "A8 dot4 + lshl_add + min, 8 px", nullptr);
    run<2, 4>(a, "D dot4 + add only", nullptr);
    run<3, 4>(a, "E lshl_add + min + xor", nullptr);
    printf("B == A: %d, C == A: %d\n", (int)(ra == rb), (int)(ra == rc));
    return 0;
}
