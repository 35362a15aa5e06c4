// Memory-pattern probe 2 (experiments only): tiled copy, wave strip SW px wide, marching down seg rows.
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int xcd_tile(int bid, int total)
{
    int per = (total + 7) >> 3;
    int t = (bid & 7) * per + (bid >> 3);
    return ((bid >> 3) < per && t < total) ? t : -1;
}
// SW: px per wave strip; lanes: chunk = lane % (SW/4), row = lane / (SW/4); rows per instr RPI = 256/SW
// NT: non-temporal hints
template <int SW, int WAVES, bool NT, bool XCD>
__global__ __launch_bounds__(64 * WAVES) void pat(const uint8_t* src0, uint8_t* dst0, size_t img, int stride, int w, int h, int tiles_x, int tiles, int seg)
{
    const int tile = XCD ? xcd_tile(blockIdx.x, tiles) : (blockIdx.x < tiles ? (int)blockIdx.x : -1);
    if (tile < 0) return;
    const uint8_t* src = src0 + img * blockIdx.y;
    uint8_t* dst = dst0 + img * blockIdx.y;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int x0 = tx * SW * WAVES, y0 = ty * seg;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int xs = x0 + SW * wave;
    constexpr int CPR = SW / 4, RPI = 64 / CPR;
    const int ch = lane % CPR, row = lane / CPR;
    if (xs + SW > w) return;
    for (int j = 0; j < seg / RPI; j++) {
        const int y = y0 + j * RPI + row;
        if (y >= h) break;
        const size_t off = (size_t)y * stride + 4 * xs + 16 * ch;
        u32x4 d;
        if (NT) d = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src + off));
        else d = *reinterpret_cast<const u32x4*>(src + off);
        d ^= 0x80808080u;
        if (NT) __builtin_nontemporal_store(d, reinterpret_cast<u32x4*>(dst + off));
        else *reinterpret_cast<u32x4*>(dst + off) = d;
    }
}
template <int SW, int WAVES, bool NT, bool XCD> void go(const uint8_t* s, uint8_t* d, int seg)
{
    const int w = 3840, h = 2160, n = 32;
    const size_t img = (size_t)w * h * 4;
    int tiles_x = w / (SW * WAVES), tiles = tiles_x * ((h + seg - 1) / seg);
    dim3 grid(XCD ? 8 * ((tiles + 7) / 8) : tiles, n);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) pat<SW, WAVES, NT, XCD><<<grid, 64 * WAVES>>>(s, d, img, 4 * w, w, h, tiles_x, tiles, seg);
    hipEventRecord(e0);
    for (int i = 0; i < 20; i++) pat<SW, WAVES, NT, XCD><<<grid, 64 * WAVES>>>(s, d, img, 4 * w, w, h, tiles_x, tiles, seg);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("SW %3d waves %d nt %d xcd %d seg %4d: %.2f us per image (%.2f TB/s)\n", SW, WAVES, (int)NT, (int)XCD, seg, ms * 1000 / 20 / n, 2.0 * img / (ms * 1000 / 20 / n) / 1e6);
}
int main()
{
    const size_t img = (size_t)3840 * 2160 * 4;
    uint8_t *s, *d; hipMalloc(&s, img * 32); hipMalloc(&d, img * 32);
    hipMemset(s, 1, img * 32);
    for (int seg : {128, 540}) {
        go<16, 4, false, true>(s, d, seg);
        go<64, 4, false, true>(s, d, seg);
        go<64, 1, false, true>(s, d, seg);
        go<256, 1, false, true>(s, d, seg);
        go<256, 1, true, true>(s, d, seg);
        go<256, 1, false, false>(s, d, seg);
        go<64, 4, true, true>(s, d, seg);
        go<16, 4, true, true>(s, d, seg);
        go<256, 3, false, true>(s, d, seg);
    }
    return 0;
}
