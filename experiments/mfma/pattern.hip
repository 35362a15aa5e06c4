// Memory-pattern probe (experiments only): what the strip-marching load/store pattern costs without any arithmetic.
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
// MODE 0: loads as the H pass (16 rows x 8 chunks, 2 instr), stores as the V pass (lane = row r, chunk g)
// MODE 1: loads the same, stores 4 rows x 256 B per wave instr?? (not possible per wave: 64 B wide) -> stores lane=(row l>>2, chunk l&3)
// MODE 2: loads only the 64-byte strip (lane = row l>>2, chunk l&3), stores the same way: a plain tiled copy
// MODE 3: as 0 without stores; MODE 4: as 0 without loads
template <int MODE>
__global__ __launch_bounds__(256) void pat(const uint8_t* src0, uint8_t* dst0, size_t img, int stride, int w, int h, int tiles_x, int tiles, int seg)
{
    const int tile = xcd_tile(blockIdx.x, tiles);
    if (tile < 0) return;
    const uint8_t* src = src0 + img * blockIdx.y;
    uint8_t* dst = dst0 + img * blockIdx.y;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int x0 = tx * 64, y0 = ty * seg;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int xs = x0 + 16 * wave;
    const int r = lane & 15, g = lane >> 4, lrow = lane >> 3, lch = (lane & 7) > 6 ? 6 : (lane & 7);
    if (xs - 6 < 0 || xs + 21 >= w || y0 - 6 < 0 || y0 + seg + 16 > h) return;   // interior only
    u32x4 acc = {0, 0, 0, 0};
    for (int j = 0; j < seg / 16; j++) {
        u32x4 d0 = {1, 2, 3, 4}, d1 = {5, 6, 7, 8};
        if (MODE == 0 || MODE == 1 || MODE == 3) {
            const uint8_t* sb = src + (size_t)(y0 - 6 + 16 * j) * stride + 4 * (xs - 6);
            d0 = *reinterpret_cast<const u32x4*>(sb + lrow * stride + 16 * lch);
            d1 = *reinterpret_cast<const u32x4*>(sb + (lrow + 8) * stride + 16 * lch);
        } else if (MODE == 2) {
            d0 = *reinterpret_cast<const u32x4*>(src + (size_t)(y0 + 16 * j + (lane >> 2)) * stride + 4 * xs + 16 * (lane & 3));
        }
        acc ^= d0 ^ d1;
        if (MODE == 0 || MODE == 4) {
            *reinterpret_cast<u32x4*>(dst + (size_t)(y0 + 16 * j + r) * stride + 4 * (xs + 4 * g)) = acc;
        } else if (MODE == 1 || MODE == 2) {
            *reinterpret_cast<u32x4*>(dst + (size_t)(y0 + 16 * j + (lane >> 2)) * stride + 4 * xs + 16 * (lane & 3)) = acc;
        }
    }
    if (MODE == 3 && acc[0] == 0x12345678u) dst[0] = 1;
}
template <int MODE> void go(const uint8_t* s, uint8_t* d, int seg)
{
    const int w = 3840, h = 2160, n = 32;
    const size_t img = (size_t)w * h * 4;
    int tiles_x = w / 64, tiles = tiles_x * ((h + seg - 1) / seg);
    dim3 grid(8 * ((tiles + 7) / 8), n);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) pat<MODE><<<grid, 256>>>(s, d, img, 4 * w, w, h, tiles_x, tiles, seg);
    hipEventRecord(e0);
    for (int i = 0; i < 20; i++) pat<MODE><<<grid, 256>>>(s, d, img, 4 * w, w, h, tiles_x, tiles, seg);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d seg %d: %.2f us per image\n", MODE, seg, ms * 1000 / 20 / n);
}
int main()
{
    const size_t img = (size_t)3840 * 2160 * 4;
    uint8_t *s, *d; hipMalloc(&s, img * 32); hipMalloc(&d, img * 32);
    hipMemset(s, 1, img * 32);
    for (int seg : {128, 256}) { go<0>(s, d, seg); go<1>(s, d, seg); go<2>(s, d, seg); go<3>(s, d, seg); go<4>(s, d, seg); }
    return 0;
}
