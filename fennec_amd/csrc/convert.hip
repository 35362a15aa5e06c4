// toNRGBARef / convertToNRGBA (convert.go:22-64) of a decoded JPEG: image.YCbCr (or image.Gray)
// planes -> NRGBA on gfx950.  SURVEY 8(f) item 1: the decoder's planes cross PCIe (1.5 bytes per
// pixel at 4:2:0 instead of 4) and the per-pixel At().RGBA() loop of the reference -- slow on the
// host -- runs here.  Integer arithmetic of Go's image/color package (image.YCbCr.COffset +
// color.YCbCr.RGBA(), go.mod:3 pins go 1.25.5; the published algorithm is quoted below):
// bit-exact against the CPU restatement the tests hold.
#include "common.hpp"
#include "devutil.hpp"

namespace fnx {

struct YccArgs {
    const uint8_t *y, *cb, *cr;   // cb == nullptr: image.Gray
    uint8_t *dst;
    int ystride, cstride, dstride, w, h;
    int xshift, yshift;           // chroma sample of (x, y) = (x >> xshift, y >> yshift): image.YCbCr.COffset
};

// (the per-pixel arithmetic is devutil.hpp's ycc_nrgba_px)
__global__ __launch_bounds__(256) void ycbcr_to_nrgba_kernel(YccArgs a)
{
    const int x0 = 4 * (blockIdx.x * 64 + (threadIdx.x & 63));
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x0 >= a.w || y >= a.h) return;
    const uint8_t *yrow = a.y + static_cast<size_t>(y) * a.ystride;
    const int cnt = min(4, a.w - x0);
    uint32_t out[4];
    if (!a.cb) {
        for (int e = 0; e < cnt; e++) {
            const uint32_t v = yrow[x0 + e];                      // color.Gray.RGBA() = y * 0x101, >> 8 = y
            out[e] = v | (v << 8) | (v << 16) | 0xff000000u;
        }
    } else {
        const size_t crow = static_cast<size_t>(y >> a.yshift) * a.cstride;
        for (int e = 0; e < cnt; e++) {
            const int x = x0 + e;
            out[e] = ycc_nrgba_px(yrow[x], a.cb[crow + (x >> a.xshift)], a.cr[crow + (x >> a.xshift)]);
        }
    }
    uint8_t *dp = a.dst + static_cast<size_t>(y) * a.dstride + 4 * static_cast<size_t>(x0);
    if (cnt == 4 && (reinterpret_cast<uintptr_t>(dp) & 15u) == 0) {
        *reinterpret_cast<u32x4 *>(dp) = (u32x4){out[0], out[1], out[2], out[3]};
    } else {
        for (int e = 0; e < cnt; e++) *reinterpret_cast<uint32_t *>(dp + 4 * e) = out[e];
    }
}

// ratio: image.YCbCrSubsampleRatio (0 4:4:4, 1 4:2:2, 2 4:2:0, 3 4:4:0, 4 4:1:1, 5 4:1:0); device pointers
int launch_ycbcr_to_nrgba(fnx_ctx *ctx, const uint8_t *y, int ystride, const uint8_t *cb, const uint8_t *cr,
                          int cstride, int ratio, int w, int h, uint8_t *dst, int dstride)
{
    if (w <= 0 || h <= 0) return FNX_OK;
    static const int xs[6] = {0, 1, 1, 0, 2, 2}, ys[6] = {0, 0, 1, 1, 0, 1};
    YccArgs a{};
    a.y = y; a.cb = cb; a.cr = cr; a.dst = dst;
    a.ystride = ystride; a.cstride = cstride; a.dstride = dstride; a.w = w; a.h = h;
    a.xshift = xs[ratio]; a.yshift = ys[ratio];
    hipLaunchKernelGGL(ycbcr_to_nrgba_kernel, dim3((w + 255) / 256, (h + 3) / 4), dim3(256), 0, ctx->stream, a);
    FNX_HIP(hipGetLastError());
    return FNX_OK;
}

}  // namespace fnx
