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

// A four-component file (r5): reader.go applyBlack makes an image.CMYK of the four planes -- Adobe transform 0: C, M, Y, K = 255 - the
// stored samples; any other transform (YCbCrK): the first three planes through color.YCbCrToRGB stand for C, M, Y as they are (the
// RGB -> CMY inversion cancels Adobe's), K = 255 - s -- and convert.go:34-64 reads it through color.CMYK.RGBA():
// w = 0xffff - K * 0x101, r = (0xffff - C * 0x101) * w / 0xffff, then the opaque branch's r >> 8.
struct CmykArgs {
    const uint8_t *p[4];
    uint8_t *dst;
    int stride, dstride, w, h, adobe;
};

__global__ __launch_bounds__(256) void cmyk_to_nrgba_kernel(CmykArgs a)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= a.w) return;
    const size_t i = static_cast<size_t>(y) * a.stride + x;
    uint32_t c, m, yy;
    const uint32_t k = 255u - a.p[3][i];
    if (a.adobe == 0) {
        c = 255u - a.p[0][i]; m = 255u - a.p[1][i]; yy = 255u - a.p[2][i];
    } else {
        const int32_t y1 = static_cast<int32_t>(a.p[0][i]) * 0x10101, cb1 = static_cast<int32_t>(a.p[1][i]) - 128, cr1 = static_cast<int32_t>(a.p[2][i]) - 128;
        const int32_t r = y1 + 91881 * cr1, g = y1 - 22554 * cb1 - 46802 * cr1, b = y1 + 116130 * cb1;
        c = (static_cast<uint32_t>(r) & 0xff000000u) == 0 ? static_cast<uint32_t>(r >> 16) : (r < 0 ? 0u : 255u);
        m = (static_cast<uint32_t>(g) & 0xff000000u) == 0 ? static_cast<uint32_t>(g >> 16) : (g < 0 ? 0u : 255u);
        yy = (static_cast<uint32_t>(b) & 0xff000000u) == 0 ? static_cast<uint32_t>(b >> 16) : (b < 0 ? 0u : 255u);
    }
    const uint32_t w = 0xffffu - k * 0x101u;
    const uint32_t r8 = ((0xffffu - c * 0x101u) * w / 0xffffu) >> 8, g8 = ((0xffffu - m * 0x101u) * w / 0xffffu) >> 8,
                   b8 = ((0xffffu - yy * 0x101u) * w / 0xffffu) >> 8;
    *reinterpret_cast<uint32_t *>(a.dst + static_cast<size_t>(y) * a.dstride + 4 * static_cast<size_t>(x)) = r8 | (g8 << 8) | (b8 << 16) | 0xff000000u;
}

int launch_cmyk_to_nrgba(fnx_ctx *ctx, const uint8_t *const planes[4], int stride, int adobe, int w, int h, uint8_t *dst, int dstride)
{
    if (w <= 0 || h <= 0) return FNX_OK;
    CmykArgs a{};
    for (int c = 0; c < 4; c++) a.p[c] = planes[c];
    a.dst = dst; a.stride = stride; a.dstride = dstride; a.w = w; a.h = h; a.adobe = adobe;
    hipLaunchKernelGGL(cmyk_to_nrgba_kernel, dim3((w + 255) / 256, h), dim3(256), 0, ctx->stream, a);
    FNX_HIP(hipGetLastError());
    return FNX_OK;
}

}  // namespace fnx
