// gaussianBlur3x3 / Sharpen / AdaptiveSharpen (effects.go:10-141) on gfx950.
// One fused launch per op: the 3x3 binomial blur is exact in integers
// (clampF(sum/16.0) == (sum+8)>>4 for non-negative integer sums), the unsharp step and the
// Sobel edge mask are fp64, unfused, in the reference's operation order (TU built with
// -ffp-contract=off; fp64 sqrt and divide are correctly rounded) => bit-exact uint8 output.
#include "common.hpp"
#include "devutil.hpp"

namespace fnx {

enum FxMode { FX_BLUR3 = 0, FX_SHARPEN = 1, FX_ADAPTIVE = 2 };

struct FxArgs {
    const uint8_t *src;
    uint8_t *dst;
    int sstride, dstride, w, h;
    double amount;
};

// Workgroup = 64 x 8 output pixels (2 per lane).  The 66 x 10 source tile is loaded once into
// LDS -- as two words per pixel with the channels spread into 16-bit fields (R|B<<16, G|A<<16),
// plus its BT.601 luminance -- so every pixel's luminance is computed once instead of once per
// Sobel neighbour (8x), the 3x3 neighbourhood costs LDS reads, not global loads, and the binomial
// sum (max 16*255 per field) runs on two channels per add.
constexpr int FX_TW = 64, FX_TH = 8;

template <int MODE>
__global__ __launch_bounds__(256) void fx_kernel(FxArgs a)
{
    constexpr int LW = FX_TW + 2, LH = FX_TH + 2;
    __shared__ uint32_t s_rb[LH * LW], s_ga[LH * LW];
    __shared__ double s_lum[MODE == FX_ADAPTIVE ? LH * LW : 1];
    const int x0 = blockIdx.x * FX_TW, y0 = blockIdx.y * FX_TH;
    const int tid = threadIdx.x;
    for (int i = tid; i < LH * LW; i += 256) {
        const int ly = i / LW, lx = i - ly * LW;
        // clamped reads: out-of-image tile cells are only ever neighbours of border pixels, which
        // are copies of the source and never look at them
        const int x = clampi(x0 + lx - 1, 0, a.w - 1), y = clampi(y0 + ly - 1, 0, a.h - 1);
        const uint32_t p = ld_px(a.src + static_cast<size_t>(y) * a.sstride, x);
        s_rb[i] = p & 0x00ff00ffu;
        s_ga[i] = (p >> 8) & 0x00ff00ffu;
        if (MODE == FX_ADAPTIVE) s_lum[i] = lum601(p);
    }
    __syncthreads();
    const int lx = tid & 63;
    const int x = x0 + lx;
    if (x >= a.w) return;
#pragma unroll
    for (int rep = 0; rep < 2; rep++) {
        const int ly = (tid >> 6) + 4 * rep;
        const int y = y0 + ly;
        if (y >= a.h) continue;
        const int ci = (ly + 1) * LW + lx + 1;          // tile cell of (x, y)
        const uint32_t crb = s_rb[ci], cga = s_ga[ci];
        const uint32_t c = crb | (cga << 8);
        uint32_t out = c;   // borders and alpha are copies of the source (effects.go:68,120)
        if (x >= 1 && y >= 1 && x < a.w - 1 && y < a.h - 1) {
            // [1 2 1; 2 4 2; 1 2 1] on two channels per word; (sum + 8) >> 4 per field (effects.go:125-135)
            const uint32_t srb = (s_rb[ci - LW - 1] + s_rb[ci - LW + 1] + s_rb[ci + LW - 1] + s_rb[ci + LW + 1]) +
                                 2 * (s_rb[ci - LW] + s_rb[ci - 1] + s_rb[ci + 1] + s_rb[ci + LW]) + 4 * crb + 0x00080008u;
            const uint32_t sga = (s_ga[ci - LW - 1] + s_ga[ci - LW + 1] + s_ga[ci + LW - 1] + s_ga[ci + LW + 1]) +
                                 2 * (s_ga[ci - LW] + s_ga[ci - 1] + s_ga[ci + 1] + s_ga[ci + LW]) + 4 * cga + 0x00080008u;
            const uint32_t blur[3] = {(srb >> 4) & 0xffu, (sga >> 4) & 0xffu, (srb >> 20) & 0xffu};
            if (MODE == FX_BLUR3) {
                out = blur[0] | (blur[1] << 8) | (blur[2] << 16) | (c & 0xff000000u);
            } else {
                double amt = a.amount;
                if (MODE == FX_ADAPTIVE) {                       // localEdgeStrength, effects.go:93-112
                    const double l00 = s_lum[ci - LW - 1], l01 = s_lum[ci - LW], l02 = s_lum[ci - LW + 1];
                    const double l10 = s_lum[ci - 1], l12 = s_lum[ci + 1];
                    const double l20 = s_lum[ci + LW - 1], l21 = s_lum[ci + LW], l22 = s_lum[ci + LW + 1];
                    const double gx = -l00 + l02 - 2 * l10 + 2 * l12 - l20 + l22;
                    const double gy = -l00 - 2 * l01 - l02 + l20 + 2 * l21 + l22;
                    const double mag = sqrt(gx * gx + gy * gy);
                    double normalized = mag / 400.0;
                    if (normalized > 1) normalized = 1;
                    amt = a.amount * normalized;                 // localAmount (effects.go:74)
                }
                out = c & 0xff000000u;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    const double orig = u8_to_f64((c >> (8 * ch)) & 0xffu);
                    const double bl = u8_to_f64(blur[ch]);
                    const double val = orig + amt * (orig - bl); // effects.go:37,82
                    out |= clampF_dev(val) << (8 * ch);
                }
            }
        }
        *(g_u32w *)(a.dst + static_cast<size_t>(y) * a.dstride + 4 * static_cast<size_t>(x)) = out;
    }
}

template <int MODE>
static int launch_fx(fnx_ctx *ctx, const uint8_t *src, int sstride, int w, int h, double amount,
                     uint8_t *dst, int dstride)
{
    if (w <= 0 || h <= 0) return FNX_OK;
    FxArgs a{src, dst, sstride, dstride, w, h, amount};
    dim3 grid((w + FX_TW - 1) / FX_TW, (h + FX_TH - 1) / FX_TH);
    hipLaunchKernelGGL((fx_kernel<MODE>), grid, dim3(256), 0, ctx->stream, a);
    FNX_HIP(hipGetLastError());
    return FNX_OK;
}

int launch_blur3x3(fnx_ctx *ctx, const uint8_t *src, int sstride, int w, int h, uint8_t *dst, int dstride)
{
    return launch_fx<FX_BLUR3>(ctx, src, sstride, w, h, 0.0, dst, dstride);
}

int launch_sharpen(fnx_ctx *ctx, bool adaptive, const uint8_t *src, int sstride, int w, int h,
                   double amount, uint8_t *dst, int dstride)
{
    return adaptive ? launch_fx<FX_ADAPTIVE>(ctx, src, sstride, w, h, amount, dst, dstride)
                    : launch_fx<FX_SHARPEN>(ctx, src, sstride, w, h, amount, dst, dstride);
}

}  // namespace fnx
