// Lanczos-3 separable resize passes (resize.go:77-161) on gfx950.
// fp64, unfused, taps in table order, premultiplied-alpha accumulation exactly as the
// reference (TU built with -ffp-contract=off) => bit-exact uint8 output given the same
// tap table.  Thread per output pixel; lanes run along x in both passes so the V pass
// (which the reference walks column-wise, stride-hostile on a CPU) is coalesced here.
#include "common.hpp"
#include "devutil.hpp"

namespace fnx {

struct ResizeArgs {
    const uint8_t *src;
    uint8_t *dst;
    int sstride, dstride;
    int outW, outH;           // dst dims of this pass
    const int32_t *off;
    const int32_t *idx;
    const double *wt;
};

template <bool VERT>
__global__ __launch_bounds__(256) void resize_pass_kernel(ResizeArgs a)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.outW || y >= a.outH) return;
    const int d = VERT ? y : x;
    const int t0 = a.off[d], t1 = a.off[d + 1];
    double r = 0, g = 0, b = 0, al = 0;
    for (int t = t0; t < t1; t++) {
        const int s = a.idx[t];
        const uint32_t p = VERT ? ld_px(a.src + static_cast<size_t>(s) * a.sstride, x)
                                : ld_px(a.src + static_cast<size_t>(y) * a.sstride, s);
        const double sa = static_cast<double>(p >> 24);
        const double w = a.wt[t];
        const double aw = sa * w;                               // resize.go:99
        r += static_cast<double>(p & 0xffu) * aw;              // resize.go:100-103
        g += static_cast<double>((p >> 8) & 0xffu) * aw;
        b += static_cast<double>((p >> 16) & 0xffu) * aw;
        al += aw;
    }
    uint32_t o = 0;                                             // zero-initialised dst pixel
    if (al > 0.5) {                                             // resize.go:107-113
        const double inv = 1.0 / al;
        o = clampF_dev(r * inv) | (clampF_dev(g * inv) << 8) | (clampF_dev(b * inv) << 16) |
            (clampF_dev(al) << 24);
    }
    *reinterpret_cast<uint32_t *>(a.dst + static_cast<size_t>(y) * a.dstride + 4 * static_cast<size_t>(x)) = o;
}

int launch_resize_h(fnx_ctx *ctx, const uint8_t *src, int sstride, int srcW, int srcH,
                    const int32_t *d_off, const int32_t *d_idx, const double *d_wt, uint8_t *dst,
                    int dstride, int dstW)
{
    (void)srcW;
    if (dstW <= 0 || srcH <= 0) return FNX_OK;
    ResizeArgs a{src, dst, sstride, dstride, dstW, srcH, d_off, d_idx, d_wt};
    dim3 grid((dstW + 63) / 64, (srcH + 3) / 4);
    hipLaunchKernelGGL((resize_pass_kernel<false>), grid, dim3(256), 0, ctx->stream, a);
    FNX_HIP(hipGetLastError());
    return FNX_OK;
}

int launch_resize_v(fnx_ctx *ctx, const uint8_t *src, int sstride, int srcW, int srcH,
                    const int32_t *d_off, const int32_t *d_idx, const double *d_wt, uint8_t *dst,
                    int dstride, int dstH)
{
    (void)srcH;
    if (srcW <= 0 || dstH <= 0) return FNX_OK;
    ResizeArgs a{src, dst, sstride, dstride, srcW, dstH, d_off, d_idx, d_wt};
    dim3 grid((srcW + 63) / 64, (dstH + 3) / 4);
    hipLaunchKernelGGL((resize_pass_kernel<true>), grid, dim3(256), 0, ctx->stream, a);
    FNX_HIP(hipGetLastError());
    return FNX_OK;
}

}  // namespace fnx
