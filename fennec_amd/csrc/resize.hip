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

// one tap: aw = sa*w; r += R*aw; g += G*aw; b += B*aw; a += aw   (resize.go:95-103)
__device__ __forceinline__ void resize_tap(uint32_t p, double w, double &r, double &g, double &b, double &al)
{
    const double aw = u8_to_f64(p >> 24) * w;
    r += u8_to_f64(p & 0xffu) * aw;
    g += u8_to_f64((p >> 8) & 0xffu) * aw;
    b += u8_to_f64((p >> 16) & 0xffu) * aw;
    al += aw;
}

template <bool VERT>
__global__ __launch_bounds__(256) void resize_pass_kernel(ResizeArgs a)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    // a wave is one row of 64 outputs: in the V pass its tap list is wave-uniform, so make the
    // row index provably scalar and the table reads become s_load (no VMEM, no per-lane latency)
    const int y = blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (x >= a.outW || y >= a.outH) return;
    const int d = VERT ? y : x;
    const int t0 = a.off[d], t1 = a.off[d + 1];
    const uint8_t *hrow = a.src + static_cast<size_t>(y) * a.sstride;
    auto px = [&](int s) -> uint32_t {
        return VERT ? ld_px(a.src + static_cast<size_t>(s) * a.sstride, x) : ld_px(hrow, s);
    };
    double r = 0, g = 0, b = 0, al = 0;
    int t = t0;
    // 4 taps per trip: the 4 index loads, then the 4 pixel loads, are in flight together; the
    // arithmetic stays strictly in tap order
    for (; t + 4 <= t1; t += 4) {
        const int s0 = a.idx[t], s1 = a.idx[t + 1], s2 = a.idx[t + 2], s3 = a.idx[t + 3];
        const double w0 = a.wt[t], w1 = a.wt[t + 1], w2 = a.wt[t + 2], w3 = a.wt[t + 3];
        const uint32_t p0 = px(s0), p1 = px(s1), p2 = px(s2), p3 = px(s3);
        resize_tap(p0, w0, r, g, b, al);
        resize_tap(p1, w1, r, g, b, al);
        resize_tap(p2, w2, r, g, b, al);
        resize_tap(p3, w3, r, g, b, al);
    }
    for (; t < t1; t++) resize_tap(px(a.idx[t]), a.wt[t], r, g, b, al);
    uint32_t o = 0;                                             // zero-initialised dst pixel
    if (al > 0.5) {                                             // resize.go:107-113
        const double inv = 1.0 / al;
        o = clampF_dev(r * inv) | (clampF_dev(g * inv) << 8) | (clampF_dev(b * inv) << 16) |
            (clampF_dev(al) << 24);
    }
    *(g_u32w *)(a.dst + static_cast<size_t>(y) * a.dstride + 4 * static_cast<size_t>(x)) = o;
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
