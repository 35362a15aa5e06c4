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

// H pass with the tap table in registers.  Every row of the image uses the SAME taps for output
// column x (resize.go:82: weights are per column), so a lane owns one column, loads its <= 4*NV taps
// once and walks down RH_ROWS rows: per output pixel that is NV 16-byte loads of the (contiguous)
// source window instead of one index + one weight + one pixel load per tap.  Arithmetic per tap and
// tap order are resize_tap's; columns with fewer taps are padded with zero weights, which add +0.0
// to the accumulators and cannot change a clampF result.  Requires contiguous tap indices (the host
// checks; zero-weight taps dropped by precomputeWeights can leave gaps -> generic kernel).
constexpr int RH_ROWS = 16;

struct ResizeRowsArgs {
    const uint8_t *src;
    uint8_t *dst;
    int sstride, dstride;
    int outW, rows, srcW;
    const int32_t *off;
    const int32_t *idx;
    const double *wt;
};

template <int NV>
__global__ __launch_bounds__(256) void resize_h_rows_kernel(ResizeRowsArgs a)
{
    constexpr int NT = 4 * NV;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * RH_ROWS;
    if (x >= a.outW || y0 >= a.rows) return;
    const int t0 = a.off[x], n = a.off[x + 1] - t0;
    const int s0 = n > 0 ? a.idx[t0] : 0;                       // first source column of the window
    const bool vec = s0 + NT <= a.srcW;                         // whole window inside the row
    const int y1 = min(a.rows, y0 + RH_ROWS);
    auto load_row = [&](int y, u32x4 (&v)[NV]) {
        const uint8_t *row = a.src + static_cast<size_t>(y) * a.sstride;
        if (vec) {
#pragma unroll
            for (int q = 0; q < NV; q++) v[q] = *(g_u32x4 *)(row + 4 * static_cast<size_t>(s0 + 4 * q));
        } else {
#pragma unroll
            for (int q = 0; q < NV; q++)
#pragma unroll
                for (int e = 0; e < 4; e++) v[q][e] = ld_px(row, min(s0 + 4 * q + e, a.srcW - 1));
        }
    };
    // Opaque windows (every photograph): with A == 255 the per-tap alpha weight aw = 255 * w and the
    // alpha sum are the same for every row of this column, so they -- and the reference's 1.0 / a -- are
    // computed once per lane, by the same operations in the same order (resize.go:95-113), and a row
    // costs 3 multiply-adds per tap instead of a convert, 4 multiply-adds and a division.  Bit-identical
    // by construction; rows whose wave sees any other alpha take the general path.
    double aw[NT], al255 = 0.0;
#pragma unroll
    for (int k = 0; k < NT; k++) {
        aw[k] = 255.0 * (k < n ? a.wt[t0 + k] : 0.0);
        al255 += aw[k];
    }
    const bool al_ok = al255 > 0.5;
    const double inv255 = al_ok ? 1.0 / al255 : 0.0;
    const uint32_t a255 = al_ok ? clampF_dev(al255) << 24 : 0u;
    for (int y = y0; y < y1; y++) {
        u32x4 v[NV];
        load_row(y, v);
        uint32_t andp = 0xffffffffu;
#pragma unroll
        for (int q = 0; q < NV; q++) andp &= (v[q][0] & v[q][1]) & (v[q][2] & v[q][3]);
        uint32_t o = 0;                                         // zero-initialised dst pixel
        if (__all((andp >> 24) == 0xffu)) {                     // wave-uniform: no divergence
            double r = 0, g = 0, b = 0;
#pragma unroll
            for (int k = 0; k < NT; k++) {
                const uint32_t p = v[k / 4][k % 4];
                r += u8_to_f64(p & 0xffu) * aw[k];
                g += u8_to_f64((p >> 8) & 0xffu) * aw[k];
                b += u8_to_f64((p >> 16) & 0xffu) * aw[k];
            }
            if (al_ok) o = clampF_dev(r * inv255) | (clampF_dev(g * inv255) << 8) | (clampF_dev(b * inv255) << 16) | a255;
        } else {
            double r = 0, g = 0, b = 0, al = 0;                 // weights re-read from the table: this path is rare
#pragma unroll
            for (int k = 0; k < NT; k++) resize_tap(v[k / 4][k % 4], k < n ? a.wt[t0 + k] : 0.0, r, g, b, al);
            if (al > 0.5) {                                     // resize.go:107-113
                const double inv = 1.0 / al;
                o = clampF_dev(r * inv) | (clampF_dev(g * inv) << 8) | (clampF_dev(b * inv) << 16) |
                    (clampF_dev(al) << 24);
            }
        }
        *(g_u32w *)(a.dst + static_cast<size_t>(y) * a.dstride + 4 * static_cast<size_t>(x)) = o;
    }
}

// V pass for contiguous tap lists.  A workgroup owns 256 columns x RV_G consecutive output rows and walks
// DOWN the source rows those outputs use, once: every source pixel is loaded and converted one time and
// feeds each output row whose window holds it (at a 2x downscale a source row serves 6 output rows, and
// the thread-per-output kernel converted it 6 times: PMC counted 24-31 VALU instructions per tap, 16 of
// them conversions and addressing).  Taps are row-uniform, so the window tests are scalar branches and the
// weights come from a small LDS table (one broadcast read per use).  Per output the arithmetic is
// resize_tap's, in ascending tap order (= ascending source row, the lists being contiguous).
constexpr int RV_G = 4, RV_NT = 32;

struct ResizeColsArgs {
    const uint8_t *src;
    uint8_t *dst;
    int sstride, dstride;
    int w, outH;
    const int32_t *off;
    const int32_t *idx;
    const double *wt;
};

__global__ __launch_bounds__(256) void resize_v_cols_kernel(ResizeColsArgs a)
{
    __shared__ double s_w[RV_G][RV_NT];
    __shared__ int s_s0[RV_G], s_n[RV_G];
    const int tid = threadIdx.x;
    const int y0 = blockIdx.y * RV_G;
    if (tid < RV_G) {
        const int y = y0 + tid;
        int n = 0, s0 = 0;
        if (y < a.outH) {
            const int t0 = a.off[y];
            n = a.off[y + 1] - t0;
            s0 = n > 0 ? a.idx[t0] : 0;
        }
        s_s0[tid] = s0;
        s_n[tid] = n;
    }
    if (tid < RV_G * RV_NT) {
        const int g = tid / RV_NT, k = tid - g * RV_NT, y = y0 + g;
        double w = 0.0;
        if (y < a.outH) {
            const int t0 = a.off[y];
            if (k < a.off[y + 1] - t0) w = a.wt[t0 + k];
        }
        s_w[g][k] = w;
    }
    __syncthreads();
    const int x = blockIdx.x * 256 + tid;
    if (x >= a.w) return;
    int s0[RV_G], n[RV_G];
    int smin = 0x7fffffff, smax = 0;
#pragma unroll
    for (int g = 0; g < RV_G; g++) {
        s0[g] = __builtin_amdgcn_readfirstlane(s_s0[g]);
        n[g] = __builtin_amdgcn_readfirstlane(s_n[g]);
        if (n[g] > 0) {
            smin = min(smin, s0[g]);
            smax = max(smax, s0[g] + n[g]);
        }
    }
    double r[RV_G], gg[RV_G], b[RV_G], al[RV_G];
#pragma unroll
    for (int g = 0; g < RV_G; g++) r[g] = gg[g] = b[g] = al[g] = 0;
    const uint8_t *col = a.src + 4 * static_cast<size_t>(x);
    // software pipeline: the next row's pixel is in flight while this row is used, and the RV_G weights
    // of a row are read together (clamped index; unused ones are simply not applied) -- one LDS wait per
    // row instead of one per use
    uint32_t pn = smin < smax ? *(g_u32 *)(col + static_cast<size_t>(smin) * a.sstride) : 0u;
    for (int s = smin; s < smax; s++) {
        const uint32_t p = pn;
        if (s + 1 < smax) pn = *(g_u32 *)(col + static_cast<size_t>(s + 1) * a.sstride);
        double w[RV_G];
#pragma unroll
        for (int g = 0; g < RV_G; g++) w[g] = s_w[g][min(max(s - s0[g], 0), RV_NT - 1)];
        const double cr = u8_to_f64(p & 0xffu), cg = u8_to_f64((p >> 8) & 0xffu), cb = u8_to_f64((p >> 16) & 0xffu);
        const double ca = u8_to_f64(p >> 24);
#pragma unroll
        for (int g = 0; g < RV_G; g++) {
            if (static_cast<unsigned>(s - s0[g]) < static_cast<unsigned>(n[g])) {
                const double aw = ca * w[g];                    // resize.go:95-103, as resize_tap
                r[g] += cr * aw;
                gg[g] += cg * aw;
                b[g] += cb * aw;
                al[g] += aw;
            }
        }
    }
#pragma unroll
    for (int g = 0; g < RV_G; g++) {
        const int y = y0 + g;
        if (y < a.outH) {
            uint32_t o = 0;
            if (al[g] > 0.5) {                                  // resize.go:107-113
                const double inv = 1.0 / al[g];
                o = clampF_dev(r[g] * inv) | (clampF_dev(gg[g] * inv) << 8) | (clampF_dev(b[g] * inv) << 16) |
                    (clampF_dev(al[g]) << 24);
            }
            *(g_u32w *)(a.dst + static_cast<size_t>(y) * a.dstride + 4 * static_cast<size_t>(x)) = o;
        }
    }
}

// max taps of any output (0: indices not contiguous somewhere) -- host tables
int resize_contiguous_taps(const int32_t *off, const int32_t *idx, int nout)
{
    int maxn = 0;
    for (int d = 0; d < nout; d++) {
        const int n = off[d + 1] - off[d];
        for (int k = 1; k < n; k++)
            if (idx[off[d] + k] != idx[off[d]] + k) return 0;
        if (n > maxn) maxn = n;
    }
    return maxn;
}

// contig_taps: resize_contiguous_taps() of the H table (0: unknown / not contiguous)
int launch_resize_h(fnx_ctx *ctx, const uint8_t *src, int sstride, int srcW, int srcH,
                    const int32_t *d_off, const int32_t *d_idx, const double *d_wt, uint8_t *dst,
                    int dstride, int dstW, int contig_taps)
{
    if (dstW <= 0 || srcH <= 0) return FNX_OK;
    if (contig_taps > 0 && contig_taps <= 16 && srcW >= 4) {
        ResizeRowsArgs ra{src, dst, sstride, dstride, dstW, srcH, srcW, d_off, d_idx, d_wt};
        dim3 grid((dstW + 63) / 64, (srcH + 4 * RH_ROWS - 1) / (4 * RH_ROWS));
        if (contig_taps <= 8) hipLaunchKernelGGL((resize_h_rows_kernel<2>), grid, dim3(256), 0, ctx->stream, ra);
        else if (contig_taps <= 12) hipLaunchKernelGGL((resize_h_rows_kernel<3>), grid, dim3(256), 0, ctx->stream, ra);
        else hipLaunchKernelGGL((resize_h_rows_kernel<4>), grid, dim3(256), 0, ctx->stream, ra);
        FNX_HIP(hipGetLastError());
        return FNX_OK;
    }
    ResizeArgs a{src, dst, sstride, dstride, dstW, srcH, d_off, d_idx, d_wt};
    dim3 grid((dstW + 63) / 64, (srcH + 3) / 4);
    hipLaunchKernelGGL((resize_pass_kernel<false>), grid, dim3(256), 0, ctx->stream, a);
    FNX_HIP(hipGetLastError());
    return FNX_OK;
}

int launch_resize_v(fnx_ctx *ctx, const uint8_t *src, int sstride, int srcW, int srcH,
                    const int32_t *d_off, const int32_t *d_idx, const double *d_wt, uint8_t *dst,
                    int dstride, int dstH, int contig_taps)
{
    (void)srcH;
    if (srcW <= 0 || dstH <= 0) return FNX_OK;
    if (contig_taps > 0 && contig_taps <= RV_NT) {
        ResizeColsArgs ca{src, dst, sstride, dstride, srcW, dstH, d_off, d_idx, d_wt};
        hipLaunchKernelGGL(resize_v_cols_kernel, dim3((srcW + 255) / 256, (dstH + RV_G - 1) / RV_G), dim3(256), 0,
                           ctx->stream, ca);
        FNX_HIP(hipGetLastError());
        return FNX_OK;
    }
    ResizeArgs a{src, dst, sstride, dstride, srcW, dstH, d_off, d_idx, d_wt};
    dim3 grid((srcW + 63) / 64, (dstH + 3) / 4);
    hipLaunchKernelGGL((resize_pass_kernel<true>), grid, dim3(256), 0, ctx->stream, a);
    FNX_HIP(hipGetLastError());
    return FNX_OK;
}

}  // namespace fnx
