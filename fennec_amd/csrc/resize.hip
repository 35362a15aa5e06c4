// Lanczos-3 separable resize passes (resize.go:77-161) on gfx950.
// fp64, unfused, taps in table order, premultiplied-alpha accumulation exactly as the
// reference (TU built with -ffp-contract=off) => bit-exact uint8 output given the same
// tap table.  Thread per output pixel; lanes run along x in both passes so the V pass
// (which the reference walks column-wise, stride-hostile on a CPU) is coalesced here.
#include "common.hpp"
#include "devutil.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <memory>

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

// ------------------------------------------------------------------------------------
// Guard-exact fp32 passes (opaque windows: every photograph) -- the fast path of lanczosResize
// ------------------------------------------------------------------------------------
// For a window whose pixels all have A = 255 the reference's output (resize.go:93-113) is
//     clampF(fl(r * inv)),  r = sum_k fl(R_k * aw_k),  aw_k = fl(255 w_k),  inv = fl(1 / sum_k aw_k)
// whose real value is X = sum_k R_k W_k with W_k = aw_k / sum aw; the fp64 chain is within 1e-9 of X (~1e-13
// in fact).  The guard kernels accumulate  acc = (0.5 - G) + sum_k fl32(W_k) R_k  with fp32 FMAs: the weights
// are off by <= 2^-24 relative (255 S 2^-24 in all, S = sum |W_k|), every FMA rounds by at most half an ulp of a
// value below 255 S + 1, the seed add by the same; E is their sum over the dense window.  Packed (truncating,
// saturating) at acc and at acc + 2G with G > E + the second add's half ulp + 1e-6: when both bytes agree no
// integer lies within G - E of X + 0.5 and floor(X + 0.5) -- the reference's clampF -- is that byte.  When they
// differ, or when the window holds any alpha != 255, the output goes on a workgroup list and is recomputed by
// resize_exact_px: fp64, unfused, tap order, premultiplied alpha, the a > 0.5 rule -- the reference's arithmetic.
// A list overflow recomputes the workgroup's whole block that way.  Proven, not sampled.
//
// The taps of `HO` adjacent outputs are expanded on the host into a dense HO x NPX weight matrix over the union
// of their windows (zeros where an output does not use a pixel: fma(x, 0, acc) == acc exactly), so that the
// kernels' loops are fully unrolled although tap counts and window starts vary from output to output.
//   H pass: a lane owns one group of HO output columns, keeps its weight matrix in registers and walks down
//           `rows` rows; a pixel is converted once for all the outputs it feeds.
//   V pass: a lane owns 2 adjacent columns and VG = 4 consecutive output rows; the group's weights are
//           wave-uniform (scalar loads), every source row of the union is loaded and converted once.
constexpr int RG_FIX_CAP = 1024;
constexpr int RG_VG = 4;

struct ResizeGuardArgs {
    const uint8_t *src;
    uint8_t *dst;
    int sstride, dstride;
    int srcN;                 // source extent along the filtered axis
    int nout;                 // outputs along the filtered axis
    int other;                // extent of the other axis (rows for H, columns for V)
    int ngroups, rows;        // H: groups of HO outputs, rows per lane.  V: groups of VG rows
    int npx;                  // V: padded union rows per group (dense stride)
    float guard;
    const float *dense;       // H: [(j * NPX + i) * ngroups + g]   V: [(g * npx + i) * VG + j]
    const int32_t *s0;        // first source index of each group's union window
    const int32_t *cnt;       // V: union rows of each group
    const uint32_t *alpha;    // H: clampF(a) of the group's outputs, one byte each (output j in byte j)
                              // V: one word per output row, the byte already in bits 24..31
    const double *aw;         // fp64 aw = 255 w of the dense window (0.0 where unused): H [(j * NPX + i) * ngroups + g],
                              // V [(g * npx + i) * VG + j]
    const double *inv;        // 1.0 / a per output (a = sum of its aw, in tap order)
    // exact fix-ups
    const int32_t *off, *idx;
    const double *wt;
    // lanczosResize's two passes talk: the H pass leaves, per workgroup, how many of its 4 waves found their first row
    // dense with flags (0..4) in hint[by * hint_gx + bx]; the V pass of the SAME call reads the cells over its columns
    // and source rows and, where most waves were dense (ramps at integer ratios, translucent regions), goes straight
    // to its exact sweep instead of computing the fp32 form first and throwing it away.  A heuristic about cost only:
    // the exact sweep is the reference's arithmetic whatever the hint says.  nullptr: no hint.
    uint32_t *hint;
    int hint_gx, hint_gy, hint_rows;   // H grid; tmp rows per H workgroup (its columns: 64 * RG_HO outputs)
};

// one output exactly as resizeH / resizeV compute it (resize.go:93-113 / 137-156).  The guard kernels only run
// on contiguous tap lists, so tap k reads source index idx[t0] + k: the loop fetches 8 taps' weights and pixels
// with independent, branch-free loads (clamped to the last tap) before it applies them in order -- a tap at a
// time it is a chain of dependent cache misses, ~1.4 us per tap, and a handful of listed outputs per workgroup
// then outlast the whole fp32 pass.
template <bool VERT>
__device__ __forceinline__ uint32_t resize_exact_px(const ResizeGuardArgs &a, int x, int y)
{
    const int d = VERT ? y : x;
    const int t0 = a.off[d], n = a.off[d + 1] - t0;
    const int s0 = a.idx[t0];
    const uint8_t *base = VERT ? a.src + 4 * static_cast<size_t>(x) : a.src + static_cast<size_t>(y) * a.sstride;
    const size_t step = VERT ? static_cast<size_t>(a.sstride) : 4;
    double r = 0, g = 0, b = 0, al = 0;
    for (int k0 = 0; k0 < n; k0 += 8) {
        uint32_t p[8];
        double w[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int k = min(k0 + e, n - 1);
            p[e] = *(g_u32 *)(base + static_cast<size_t>(s0 + k) * step);
            w[e] = a.wt[t0 + k];
        }
#pragma unroll
        for (int e = 0; e < 8; e++)
            if (k0 + e < n) resize_tap(p[e], w[e], r, g, b, al);
    }
    uint32_t o = 0;
    if (al > 0.5) {
        const double inv = 1.0 / al;
        o = clampF_dev(r * inv) | (clampF_dev(g * inv) << 8) | (clampF_dev(b * inv) << 16) | (clampF_dev(al) << 24);
    }
    return o;
}

// acc.x += f * w.x, acc.y += f * w.y: v_pk_fma_f32 with the low half of the first source feeding both lanes
// (op_sel_hi), so the converted channel is not copied into a pair first
__device__ __forceinline__ v2f fma_bcast(float f, v2f w, v2f acc)
{
    v2f ff;
    asm("" : "=v"(ff));          // an (undefined) aligned register pair; only its low half is read
    ff.x = f;
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(ff), "v"(w));
    return acc;
}

constexpr int RG_HO = 2;    // outputs per H group.  Two is the sweet spot: at a 2x downscale the dense window is 16 pixels for
                            // 12 real taps (4 outputs: 20 for 12), a 32-entry weight matrix leaves room for 6 waves per SIMD
                            // (4 outputs: 80 entries, 1-2 waves), and converts + FMAs per output come out the same

// Flag handling.  Photographs flag ~0.1 % of the outputs (a rounding boundary within G): those go on a
// workgroup list and are recomputed by resize_exact_px after the main loop.  Synthetic content can flag
// everything -- a linear ramp at an integer ratio puts EVERY output of a row on an exact tie -- so a wave
// that finds RG_DENSE or more flagged lanes in a row leaves that row to a second, exact loop instead: fp64,
// unfused, ascending taps (dense window, zero weights add +0.0), the reference's arithmetic for opaque windows
// with the fp64 weights aw = 255 w staged in LDS and inv = 1 / sum aw; rows with a window that is not opaque
// call resize_exact_px.  The list therefore holds at most (RG_DENSE - 1) lanes x HO outputs per wave
// row and cannot overflow (RG_FIX_CAP >= 4 waves x 16 rows x 2 x (RG_DENSE - 1)).
constexpr int RG_DENSE = 8;
static_assert(4 * 16 * 2 * (RG_DENSE - 1) <= RG_FIX_CAP, "fix-up list capacity");

template <int NV, bool PF>
__global__ __launch_bounds__(256) void resize_h_guard_kernel(ResizeGuardArgs a)
{
    constexpr int HO = RG_HO, NPX = 4 * NV;
    __shared__ uint32_t s_fix[RG_FIX_CAP];
    __shared__ int s_nfix, s_ndense;
    // per workgroup (64 groups): fp64 aw [HO][NPX][64] for the exact loop, then the fp32 weight pairs
    // [NPX][64] (output 0, output 1) of the guard loop -- one conflict-free ds_read_b64 per pixel instead of
    // 2 NPX registers per lane (the 8-vector window then still runs at 4+ waves per SIMD with its prefetch)
    extern __shared__ __attribute__((aligned(16))) double s_aw[];
    v2f *s_w = reinterpret_cast<v2f *>(s_aw + HO * NPX * 64);
    const int tid = threadIdx.x;
    constexpr bool WREG = NV <= 4;      // the wave's weight pairs ride in registers (32 VGPRs) instead of 16 LDS reads per row
    if (tid == 0) { s_nfix = 0; s_ndense = 0; }
    if constexpr (WREG) __syncthreads();        // the list's counter is in place before anybody adds to it: crossed at once,
                                                // with nothing in flight yet (no weights are staged on this path)
    // issued before the staging below so that the three latencies overlap (a workgroup is short)
    const int gc_ = min(static_cast<int>(blockIdx.x) * 64 + (tid & 63), a.ngroups - 1);
    const int s0_ = a.s0[gc_];
    const uint32_t ab_ = a.alpha[gc_];
    // ... and the wave's first two rows go out before the weights are staged and the barrier is crossed (rows clamped into
    // the image: a wave past the last row reads it again and stores nothing)
    const int yw_ = (blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(tid >> 6)) * a.rows;
    u32x4 vn[PF ? NV : 1], vm[PF ? NV : 1];
    if constexpr (PF) {
        const uint8_t *r0 = a.src + static_cast<size_t>(min(yw_, a.other - 1)) * a.sstride;
        const uint8_t *r1 = a.src + static_cast<size_t>(min(yw_ + 1, min(a.other, yw_ + a.rows) - 1)) * a.sstride;   // in [0, other - 1] either way
#pragma unroll
        for (int q = 0; q < NV; q++) {
            vn[q] = *(g_u32x4 *)(r0 + 4 * static_cast<size_t>(s0_ + 4 * q));
            vm[q] = *(g_u32x4 *)(r1 + 4 * static_cast<size_t>(s0_ + 4 * q));
        }
    }
    // WREG: a lane's weight pairs straight from the plan's table (coalesced over the lanes' groups) -- no staging, no
    // barrier between a wave and its first row
    v2f wreg[WREG ? NPX : 1];
    if constexpr (WREG) {
#pragma unroll
        for (int i = 0; i < NPX; i++)
            wreg[i] = (v2f){a.dense[static_cast<size_t>(i) * a.ngroups + gc_], a.dense[static_cast<size_t>(NPX + i) * a.ngroups + gc_]};
    } else {
        const int g0 = blockIdx.x * 64;
        for (int e = tid; e < NPX * 64; e += 256) {
            const int gl = e & 63, i = e >> 6;
            const bool in = g0 + gl < a.ngroups;
            s_w[e] = (v2f){in ? a.dense[static_cast<size_t>(i) * a.ngroups + g0 + gl] : 0.0f,
                           in ? a.dense[static_cast<size_t>(NPX + i) * a.ngroups + g0 + gl] : 0.0f};
        }
        __syncthreads();
    }
    const int lane = tid & 63;
    const int g = blockIdx.x * 64 + lane;
    // first row of this wave: provably scalar, so row pointers are SGPR pairs and the window loads need one
    // 32-bit offset register each
    const int yw = yw_;
    const bool active = g < a.ngroups && yw < a.other;
    uint32_t exact_rows = 0;                                        // bit r: row yw + r of this wave awaits the exact loop
    if (yw < a.other) {                                             // wave-uniform
        const int gc = gc_;                                         // idle lanes shadow the last group (no stores)
        static_assert(HO == 2, "the two outputs of a group ride in the two lanes of the packed FMAs");
        const int s0 = s0_;                                          // the host keeps s0 + NPX <= srcN (shifted windows)
        const uint32_t ab = ab_;
        const int d0 = gc * HO;
        const bool full = d0 + HO <= a.nout;
        const bool st8 = full && ((reinterpret_cast<uintptr_t>(a.dst) | static_cast<uintptr_t>(a.dstride)) & 7u) == 0;
        const float seed = 0.5f - a.guard, g2 = 2.0f * a.guard;
        const int y1 = min(a.other, yw + a.rows);
        // NO branch may surround these loads: with one the compiler waits for vmcnt(0) in every iteration and the
        // prefetch below is lost (row indices are clamped instead of tested)
        auto load_row = [&](int y, u32x4 (&v)[NV]) {
            const uint8_t *row = a.src + static_cast<size_t>(y) * a.sstride;
#pragma unroll
            for (int q = 0; q < NV; q++) v[q] = *(g_u32x4 *)(row + 4 * static_cast<size_t>(s0 + 4 * q));
        };
        // a wave walks its rows one after the other, so what it has in flight is what it prefetched: two rows
        // ahead (the chip needs ~12 MB in flight to stream at HBM rate; one row ahead measured 2.7 TB/s at best)
        // (the first two rows were requested before the barrier)
        // rows whose flags are dense (ramps, flat ties, translucent regions) are left to the exact loop below;
        // while that lasts only every 4th row still tries the fp32 form (`sticky`)
        bool sticky = false;
        for (int y = yw; y < y1; y++) {
            if (sticky && ((y - yw) & 3) != 0) {
                exact_rows |= 1u << (y - yw);
                if constexpr (PF) {
#pragma unroll
                    for (int q = 0; q < NV; q++) vn[q] = vm[q];
                    load_row(min(y + 2, y1 - 1), vm);
                }
                continue;
            }
            u32x4 v[NV];
            if constexpr (PF) {
#pragma unroll
                for (int q = 0; q < NV; q++) { v[q] = vn[q]; vn[q] = vm[q]; }
                load_row(min(y + 2, y1 - 1), vm);                   // two rows in flight while this one is used
            } else {
                load_row(y, v);
            }
            uint32_t andp = 0xffffffffu;
#pragma unroll
            for (int q = 0; q < NV; q++) andp &= (v[q][0] & v[q][1]) & (v[q][2] & v[q][3]);
            v2f accr = {seed, seed}, accg = accr, accb = accr;     // .x: output 0, .y: output 1
#pragma unroll
            for (int i = 0; i < NPX; i++) {
                const uint32_t p = v[i / 4][i % 4];
                const v2f wi = WREG ? wreg[WREG ? i : 0] : s_w[i * 64 + lane];
                accr = fma_bcast(static_cast<float>(p & 0xffu), wi, accr);
                accg = fma_bcast(static_cast<float>((p >> 8) & 0xffu), wi, accg);
                accb = fma_bcast(static_cast<float>((p >> 16) & 0xffu), wi, accb);
                if (i & 1) __builtin_amdgcn_sched_barrier(0);       // two pixels' weight reads and converts in flight, not all of them
            }
            uint32_t o[HO], o2[HO];
            const v2f hr = accr + (v2f){g2, g2}, hg = accg + (v2f){g2, g2}, hb = accb + (v2f){g2, g2};
            fp32_round_toward_zero();
            o[0] = pk8(accb.x, 2, pk8(accg.x, 1, pk8(accr.x, 0, (ab & 0xffu) << 24)));
            o[1] = pk8(accb.y, 2, pk8(accg.y, 1, pk8(accr.y, 0, (ab & 0xff00u) << 16)));
            o2[0] = pk8(hb.x, 2, pk8(hg.x, 1, pk8(hr.x, 0, (ab & 0xffu) << 24)));
            o2[1] = pk8(hb.y, 2, pk8(hg.y, 1, pk8(hr.y, 0, (ab & 0xff00u) << 16)));
            fp32_round_nearest();
            const bool opaque = (andp >> 24) == 0xffu;
            uint32_t flagged = 0;                                   // bit j: output j awaits the exact recompute
            if (o[0] != o2[0]) flagged |= 1u;
            if (o[1] != o2[1]) flagged |= 2u;
            if (!opaque) flagged = 3u;                              // some alpha != 255 in the window: general arithmetic
            if (!active) flagged = 0;
            if (__popcll(__ballot(flagged != 0)) >= RG_DENSE) {     // wave-uniform: the whole row goes to the exact loop
                if (y == yw) {
                    // the wave's FIRST row is dense: all of its rows go to the exact loop at once.  On tie-dense content
                    // (the reference's ramps at an integer ratio) walking the rest of the rows here -- loading every
                    // one, trying every fourth -- was 6.4 us of a wave's 27 (per-wave timestamps) for nothing
                    exact_rows = (y1 - yw >= 32) ? 0xffffffffu : ((1u << (y1 - yw)) - 1u);
                    if (lane == 0) atomicAdd(&s_ndense, 1);
                    break;
                }
                exact_rows |= 1u << (y - yw);
                sticky = true;
                continue;
            }
            sticky = false;
            if (active) {
                uint8_t *dp = a.dst + static_cast<size_t>(y) * a.dstride + 4 * static_cast<size_t>(d0);
                if (st8 && flagged == 0) {
                    *(__attribute__((address_space(1))) u32x2 *)dp = (u32x2){o[0], o[1]};
                } else {
#pragma unroll
                    for (int j = 0; j < HO; j++)
                        if (d0 + j < a.nout) {
                            if (!((flagged >> j) & 1u)) {
                                *(g_u32w *)(dp + 4 * j) = o[j];
                            } else {
                                const int e = atomicAdd(&s_nfix, 1);
                                s_fix[e] = (static_cast<uint32_t>(y - blockIdx.y * 4 * a.rows) << 16) | static_cast<uint32_t>(d0 + j - blockIdx.x * 64 * HO);
                            }
                        }
                }
            }
        }
    }
    // ---- exact loop: resizeH's own arithmetic (resize.go:93-113) for the marked rows.  Opaque windows:
    // aw = 255 w and inv = 1 / sum aw are per column, so a row is r += R * aw (unfused, ascending taps) and
    // clampF(r * inv); other windows: resize_exact_px.  The dense window's zero weights would add +0.0 (r + R * 0.0 == r
    // exactly: r is never -0.0), so a (pixel, output) pair whose weight is zero in EVERY lane of the wave is skipped
    // outright -- wave-uniform masks, scalar branches: at a 2:1 ratio 24 of the 32 pairs are left, and two of the
    // sixteen pixels are not even converted.  The fp64 weights are staged in LDS, and only if some wave of the workgroup
    // marked a row.  (AWREG -- a lane's 2 x NPX fp64 weights in registers instead -- costs the whole kernel its
    // occupancy: 163 VGPRs at NV = 4 against 119, 111 against 80 at NV = 2; kept for experiments.)
    constexpr bool AWREG = false;
    bool any_exact = exact_rows != 0;
    if constexpr (!AWREG) {
        any_exact = __syncthreads_or(exact_rows != 0);
        if (any_exact) {
            const int g0 = blockIdx.x * 64;
            for (int e = tid; e < HO * NPX * 64; e += 256) {
                const int gl = e & 63, ji = e >> 6;
                s_aw[e] = g0 + gl < a.ngroups ? a.aw[static_cast<size_t>(ji) * a.ngroups + g0 + gl] : 0.0;
            }
            __syncthreads();
        }
    }
    if (exact_rows) {                                               // wave-uniform
        const int gc = min(g, a.ngroups - 1);
        const int s0 = a.s0[gc];
        const uint32_t ab = a.alpha[gc];
        const int d0 = gc * HO;
        const bool st8 = d0 + HO <= a.nout && ((reinterpret_cast<uintptr_t>(a.dst) | static_cast<uintptr_t>(a.dstride)) & 7u) == 0;
        auto load_row = [&](int y, u32x4 (&v)[NV]) {
            const uint8_t *row = a.src + static_cast<size_t>(y) * a.sstride;
#pragma unroll
            for (int q = 0; q < NV; q++) v[q] = *(g_u32x4 *)(row + 4 * static_cast<size_t>(s0 + 4 * q));
        };
        // the marked rows one after the other (exact_rows is wave-uniform), the next one's window in flight while
        // this one is computed: without that every row would wait out a memory latency on its own
        uint32_t todo = __builtin_amdgcn_readfirstlane(exact_rows);
        u32x4 vnx[NV];
        load_row(yw + __builtin_ctz(todo), vnx);
        const double inv0 = a.inv[d0], inv1 = a.inv[min(d0 + 1, a.nout - 1)];
        double w0[AWREG ? NPX : 1], w1[AWREG ? NPX : 1];
        if constexpr (AWREG) {
#pragma unroll
            for (int i = 0; i < NPX; i++) {
                w0[i] = a.aw[static_cast<size_t>(i) * a.ngroups + gc];
                w1[i] = a.aw[static_cast<size_t>(NPX + i) * a.ngroups + gc];
            }
        }
        uint32_t nz0 = 0, nz1 = 0;                                  // bit i: pixel i feeds output 0 / 1 in some lane of this wave
#pragma unroll
        for (int i = 0; i < NPX; i++) {
            const double x0 = AWREG ? w0[AWREG ? i : 0] : s_aw[i * 64 + lane];
            const double x1 = AWREG ? w1[AWREG ? i : 0] : s_aw[(NPX + i) * 64 + lane];
            if (__ballot(x0 != 0.0) != 0ull) nz0 |= 1u << i;
            if (__ballot(x1 != 0.0) != 0ull) nz1 |= 1u << i;
        }
        nz0 = __builtin_amdgcn_readfirstlane(nz0);
        nz1 = __builtin_amdgcn_readfirstlane(nz1);
        while (todo) {
            const int y = yw + __builtin_ctz(todo);
            todo &= todo - 1;
            u32x4 v[NV];
#pragma unroll
            for (int q = 0; q < NV; q++) v[q] = vnx[q];
            load_row(todo ? yw + __builtin_ctz(todo) : y, vnx);      // (the last row loads itself again: no branch around a load)
            uint32_t andp = 0xffffffffu;
#pragma unroll
            for (int q = 0; q < NV; q++) andp &= (v[q][0] & v[q][1]) & (v[q][2] & v[q][3]);
            uint32_t o0, o1;
            if (__all((andp >> 24) == 0xffu || !active)) {
                double r0 = 0, g0 = 0, b0 = 0, r1 = 0, g1 = 0, b1 = 0;
                // the masks are laundered once per row: otherwise the compiler hoists all 3 NPX tests out of the row
                // loop as 64-bit lane masks and spills SGPRs into VGPR lanes (two v_readlane per branch)
                uint32_t m0 = nz0, m1 = nz1;
                asm volatile("" : "+s"(m0), "+s"(m1));
                // weights one pixel ahead of their use (unconditional LDS reads: a read issued inside a branch is waited
                // for on the spot, ~100 clocks per (pixel, output) pair)
                double nx0 = AWREG ? 0.0 : s_aw[lane], nx1 = AWREG ? 0.0 : s_aw[NPX * 64 + lane];
#pragma unroll
                for (int i = 0; i < NPX; i++) {
                    const double aw0 = AWREG ? w0[AWREG ? i : 0] : nx0, aw1 = AWREG ? w1[AWREG ? i : 0] : nx1;
                    if constexpr (!AWREG)
                        if (i + 1 < NPX) { nx0 = s_aw[(i + 1) * 64 + lane]; nx1 = s_aw[(NPX + i + 1) * 64 + lane]; }
                    if (((m0 | m1) >> i) & 1u) {                         // scalar
                        const uint32_t p = v[i / 4][i % 4];
                        const double fr = u8_to_f64(p & 0xffu), fg = u8_to_f64((p >> 8) & 0xffu), fb = u8_to_f64((p >> 16) & 0xffu);
                        // (the empty asm keeps the compiler from if-converting the body into six multiply-adds + six selects)
                        if ((m0 >> i) & 1u) { asm volatile(""); r0 = r0 + fr * aw0; g0 = g0 + fg * aw0; b0 = b0 + fb * aw0; }
                        if ((m1 >> i) & 1u) { asm volatile(""); r1 = r1 + fr * aw1; g1 = g1 + fg * aw1; b1 = b1 + fb * aw1; }
                    }
                }
                o0 = clampF_fast64(r0 * inv0) | (clampF_fast64(g0 * inv0) << 8) | (clampF_fast64(b0 * inv0) << 16) | ((ab & 0xffu) << 24);
                o1 = clampF_fast64(r1 * inv1) | (clampF_fast64(g1 * inv1) << 8) | (clampF_fast64(b1 * inv1) << 16) | ((ab & 0xff00u) << 16);
            } else {
                o0 = resize_exact_px<false>(a, d0, y);
                o1 = resize_exact_px<false>(a, min(d0 + 1, a.nout - 1), y);
            }
            if (active) {
                uint8_t *dp = a.dst + static_cast<size_t>(y) * a.dstride + 4 * static_cast<size_t>(d0);
                if (st8) {
                    *(__attribute__((address_space(1))) u32x2 *)dp = (u32x2){o0, o1};
                } else {
                    *(g_u32w *)dp = o0;
                    if (d0 + 1 < a.nout) *(g_u32w *)(dp + 4) = o1;
                }
            }
        }
    }
    (void)any_exact;
    __syncthreads();
    if (tid == 0 && a.hint) a.hint[blockIdx.y * a.hint_gx + blockIdx.x] = static_cast<uint32_t>(s_ndense);
    // exact outputs for the listed ones (never stored above)
    const int nfix = s_nfix;
    const int bx0 = blockIdx.x * 64 * HO, by0 = blockIdx.y * 4 * a.rows;
    for (int e = tid; e < nfix; e += 256) {
        const int x = bx0 + static_cast<int>(s_fix[e] & 0xffffu), y = by0 + static_cast<int>(s_fix[e] >> 16);
        *(g_u32w *)(a.dst + static_cast<size_t>(y) * a.dstride + 4 * static_cast<size_t>(x)) = resize_exact_px<false>(a, x, y);
    }
}

// V pass: workgroup = 1024 columns (4 per lane: 16-byte loads and stores) x one group of VG output rows; the
// group's weights are wave-uniform (scalar loads), every source row of the union is loaded -- two rows ahead --
// and converted once for the VG rows it feeds.  Flags as in the H pass: sparse ones go on the list; an output row
// with RG_DENSE or more flagged lanes in a wave is left to the exact sweep, which walks the union again (L1 / L2)
// with fp64 accumulators for all VG rows of a column pair at a time and the dense fp64 weights aw = 255 w
// (zeros outside a row's taps add +0.0): resizeV's arithmetic for opaque columns (resize.go:137-156); columns
// with any alpha != 255 in the union call resize_exact_px.  The list holds at most (RG_DENSE - 1) lanes x 4
// columns per wave and output row.
constexpr int RG_VPX = 4;
static_assert(4 * RG_VG * RG_VPX * (RG_DENSE - 1) <= RG_FIX_CAP, "fix-up list capacity");

template <bool ALIGNED>
__global__ __launch_bounds__(256) void resize_v_guard_kernel(ResizeGuardArgs a)
{
    constexpr int VG = RG_VG, PX = RG_VPX;
    __shared__ uint32_t s_fix[RG_FIX_CAP];
    __shared__ int s_nfix;
    // the group's weights, [union row][VG]: fp32 for the guard loop, fp64 aw = 255 w for the exact sweep.  The
    // row loops read them as LDS broadcasts -- a scalar global load per row would put its miss latency (~1 us,
    // nothing to overlap it with) into every iteration.  The host keeps the union at <= 64 rows.
    __shared__ __attribute__((aligned(16))) float s_wv[64 * VG];
    __shared__ __attribute__((aligned(16))) double s_awv[(64 + 4) * VG];   // + one zero trip of the exact sweep
    __shared__ uint32_t s_tm[16 + 1];                                // per trip of 4 union rows: bit 4 k + j = row k feeds output row j
    const int tid = threadIdx.x;
    const int grp = blockIdx.y;
    if (tid == 0) s_nfix = 0;
    // every small table read is issued up front: a workgroup is short, and each dependent miss (~1 us) in its
    // prologue or epilogue would be a tenth of its life
    const int s0 = a.s0[grp], nr = a.cnt[grp];                      // wave-uniform
    uint32_t alv[VG];
    double invv[VG];
#pragma unroll
    for (int j = 0; j < VG; j++) {
        const int yy = min(grp * VG + j, a.nout - 1);
        alv[j] = a.alpha[yy];
        invv[j] = a.inv[yy];
    }
    // the H pass's verdict on the tmp rows and columns this workgroup reads (see ResizeGuardArgs::hint): lane e of every
    // wave takes one cell, two row bands x eight column blocks at most
    uint32_t cell = 0;
    int ncell = 0;
    if (a.hint) {
        const int bx0 = (blockIdx.x * 256 * PX) / (64 * RG_HO), bx1 = min(bx0 + (256 * PX) / (64 * RG_HO), a.hint_gx);
        const int by0 = min(s0 / a.hint_rows, a.hint_gy - 1), by1 = min((s0 + nr - 1) / a.hint_rows, a.hint_gy - 1);
        const int nx = bx1 - bx0;
        ncell = nx * (by1 - by0 + 1);
        const int e = tid & 63;
        if (e < ncell) cell = a.hint[(by0 + e / nx) * a.hint_gx + bx0 + e % nx];
    }
    // ... and so are the first four source rows: issued before the weights are staged and the barrier is crossed, so that
    // the two latencies overlap (idle lanes and waves read the last columns: clamped, harmless)
    const int xl = (blockIdx.x * 256 + tid) * PX;
    const int x = min(xl, a.other - PX);
    const uint8_t *col = a.src + 4 * static_cast<size_t>(x);
    auto load = [&](int s) -> u32x4 {
        const uint8_t *p = col + static_cast<size_t>(s) * a.sstride;
        if constexpr (ALIGNED) {
            return *(g_u32x4 *)p;
        } else {
            return (u32x4){*(g_u32 *)p, *(g_u32 *)(p + 4), *(g_u32 *)(p + 8), *(g_u32 *)(p + 12)};
        }
    };
    u32x4 t1 = load(s0), t2 = load(s0 + min(1, nr - 1)), t3 = load(s0 + min(2, nr - 1)), t4 = load(s0 + min(3, nr - 1));
    {
        const int nw = nr * VG;
        double awt = 0.0;
        if (tid < nw) {
            s_wv[tid] = a.dense[static_cast<size_t>(grp) * a.npx * VG + tid];
            awt = a.aw[static_cast<size_t>(grp) * a.npx * VG + tid];
        }
        s_awv[tid] = awt;                                           // past the union: zeros (the sweep walks it in fours)
        if (tid < 4 * VG) s_awv[64 * VG + tid] = 0.0;
        // which (union row, output row) pairs carry a weight at all: a wave's ballot is 16 rows = 4 trips
        const uint64_t nzb = __ballot(awt != 0.0);
        const int l = tid & 63;
        if (l < 4) s_tm[(tid >> 6) * 4 + l] = static_cast<uint32_t>(nzb >> (16 * l)) & 0xffffu;
        if (tid == 0) s_tm[16] = 0;
    }
    __syncthreads();
    const int xw = (blockIdx.x * 256 + (tid & ~63)) * PX;           // first column of this wave
    const int y0 = grp * VG;
    if (xw < a.other) {                                             // wave-uniform
        const bool active = xl < a.other;
        // a lane whose 4 columns would stick out of the image shifts left (the host keeps other >= 4): it then
        // recomputes columns its neighbour owns and stores the same values -- no branch around the loads (see
        // the H pass).  ALIGNED: src base, stride and other % 4 allow 16-byte loads at every lane.
        const int ncol = active ? PX : 0;
        const uint32_t own = (1u << ncol) - 1u;
        const bool st16 = ALIGNED && ((reinterpret_cast<uintptr_t>(a.dst) | static_cast<uintptr_t>(a.dstride)) & 15u) == 0;
        const uint32_t valid_rows = (y0 + VG <= a.nout) ? (1u << VG) - 1u : (1u << (a.nout - y0)) - 1u;
        const bool hinted = ncell > 0 && 2 * __popcll(__ballot(cell >= 2u)) >= ncell;   // wave-uniform
        uint32_t dense_rows = 0;
        if (hinted) {
            dense_rows = valid_rows;                                // every row of the group straight to the exact sweep
        } else {
            const float seed = 0.5f - a.guard, g2 = 2.0f * a.guard;
            v2f acc[VG][6];                                         // (r0,g0) (b0,r1) (g1,b1) (r2,g2) (b2,r3) (g3,b3)
#pragma unroll
            for (int j = 0; j < VG; j++)
#pragma unroll
                for (int q = 0; q < 6; q++) acc[j][q] = (v2f){seed, seed};
            u32x4 andp = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
            // four source rows in flight per lane (see the H pass: a wave's own prefetch is all it has outstanding)
            for (int i = 0; i < nr; i++) {
                const u32x4 t = t1;
                t1 = t2; t2 = t3; t3 = t4;
                t4 = load(s0 + min(i + 4, nr - 1));
                andp &= t;
                v2f f[6];
                f[0] = (v2f){static_cast<float>(t[0] & 0xffu), static_cast<float>((t[0] >> 8) & 0xffu)};
                f[1] = (v2f){static_cast<float>((t[0] >> 16) & 0xffu), static_cast<float>(t[1] & 0xffu)};
                f[2] = (v2f){static_cast<float>((t[1] >> 8) & 0xffu), static_cast<float>((t[1] >> 16) & 0xffu)};
                f[3] = (v2f){static_cast<float>(t[2] & 0xffu), static_cast<float>((t[2] >> 8) & 0xffu)};
                f[4] = (v2f){static_cast<float>((t[2] >> 16) & 0xffu), static_cast<float>(t[3] & 0xffu)};
                f[5] = (v2f){static_cast<float>((t[3] >> 8) & 0xffu), static_cast<float>((t[3] >> 16) & 0xffu)};
#pragma unroll
                for (int j = 0; j < VG; j++) {
                    const float wj = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s_wv[i * VG + j])));
#pragma unroll
                    for (int q = 0; q < 6; q++) acc[j][q] = __builtin_elementwise_fma(f[q], (v2f){wj, wj}, acc[j][q]);
                }
            }
            // windows are per output row; the union's AND is a conservative opacity test for all VG of them
            uint32_t opq = 0;                                       // bit e: column e is opaque over the union
#pragma unroll
            for (int e = 0; e < PX; e++)
                if ((andp[e] >> 24) == 0xffu) opq |= 1u << e;
#pragma unroll
            for (int j = 0; j < VG; j++) {
                const int y = y0 + j;
                if (y < a.nout) {                                   // wave-uniform
                    const uint32_t al = alv[j];
                    u32x4 o, p;
                    fp32_round_toward_zero();
                    o[0] = pk8(acc[j][1].x, 2, pk8(acc[j][0].y, 1, pk8(acc[j][0].x, 0, al)));
                    o[1] = pk8(acc[j][2].y, 2, pk8(acc[j][2].x, 1, pk8(acc[j][1].y, 0, al)));
                    o[2] = pk8(acc[j][4].x, 2, pk8(acc[j][3].y, 1, pk8(acc[j][3].x, 0, al)));
                    o[3] = pk8(acc[j][5].y, 2, pk8(acc[j][5].x, 1, pk8(acc[j][4].y, 0, al)));
                    fp32_round_nearest();
                    v2f h[6];
#pragma unroll
                    for (int q = 0; q < 6; q++) h[q] = acc[j][q] + (v2f){g2, g2};
                    fp32_round_toward_zero();
                    p[0] = pk8(h[1].x, 2, pk8(h[0].y, 1, pk8(h[0].x, 0, al)));
                    p[1] = pk8(h[2].y, 2, pk8(h[2].x, 1, pk8(h[1].y, 0, al)));
                    p[2] = pk8(h[4].x, 2, pk8(h[3].y, 1, pk8(h[3].x, 0, al)));
                    p[3] = pk8(h[5].y, 2, pk8(h[5].x, 1, pk8(h[4].y, 0, al)));
                    fp32_round_nearest();
                    uint32_t fl = ~opq;                             // bit e: column e awaits the exact recompute
#pragma unroll
                    for (int e = 0; e < PX; e++)
                        if (o[e] != p[e]) fl |= 1u << e;
                    fl &= own;
                    if (__popcll(__ballot(fl != 0)) >= RG_DENSE) {  // wave-uniform: this row goes to the exact sweep
                        dense_rows |= 1u << j;
                        continue;
                    }
                    if (active) {
                        uint8_t *dp = a.dst + static_cast<size_t>(y) * a.dstride + 4 * static_cast<size_t>(x);
                        if (st16 && fl == 0) {
                            *(g_u32x4w *)dp = o;
                        } else {
#pragma unroll
                            for (int e = 0; e < PX; e++)
                                if (e < ncol) {
                                    if (!((fl >> e) & 1u)) *(g_u32w *)(dp + 4 * e) = o[e];
                                    else s_fix[atomicAdd(&s_nfix, 1)] = (static_cast<uint32_t>(j) << 28) | static_cast<uint32_t>(x + e);   // absolute column (< 2^24)
                                }
                        }
                    }
                }
            }
        }
        if (dense_rows) {                                           // wave-uniform
            // ---- exact sweep: resizeV's own arithmetic for opaque columns (resize.go:137-156), a column pair at a time
            // (24 fp64 accumulators), four union rows per trip with the next four in flight (rows past the union are the
            // last one again: no branch around a load; their weights are zero).  A (union row, output row) pair without a
            // weight would add +0.0 (r + f * 0.0 == r exactly: r is never -0.0) and is skipped -- the weights are
            // wave-uniform, so these are scalar branches on the trip's mask: at a 2:1 ratio 12 of a group's ~19 union
            // rows feed each output row.  Output rows that are not in dense_rows are masked out the same way.
            const uint32_t rowsel = __builtin_amdgcn_readfirstlane(dense_rows) * 0x1111u;
            uint32_t opq = 0;
#pragma unroll 1
            for (int c = 0; c < PX; c += 2) {
                double r[VG][6];
#pragma unroll
                for (int j = 0; j < VG; j++)
#pragma unroll
                    for (int q = 0; q < 6; q++) r[j][q] = 0.0;
                const int c0 = min(c, max(ncol - 1, 0)), c1 = min(c + 1, max(ncol - 1, 0));
                uint32_t n0[4], n1[4];
                auto load4 = [&](int i0, uint32_t (&u0)[4], uint32_t (&u1)[4]) {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint8_t *pp = col + static_cast<size_t>(s0 + min(i0 + k, nr - 1)) * a.sstride;
                        u0[k] = *(g_u32 *)(pp + 4 * c0);
                        u1[k] = *(g_u32 *)(pp + 4 * c1);
                    }
                };
                load4(0, n0, n1);
                uint32_t and0 = 0xffffffffu, and1 = 0xffffffffu;
                uint32_t tmn = s_tm[0];
                const double2 *wrow = reinterpret_cast<const double2 *>(s_awv);
#pragma unroll 1
                for (int i = 0; i < nr; i += 4) {
                    uint32_t u0[4], u1[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) { u0[k] = n0[k]; u1[k] = n1[k]; }
                    load4(i + 4, n0, n1);
                    const uint32_t tm = __builtin_amdgcn_readfirstlane(tmn) & rowsel;
                    tmn = s_tm[(i >> 2) + 1];                       // (entry 16 is zero)
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        // the row's four weights: read before the converts that hide the LDS latency, used after them
                        const double2 wa = wrow[(i + k) * 2], wb = wrow[(i + k) * 2 + 1];
                        const double wj[VG] = {wa.x, wa.y, wb.x, wb.y};
                        const uint32_t q0 = u0[k], q1 = u1[k];
                        and0 &= q0; and1 &= q1;
                        if ((tm >> (4 * k)) & 0xfu) {               // scalar
                            const double f0 = u8_to_f64(q0 & 0xffu), f1 = u8_to_f64((q0 >> 8) & 0xffu), f2 = u8_to_f64((q0 >> 16) & 0xffu);
                            const double f3 = u8_to_f64(q1 & 0xffu), f4 = u8_to_f64((q1 >> 8) & 0xffu), f5 = u8_to_f64((q1 >> 16) & 0xffu);
#pragma unroll
                            for (int j = 0; j < VG; j++) {
                                if ((tm >> (4 * k + j)) & 1u) {     // scalar
                                    asm volatile("");               // (a real branch: not six multiply-adds + six selects)
                                    const double aw = wj[j];
                                    r[j][0] = r[j][0] + f0 * aw; r[j][1] = r[j][1] + f1 * aw; r[j][2] = r[j][2] + f2 * aw;
                                    r[j][3] = r[j][3] + f3 * aw; r[j][4] = r[j][4] + f4 * aw; r[j][5] = r[j][5] + f5 * aw;
                                }
                            }
                        }
                    }
                }
                const bool q0ok = (and0 >> 24) == 0xffu, q1ok = (and1 >> 24) == 0xffu;
                if (q0ok) opq |= 1u << c;
                if (q1ok) opq |= 2u << c;
#pragma unroll
                for (int j = 0; j < VG; j++) {
                    if (!((dense_rows >> j) & 1u) || !active) continue;
                    const uint32_t al = alv[j];
                    const double inv = invv[j];
                    uint8_t *dp = a.dst + static_cast<size_t>(y0 + j) * a.dstride + 4 * static_cast<size_t>(x);
                    const uint32_t o0 = clampF_fast64(r[j][0] * inv) | (clampF_fast64(r[j][1] * inv) << 8) | (clampF_fast64(r[j][2] * inv) << 16) | al;
                    const uint32_t o1 = clampF_fast64(r[j][3] * inv) | (clampF_fast64(r[j][4] * inv) << 8) | (clampF_fast64(r[j][5] * inv) << 16) | al;
                    if (q0ok && q1ok && c + 1 < ncol && st16) {      // (st16: 8-byte aligned as well)
                        *(__attribute__((address_space(1))) u32x2 *)(dp + 4 * c) = (u32x2){o0, o1};
                    } else {
                        if (c < ncol && q0ok) *(g_u32w *)(dp + 4 * c) = o0;
                        if (c + 1 < ncol && q1ok) *(g_u32w *)(dp + 4 * (c + 1)) = o1;
                    }
                }
            }
            // columns with some alpha != 255 in the union: the general arithmetic, once the 24 accumulators are dead
            // (inlined beside them it cost the kernel a resident wave per SIMD)
            if ((opq & own) != own) {
#pragma unroll 1
                for (int j = 0; j < VG; j++) {
                    if (!((dense_rows >> j) & 1u)) continue;
#pragma unroll 1
                    for (int e = 0; e < ncol; e++)
                        if (!((opq >> e) & 1u))
                            *(g_u32w *)(a.dst + static_cast<size_t>(y0 + j) * a.dstride + 4 * static_cast<size_t>(x + e)) = resize_exact_px<true>(a, x + e, y0 + j);
                }
            }
        }
    }
    __syncthreads();
    const int nfix = s_nfix;
    for (int e = tid; e < nfix; e += 256) {
        const int xx = static_cast<int>(s_fix[e] & 0x0fffffffu), y = y0 + static_cast<int>(s_fix[e] >> 28);
        *(g_u32w *)(a.dst + static_cast<size_t>(y) * a.dstride + 4 * static_cast<size_t>(xx)) = resize_exact_px<true>(a, xx, y);
    }
}

// ------------------------------------------------------------------------------------
// lanczosResize in ONE launch (r3): resizeH into an LDS tile, resizeV out of it (resize.go:51-52: tmp stays a uint8
// image, it just never reaches memory).
//
// Workgroup = 128 output columns (64 H groups of two: a wave's lanes) x `ng` V groups of four output rows; the tile
// holds the tmp rows those V groups read (their windows' union, <= RF_RMAX rows: the host picks ng).  Phase 1 is
// resize_h_guard_kernel's body with the tile as its destination -- a wave walks a quarter of the tile's rows, weights
// in registers, rounding guard, sparse flags on a list, dense rows in the exact loop (uniform zero-weight masks).
// Phase 2 is resize_v_guard_kernel's with the tile as its source: a wave takes a V group at a time, a lane two adjacent
// columns (8-byte LDS reads, 8-byte stores), the group's weights are wave-uniform broadcasts; rows with dense flags go
// through the exact sweep, and straight to it when half of the workgroup's H waves found their rows dense.
// Same arithmetic, same proofs as the two-pass kernels; what changes is that the tmp image is neither written nor read
// (16.6 MB each way at 4K <-> 1080p), that there is one launch, prologue and drain instead of two, and that phase 2 never
// waits for memory.  The price is the tile's row halo: phase 1 computes the union, (2 ng + 6) / 2 ng of the rows at
// 1:2, (8 ng + 12) / 8 ng at 2:1.  Windows of <= 16 pixels (NV <= 4): every upscale and downscales to about 2.3:1; the
// rest, standalone passes and tables with non-monotone windows take the two-pass kernels.
// ------------------------------------------------------------------------------------
constexpr int RF_RMAX = 64;          // tmp rows a tile can hold (32 KB)
constexpr int RF_TW = 128;           // output columns of a tile
constexpr int RF_NGMAX = 16;         // V groups per tile: bounds the fix-up list (4 rows x 14 outputs per group)
static_assert(RF_NGMAX * RG_VG * 2 * (RG_DENSE - 1) <= RG_FIX_CAP && 4 * 16 * 2 * (RG_DENSE - 1) <= RG_FIX_CAP, "fix-up list capacity");

struct ResizeFusedArgs {
    ResizeGuardArgs h;               // src = the source image, nout = dstW, other = srcH (dst unused)
    ResizeGuardArgs v;               // dst = the destination, nout = dstH, other = dstW, srcN = srcH (src unused)
    int ng;                          // V groups per tile
    const uint32_t *todo;            // resize_fused_sparse_kernel: only tiles with todo[tile] == gen run (the rest is resize_mfma_kernel's)
    uint32_t gen;
    int cells, gx;                   // tiles, tiles per row
    unsigned *counter;               // resize_mfma_kernel's workgroups that gave up (zeroed again by the reader)
    unsigned long long *report;      // host-mapped: gen << 32 | that count
    // a batch of same-geometry images (fnx_lanczos_resize_batch): device arrays of their pointers; the kernels' third grid
    // dimension is the image (resize_fused_sparse_kernel: the second), `todo` holds `cells` stamps per image
    const uint8_t *const *srcs;
    uint8_t *const *dsts;
};

// the image of this workgroup (batched launches); z = its index
__device__ __forceinline__ void resize_pick_image(ResizeFusedArgs &fa, unsigned z)
{
    if (fa.srcs) {
        fa.h.src = fa.srcs[z];
        fa.v.dst = fa.dsts[z];
    }
}

// one output of resizeV exactly as the reference computes it (resize.go:137-156), its taps read from the tile
__device__ __forceinline__ uint32_t resize_exact_px_tile(const ResizeGuardArgs &v, const uint32_t *tile, int r0, int col, int y)
{
    const int t0 = v.off[y], n = v.off[y + 1] - t0;
    const int s0 = v.idx[t0];
    double r = 0, g = 0, b = 0, al = 0;
    for (int k = 0; k < n; k++) resize_tap(tile[(s0 + k - r0) * RF_TW + col], v.wt[t0 + k], r, g, b, al);
    uint32_t o = 0;
    if (al > 0.5) {
        const double inv = 1.0 / al;
        o = clampF_dev(r * inv) | (clampF_dev(g * inv) << 8) | (clampF_dev(b * inv) << 16) | (clampF_dev(al) << 24);
    }
    return o;
}

// RMAX: the tile's rows.  64 (32 KB: three workgroups per CU) where a 2:1 downscale needs them; 32 (16 KB) for upscales,
// whose V groups read few tmp rows: with the phase buffers sized by the window (NV = 2: 13 KB) four workgroups fit a CU.
// FAST: the 2:1 forms of the exact loops (scalar weights, straight-line rows).  resize_fused_sparse_kernel, which wraps
// the tile in a loop over the tiles resize_mfma_kernel handed back, is built without them: with them it spills 39 registers.
// DENSE: no fp32 form at all -- every tmp row and every output row goes straight to the exact loops (which are the
// reference's arithmetic for any content: always right, slow where the fp32 form would have been proven).  For a plan whose
// recent images came back tie-dense from the matrix kernel (its cool-down, see resize_fused): the kernel then carries
// neither the fp32 weights nor the guard's packs, and its first rows are not computed twice.
template <int NV, int RMAX, bool FAST = true, bool DENSE = false>
__device__ __forceinline__ void resize_fused_tile(const ResizeFusedArgs &fa, const int bx, const int by)
{
    constexpr int HO = RG_HO, NPX = 4 * NV, VG = RG_VG;
    static_assert(NV <= 4 && HO == 2, "weights in registers; two outputs per lane");
    __shared__ __attribute__((aligned(16))) uint32_t s_tile[RMAX * RF_TW];
    __shared__ uint32_t s_fix[RG_FIX_CAP];
    __shared__ int s_nfix, s_nfix2, s_ndense;
    // phase 1: fp64 aw [HO][NPX][64] of the exact loop.  phase 2, per wave: fp32 weights [64][VG], fp64 aw [64 + 4][VG],
    // trip masks [16 + 4]
    constexpr size_t U_H = sizeof(double) * HO * NPX * 64, U_V = 4 * (sizeof(float) * 64 * VG + sizeof(double) * 68 * VG + sizeof(uint32_t) * 20);
    __shared__ __attribute__((aligned(16))) unsigned char s_u[(U_H > U_V ? U_H : U_V + 15) & ~size_t(15)];
    const ResizeGuardArgs &a = fa.h;
    const ResizeGuardArgs &v = fa.v;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) { s_nfix = 0; s_nfix2 = 0; s_ndense = 0; }
    // the tile's tmp rows [r0, r1): the union of its V groups' windows (monotone tables: the host checks)
    const int gfirst = by * fa.ng, glast = min(gfirst + fa.ng, v.ngroups) - 1;
    const int r0 = v.s0[gfirst], r1 = v.s0[glast] + v.cnt[glast];
    const int rpw = (r1 - r0 + 3) >> 2;
    const int yw = r0 + wave * rpw, y1 = min(yw + rpw, r1);        // this wave's rows (<= 16)
    const int g = bx * 64 + lane;
    const int gc = min(g, a.ngroups - 1);                           // idle lanes shadow the last group (their columns are never stored)
    const bool active = g < a.ngroups;
    const int s0 = a.s0[gc];
    const uint32_t ab = a.alpha[gc];
    const int d0 = gc * HO;
    uint32_t *trow = s_tile + 2 * lane;                             // + (y - r0) * RF_TW: this lane's two tmp pixels of row y
    // phase 2's weights travel one V group ahead of their use, the first group's from here (a dependent miss at the
    // start of phase 2 would have nothing to hide behind): lane l holds entries l, l + 64, ... of the group's table
    double pre_aw[5];
    float pre_w[4];
    auto fetch_weights = [&](int gi) {
        const int gq = min(gi, v.ngroups - 1);
        const int nrq = v.cnt[gq] * VG;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int idx = min(lane + 64 * k, v.npx * VG - 1);
            pre_aw[k] = v.aw[static_cast<size_t>(gq) * v.npx * VG + idx];
            if (lane + 64 * k >= nrq) pre_aw[k] = 0.0;
            if (k < 4) {
                pre_w[k] = v.dense[static_cast<size_t>(gq) * v.npx * VG + idx];
                if (lane + 64 * k >= nrq) pre_w[k] = 0.0f;
            }
        }
    };
    constexpr bool PRE = RMAX > 32;      // (the 32-row form has no registers to carry them through phase 1: it fetches per group)
    if constexpr (PRE) fetch_weights(gfirst + wave);
    __syncthreads();                                                // the counters are in place
    uint32_t exact_rows = 0;                                        // bit r: row yw + r of this wave awaits the exact loop
    auto load_row = [&](int y, u32x4 (&w)[NV]) {
        const uint8_t *row = a.src + static_cast<size_t>(y) * a.sstride;
#pragma unroll
        for (int q = 0; q < NV; q++) w[q] = *(g_u32x4 *)(row + 4 * static_cast<size_t>(s0 + 4 * q));
    };
    // ---------------- phase 1: resizeH (resize.go:77-117) into the tile ----------------
    if constexpr (DENSE) {
        if (yw < y1) exact_rows = (1u << (y1 - yw)) - 1u;
        if (tid == 0) s_ndense = 4;
    }
    if (!DENSE && yw < y1) {                                        // wave-uniform
        u32x4 vn[NV], vm[NV];
        load_row(yw, vn);
        load_row(min(yw + 1, y1 - 1), vm);
        v2f wreg[NPX];
#pragma unroll
        for (int i = 0; i < NPX; i++)
            wreg[i] = (v2f){a.dense[static_cast<size_t>(i) * a.ngroups + gc], a.dense[static_cast<size_t>(NPX + i) * a.ngroups + gc]};
        const float seed = 0.5f - a.guard, g2 = 2.0f * a.guard;
        bool sticky = false;
        for (int y = yw; y < y1; y++) {
            if (sticky && ((y - yw) & 3) != 0) {
                exact_rows |= 1u << (y - yw);
#pragma unroll
                for (int q = 0; q < NV; q++) vn[q] = vm[q];
                load_row(min(y + 2, y1 - 1), vm);
                continue;
            }
            u32x4 w[NV];
#pragma unroll
            for (int q = 0; q < NV; q++) { w[q] = vn[q]; vn[q] = vm[q]; }
            load_row(min(y + 2, y1 - 1), vm);                       // two rows in flight while this one is used
            uint32_t andp = 0xffffffffu;
#pragma unroll
            for (int q = 0; q < NV; q++) andp &= (w[q][0] & w[q][1]) & (w[q][2] & w[q][3]);
            v2f accr = {seed, seed}, accg = accr, accb = accr;     // .x: output 0, .y: output 1
#pragma unroll
            for (int i = 0; i < NPX; i++) {
                const uint32_t p = w[i / 4][i % 4];
                accr = fma_bcast(static_cast<float>(p & 0xffu), wreg[i], accr);
                accg = fma_bcast(static_cast<float>((p >> 8) & 0xffu), wreg[i], accg);
                accb = fma_bcast(static_cast<float>((p >> 16) & 0xffu), wreg[i], accb);
                if (i & 1) __builtin_amdgcn_sched_barrier(0);
            }
            uint32_t o[HO], o2[HO];
            const v2f hr = accr + (v2f){g2, g2}, hg = accg + (v2f){g2, g2}, hb = accb + (v2f){g2, g2};
            fp32_round_toward_zero();
            o[0] = pk8(accb.x, 2, pk8(accg.x, 1, pk8(accr.x, 0, (ab & 0xffu) << 24)));
            o[1] = pk8(accb.y, 2, pk8(accg.y, 1, pk8(accr.y, 0, (ab & 0xff00u) << 16)));
            o2[0] = pk8(hb.x, 2, pk8(hg.x, 1, pk8(hr.x, 0, (ab & 0xffu) << 24)));
            o2[1] = pk8(hb.y, 2, pk8(hg.y, 1, pk8(hr.y, 0, (ab & 0xff00u) << 16)));
            fp32_round_nearest();
            uint32_t flagged = 0;                                   // bit j: output j awaits the exact recompute
            if (o[0] != o2[0]) flagged |= 1u;
            if (o[1] != o2[1]) flagged |= 2u;
            if ((andp >> 24) != 0xffu) flagged = 3u;                // some alpha != 255 in the window: general arithmetic
            if (!active) flagged = 0;
            if (d0 + 1 >= a.nout) flagged &= 1u;                    // (an odd width's last group has one output)
            if (__popcll(__ballot(flagged != 0)) >= RG_DENSE) {     // wave-uniform: the whole row goes to the exact loop
                if (y == yw) {                                      // the wave's FIRST row is dense: all of its rows at once
                    exact_rows = (1u << (y1 - yw)) - 1u;
                    if (lane == 0) atomicAdd(&s_ndense, 1);
                    break;
                }
                exact_rows |= 1u << (y - yw);
                sticky = true;
                continue;
            }
            sticky = false;
            uint32_t *tp = trow + (y - r0) * RF_TW;
            if (!(flagged & 1u)) tp[0] = o[0];
            else s_fix[atomicAdd(&s_nfix, 1)] = (static_cast<uint32_t>(y - r0) << 8) | static_cast<uint32_t>(2 * lane);
            if (!(flagged & 2u)) tp[1] = o[1];
            else s_fix[atomicAdd(&s_nfix, 1)] = (static_cast<uint32_t>(y - r0) << 8) | static_cast<uint32_t>(2 * lane + 1);
        }
    }
    // the exact loop of resize_h_guard_kernel (see there): fp64 weights staged only if some wave marked a row
    double *s_aw = reinterpret_cast<double *>(s_u);
    if (__syncthreads_or(exact_rows != 0)) {
        const int g0 = bx * 64;
        for (int e = tid; e < HO * NPX * 64; e += 256) {
            const int gl = e & 63, ji = e >> 6;
            s_aw[e] = g0 + gl < a.ngroups ? a.aw[static_cast<size_t>(ji) * a.ngroups + g0 + gl] : 0.0;
        }
        __syncthreads();
        if (exact_rows) {                                           // wave-uniform
            uint32_t todo = __builtin_amdgcn_readfirstlane(exact_rows);
            u32x4 vnx[NV];
            load_row(yw + __builtin_ctz(todo), vnx);
            const double inv0 = a.inv[d0], inv1 = a.inv[min(d0 + 1, a.nout - 1)];
            uint32_t nz0 = 0, nz1 = 0;                              // bit i: pixel i feeds output 0 / 1 in some lane of this wave
#pragma unroll
            for (int i = 0; i < NPX; i++) {
                if (__ballot(s_aw[i * 64 + lane] != 0.0) != 0ull) nz0 |= 1u << i;
                if (__ballot(s_aw[(NPX + i) * 64 + lane] != 0.0) != 0ull) nz1 |= 1u << i;
            }
            nz0 = __builtin_amdgcn_readfirstlane(nz0);
            nz1 = __builtin_amdgcn_readfirstlane(nz1);
            // The 2:1 form (NV == 4).  At an exact 2:1 ratio every output has the SAME twelve weights (its centre lies half way
            // between two source pixels) and output 1 of a lane starts two pixels after output 0: W[i] for pixels 0..11, W[i - 2]
            // for pixels 2..13, whatever the lane -- bar the few outputs at the image's edges whose tap lists are clamped.  A wave
            // whose lanes agree (checked here against the staged table, bit for bit) runs its rows as straight-line code: the
            // weights are scalar operands, no LDS read, no branch per tap, and the loop's loads run two rows ahead (the masked
            // form below is a chain of small blocks the scheduler cannot overlap: PMC had it waiting 42 % of its wave-cycles).
            // Lanes that disagree put their outputs on the workgroup's fix-up list (resize_exact_px, spread over all lanes).
            constexpr int UT = 12, USH = 2;
            bool uni = false;
            uint64_t oddm = 0;
            double W[UT] = {};
            if constexpr (NV == 4 && FAST) {
                const uint64_t am = __ballot(active);
                const int ref = __builtin_amdgcn_readfirstlane(__popcll(am) >> 1);
#pragma unroll
                for (int i = 0; i < UT; i++) {
                    const double t = s_aw[i * 64 + ref];             // (one address for the wave: a broadcast)
                    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(__double2loint(t)));
                    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(__double2hiint(t)));
                    W[i] = __hiloint2double(static_cast<int>(hi), static_cast<int>(lo));
                }
                bool same = true;
#pragma unroll
                for (int i = 0; i < NPX; i++) {
                    const double e0 = i < UT ? W[i] : 0.0, e1 = (i >= USH && i < USH + UT) ? W[i - USH] : 0.0;
                    same = same && s_aw[i * 64 + lane] == e0 && s_aw[(NPX + i) * 64 + lane] == e1;
                }
                oddm = __ballot(!same && active);
                uni = __popcll(oddm) <= 6 && W[0] != 0.0;
            }
            // Four outputs per lane, two rows per wave instruction: lanes 0..31 take one row, lanes 32..63 the next, each lane
            // the outputs of TWO adjacent groups (20 source pixels, of which 18 carry weight).  A source pixel is then
            // converted once per four outputs instead of once per two: 108 + 288 instructions per 4 outputs against
            // 2 x (84 + 144).  Needs the second group's window to start four pixels after the first (it does wherever the
            // weights are uniform) and 20 readable pixels from the first window's start.
            bool quad = false;
            int s0q = 0;
            if constexpr (NV == 4 && FAST) {
                if (uni) {
                    const int hl = lane & 31, gq = min(bx * 64 + 2 * hl, a.ngroups - 1), gq1 = min(gq + 1, a.ngroups - 1);
                    s0q = a.s0[gq];
                    const bool lane_odd = ((oddm >> (2 * hl)) & 3ull) != 0ull;
                    // (a lane with an edge group in it is recomputed from the list: its 20 pixels need not all exist -- the loads clamp)
                    const bool fits = lane_odd || bx * 64 + 2 * hl >= a.ngroups ||      // (lanes past the last group store nothing)
                                      (s0q + 20 <= a.srcN && (bx * 64 + 2 * hl + 1 >= a.ngroups || a.s0[gq1] == s0q + 4));
                    quad = __all(fits) && a.srcN >= 20;
                }
            }
            if (quad) {                                             // wave-uniform
                const int hl = lane & 31, half = lane >> 5;
                const int gA = bx * 64 + 2 * hl;
                const bool actA = gA < a.ngroups, actB = gA + 1 < a.ngroups;
                const bool oddA = (oddm >> (2 * hl)) & 1ull, oddB = (oddm >> (2 * hl + 1)) & 1ull;
                const int dq = HO * min(gA, a.ngroups - 1);          // first output column of this lane (HO outputs per group)
                double invq[4];
#pragma unroll
                for (int j = 0; j < 4; j++) invq[j] = a.inv[min(dq + j, a.nout - 1)];
                const uint32_t abA = a.alpha[min(gA, a.ngroups - 1)], abB = a.alpha[min(gA + 1, a.ngroups - 1)];
                auto load20 = [&](int y, u32x4 (&w)[5]) {
                    const uint8_t *row = a.src + static_cast<size_t>(y) * a.sstride;
#pragma unroll
                    for (int q = 0; q < 5; q++) w[q] = *(g_u32x4 *)(row + 4 * static_cast<size_t>(min(s0q + 4 * q, a.srcN - 4)));
                };
                // pairs of rows: (first set bit, second set bit) of what is left; a lone last row runs with its upper half idle
                uint32_t tl = todo;
                auto next_pair = [&](int &ylo, int &yhi, bool &has_hi) {
                    if (!tl) { has_hi = false; return false; }
                    ylo = yw + __builtin_ctz(tl);
                    tl &= tl - 1;
                    has_hi = tl != 0;
                    yhi = has_hi ? yw + __builtin_ctz(tl) : ylo;
                    if (has_hi) tl &= tl - 1;
                    return true;
                };
                int yl0 = yw, yh0 = yw, yl1 = yw, yh1 = yw;
                bool hh0 = false, hh1 = false;
                u32x4 ra[5];                                        // the next pair's pixels (one pair = two rows ahead)
                bool more0 = next_pair(yl0, yh0, hh0);
                load20(half ? yh0 : yl0, ra);
                uint32_t redo = 0;
                while (more0) {
                    const int y = half ? yh0 : yl0;
                    const bool mine = half ? hh0 : true;             // (a lone row: the upper half computes it again and stores nothing)
                    const int yl_now = yl0, yh_now = yh0;
                    const bool hh_now = hh0;
                    u32x4 w[5];
#pragma unroll
                    for (int q = 0; q < 5; q++) w[q] = ra[q];
                    const bool more1 = next_pair(yl1, yh1, hh1);
                    if (more1) { yl0 = yl1; yh0 = yh1; hh0 = hh1; }
                    more0 = more1;
                    load20(half ? yh0 : yl0, ra);                   // (after the last pair: the same rows again, unused)
                    uint32_t andp = 0xffffffffu;
#pragma unroll
                    for (int q = 0; q < 5; q++) andp &= (w[q][0] & w[q][1]) & (w[q][2] & w[q][3]);
                    uint32_t o[4];
                    if (!__all((andp >> 24) == 0xffu || !actA)) {   // a window that is not opaque: both rows go to the two-output form below
                        redo |= 1u << (yl_now - yw);
                        if (hh_now) redo |= 1u << (yh_now - yw);
                        continue;
                    }
                    {
                        double acc[4][3];
#pragma unroll
                        for (int j = 0; j < 4; j++) acc[j][0] = acc[j][1] = acc[j][2] = 0.0;
#pragma unroll
                        for (int i = 0; i < 3 * USH + UT; i++) {
                            const uint32_t p = w[i / 4][i % 4];
                            const double fr = u8_to_f64(p & 0xffu), fg = u8_to_f64((p >> 8) & 0xffu), fb = u8_to_f64((p >> 16) & 0xffu);
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const int t = i - USH * j;
                                if (t >= 0 && t < UT) {
                                    acc[j][0] = acc[j][0] + fr * W[t]; acc[j][1] = acc[j][1] + fg * W[t]; acc[j][2] = acc[j][2] + fb * W[t];
                                }
                            }
                        }
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const uint32_t al = j < 2 ? ((abA >> (8 * j)) & 0xffu) : ((abB >> (8 * (j - 2))) & 0xffu);
                            o[j] = clampF_fast64(acc[j][0] * invq[j]) | (clampF_fast64(acc[j][1] * invq[j]) << 8) |
                                   (clampF_fast64(acc[j][2] * invq[j]) << 16) | (al << 24);
                        }
                    }
                    if (mine) {
                        uint32_t *tp = s_tile + (y - r0) * RF_TW + 4 * hl;
                        if (!oddA && !oddB && actB) {
                            *reinterpret_cast<u32x4 *>(tp) = (u32x4){o[0], o[1], o[2], o[3]};
                        } else {
                            // (edge groups, <= 6 of the wave's: their outputs go on the fix-up list -- at most 12 entries per
                            // row, two rows per pass, within the list's bound of 14 per wave-row)
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const bool act = j < 2 ? actA : actB, od = j < 2 ? oddA : oddB;
                                if (!act) continue;
                                if (!od) tp[j] = o[j];
                                else if (dq + j < a.nout) s_fix[atomicAdd(&s_nfix, 1)] = (static_cast<uint32_t>(y - r0) << 8) | static_cast<uint32_t>(4 * hl + j);
                            }
                        }
                    }
                }
                todo = redo;
            }
            if (uni && todo) {                                      // wave-uniform
                const bool odd = (oddm >> lane) & 1ull;
                uint32_t tl = todo;                                 // rows whose loads have not been issued
                auto next_row = [&](int fallback) {
                    if (!tl) return fallback;
                    const int yy = yw + __builtin_ctz(tl);
                    tl &= tl - 1;
                    return yy;
                };
                u32x4 ra[NV], rb[NV];
                const int ya = next_row(yw);
                load_row(ya, ra);
                const int yb = next_row(ya);
                load_row(yb, rb);
                int ylast = yb;
                while (todo) {
                    const int y = yw + __builtin_ctz(todo);
                    todo &= todo - 1;
                    u32x4 w[NV];
#pragma unroll
                    for (int q = 0; q < NV; q++) { w[q] = ra[q]; ra[q] = rb[q]; }
                    ylast = next_row(ylast);
                    load_row(ylast, rb);
                    uint32_t andp = 0xffffffffu;
#pragma unroll
                    for (int q = 0; q < NV; q++) andp &= (w[q][0] & w[q][1]) & (w[q][2] & w[q][3]);
                    uint32_t o0, o1;
                    if (__all((andp >> 24) == 0xffu || !active)) {
                        double rr0 = 0, gg0 = 0, bb0 = 0, rr1 = 0, gg1 = 0, bb1 = 0;
#pragma unroll
                        for (int i = 0; i < USH + UT; i++) {
                            const uint32_t p = w[i / 4][i % 4];
                            const double fr = u8_to_f64(p & 0xffu), fg = u8_to_f64((p >> 8) & 0xffu), fb = u8_to_f64((p >> 16) & 0xffu);
                            if (i < UT) { rr0 = rr0 + fr * W[i]; gg0 = gg0 + fg * W[i]; bb0 = bb0 + fb * W[i]; }
                            if (i >= USH) { rr1 = rr1 + fr * W[i - USH]; gg1 = gg1 + fg * W[i - USH]; bb1 = bb1 + fb * W[i - USH]; }
                        }
                        o0 = clampF_fast64(rr0 * inv0) | (clampF_fast64(gg0 * inv0) << 8) | (clampF_fast64(bb0 * inv0) << 16) | ((ab & 0xffu) << 24);
                        o1 = clampF_fast64(rr1 * inv1) | (clampF_fast64(gg1 * inv1) << 8) | (clampF_fast64(bb1 * inv1) << 16) | ((ab & 0xff00u) << 16);
                    } else {
                        o0 = resize_exact_px<false>(a, d0, y);
                        o1 = resize_exact_px<false>(a, min(d0 + 1, a.nout - 1), y);
                    }
                    uint32_t *tp = trow + (y - r0) * RF_TW;
                    if (!odd) {
                        tp[0] = o0;
                        tp[1] = o1;
                    } else {                                        // (<= 6 lanes: within the list's bound of 7 lanes x 2 outputs per wave-row)
                        s_fix[atomicAdd(&s_nfix, 1)] = (static_cast<uint32_t>(y - r0) << 8) | static_cast<uint32_t>(2 * lane);
                        if (d0 + 1 < a.nout) s_fix[atomicAdd(&s_nfix, 1)] = (static_cast<uint32_t>(y - r0) << 8) | static_cast<uint32_t>(2 * lane + 1);
                    }
                }
            }
            while (todo) {
                const int y = yw + __builtin_ctz(todo);
                todo &= todo - 1;
                u32x4 w[NV];
#pragma unroll
                for (int q = 0; q < NV; q++) w[q] = vnx[q];
                load_row(todo ? yw + __builtin_ctz(todo) : y, vnx);
                uint32_t andp = 0xffffffffu;
#pragma unroll
                for (int q = 0; q < NV; q++) andp &= (w[q][0] & w[q][1]) & (w[q][2] & w[q][3]);
                uint32_t o0, o1;
                if (__all((andp >> 24) == 0xffu || !active)) {
                    double rr0 = 0, gg0 = 0, bb0 = 0, rr1 = 0, gg1 = 0, bb1 = 0;
                    uint32_t m0 = nz0, m1 = nz1;
                    asm volatile("" : "+s"(m0), "+s"(m1));
                    double nx0 = s_aw[lane], nx1 = s_aw[NPX * 64 + lane];
#pragma unroll
                    for (int i = 0; i < NPX; i++) {
                        const double aw0 = nx0, aw1 = nx1;
                        if (i + 1 < NPX) { nx0 = s_aw[(i + 1) * 64 + lane]; nx1 = s_aw[(NPX + i + 1) * 64 + lane]; }
                        if (((m0 | m1) >> i) & 1u) {                // scalar
                            const uint32_t p = w[i / 4][i % 4];
                            const double fr = u8_to_f64(p & 0xffu), fg = u8_to_f64((p >> 8) & 0xffu), fb = u8_to_f64((p >> 16) & 0xffu);
                            if ((m0 >> i) & 1u) { asm volatile(""); rr0 = rr0 + fr * aw0; gg0 = gg0 + fg * aw0; bb0 = bb0 + fb * aw0; }
                            if ((m1 >> i) & 1u) { asm volatile(""); rr1 = rr1 + fr * aw1; gg1 = gg1 + fg * aw1; bb1 = bb1 + fb * aw1; }
                        }
                    }
                    o0 = clampF_fast64(rr0 * inv0) | (clampF_fast64(gg0 * inv0) << 8) | (clampF_fast64(bb0 * inv0) << 16) | ((ab & 0xffu) << 24);
                    o1 = clampF_fast64(rr1 * inv1) | (clampF_fast64(gg1 * inv1) << 8) | (clampF_fast64(bb1 * inv1) << 16) | ((ab & 0xff00u) << 16);
                } else {
                    o0 = resize_exact_px<false>(a, d0, y);
                    o1 = resize_exact_px<false>(a, min(d0 + 1, a.nout - 1), y);
                }
                uint32_t *tp = trow + (y - r0) * RF_TW;
                tp[0] = o0;
                tp[1] = o1;
            }
        }
    }
    __syncthreads();
    {   // exact tmp pixels for the listed ones
        const int nfix = s_nfix;
        for (int e = tid; e < nfix; e += 256) {
            const int row = static_cast<int>(s_fix[e] >> 8), col = static_cast<int>(s_fix[e] & 0xffu);
            s_tile[row * RF_TW + col] = resize_exact_px<false>(a, bx * RF_TW + col, r0 + row);
        }
    }
    __syncthreads();
    const bool hinted = DENSE || s_ndense >= 2;                     // wave-uniform

    // ---------------- phase 2: resizeV (resize.go:120-160) out of the tile ----------------
    float *s_wv = reinterpret_cast<float *>(s_u) + wave * (64 * VG);
    double *s_awv = reinterpret_cast<double *>(s_u + 4 * sizeof(float) * 64 * VG) + wave * (68 * VG);
    uint32_t *s_tm = reinterpret_cast<uint32_t *>(s_u + 4 * (sizeof(float) * 64 * VG + sizeof(double) * 68 * VG)) + wave * 20;
    const int x = bx * RF_TW + 2 * lane;
    const int ncol = x + 1 < v.other ? 2 : (x < v.other ? 1 : 0);
    const uint32_t own = (1u << ncol) - 1u;
    const bool st8 = ((reinterpret_cast<uintptr_t>(v.dst) | static_cast<uintptr_t>(v.dstride)) & 7u) == 0;
    const uint32_t *tcol = s_tile + 2 * lane;
    typedef __attribute__((address_space(1))) u32x2 g_u32x2w;
    for (int gi = gfirst + wave; gi <= glast; gi += 4) {            // wave-uniform
        const int sg = v.s0[gi], nr = v.cnt[gi];
        const int y0 = gi * VG;
        uint32_t alv[VG];
        double invv[VG];
#pragma unroll
        for (int j = 0; j < VG; j++) {
            const int yy = min(y0 + j, v.nout - 1);
            alv[j] = v.alpha[yy];
            invv[j] = v.inv[yy];
        }
        // the group's weights into this wave's corner of the LDS (previous group's reads have all returned: in-order)
        __builtin_amdgcn_wave_barrier();
        if constexpr (!PRE) fetch_weights(gi);
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int idx = lane + 64 * k;
            const double awt = pre_aw[k];
            if (idx < 68 * VG) s_awv[idx] = awt;
            if (k < 4) s_wv[idx] = pre_w[k];
            const uint64_t nzb = __ballot(awt != 0.0);
            if (lane < 4) s_tm[4 * k + lane] = static_cast<uint32_t>(nzb >> (16 * lane)) & 0xffffu;
        }
        if constexpr (PRE) fetch_weights(gi + 4);                   // the next group's, in flight while this one is computed
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t *tg = tcol + (sg - r0) * RF_TW;              // the group's first union row, this lane's two columns
        const uint32_t valid_rows = (y0 + VG <= v.nout) ? (1u << VG) - 1u : (1u << (v.nout - y0)) - 1u;
        uint32_t dense_rows = 0;
        if (hinted) {
            dense_rows = valid_rows;
        } else {
            const float seed = 0.5f - v.guard, g2 = 2.0f * v.guard;
            v2f acc[VG][3];                                         // (r0,g0) (b0,r1) (g1,b1)
#pragma unroll
            for (int j = 0; j < VG; j++) acc[j][0] = acc[j][1] = acc[j][2] = (v2f){seed, seed};
            u32x2 andp = {0xffffffffu, 0xffffffffu};
            u32x2 t1 = *reinterpret_cast<const u32x2 *>(tg), t2 = *reinterpret_cast<const u32x2 *>(tg + min(1, nr - 1) * RF_TW);
            for (int i = 0; i < nr; i++) {
                const u32x2 t = t1;
                t1 = t2;
                t2 = *reinterpret_cast<const u32x2 *>(tg + min(i + 2, nr - 1) * RF_TW);
                andp &= t;
                v2f f[3];
                f[0] = (v2f){static_cast<float>(t[0] & 0xffu), static_cast<float>((t[0] >> 8) & 0xffu)};
                f[1] = (v2f){static_cast<float>((t[0] >> 16) & 0xffu), static_cast<float>(t[1] & 0xffu)};
                f[2] = (v2f){static_cast<float>((t[1] >> 8) & 0xffu), static_cast<float>((t[1] >> 16) & 0xffu)};
#pragma unroll
                for (int j = 0; j < VG; j++) {
                    const float wj = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s_wv[i * VG + j])));
#pragma unroll
                    for (int q = 0; q < 3; q++) acc[j][q] = __builtin_elementwise_fma(f[q], (v2f){wj, wj}, acc[j][q]);
                }
            }
            uint32_t opq = 0;                                       // bit e: column e is opaque over the union
            if ((andp[0] >> 24) == 0xffu) opq |= 1u;
            if ((andp[1] >> 24) == 0xffu) opq |= 2u;
#pragma unroll
            for (int j = 0; j < VG; j++) {
                const int y = y0 + j;
                if (y < v.nout) {                                   // wave-uniform
                    const uint32_t al = alv[j];
                    u32x2 o, p;
                    fp32_round_toward_zero();
                    o[0] = pk8(acc[j][1].x, 2, pk8(acc[j][0].y, 1, pk8(acc[j][0].x, 0, al)));
                    o[1] = pk8(acc[j][2].y, 2, pk8(acc[j][2].x, 1, pk8(acc[j][1].y, 0, al)));
                    fp32_round_nearest();
                    v2f hh[3];
#pragma unroll
                    for (int q = 0; q < 3; q++) hh[q] = acc[j][q] + (v2f){g2, g2};
                    fp32_round_toward_zero();
                    p[0] = pk8(hh[1].x, 2, pk8(hh[0].y, 1, pk8(hh[0].x, 0, al)));
                    p[1] = pk8(hh[2].y, 2, pk8(hh[2].x, 1, pk8(hh[1].y, 0, al)));
                    fp32_round_nearest();
                    uint32_t fl = ~opq;                             // bit e: column e awaits the exact recompute
                    if (o[0] != p[0]) fl |= 1u;
                    if (o[1] != p[1]) fl |= 2u;
                    fl &= own;
                    if (__popcll(__ballot(fl != 0)) >= RG_DENSE) {  // wave-uniform: this row goes to the exact sweep
                        dense_rows |= 1u << j;
                        continue;
                    }
                    if (ncol) {
                        uint8_t *dp = v.dst + static_cast<size_t>(y) * v.dstride + 4 * static_cast<size_t>(x);
                        if (st8 && ncol == 2 && fl == 0) {
                            *(g_u32x2w *)dp = o;
                        } else {
#pragma unroll
                            for (int e = 0; e < 2; e++)
                                if (e < ncol) {
                                    if (!((fl >> e) & 1u)) *(g_u32w *)(dp + 4 * e) = o[e];
                                    else s_fix[atomicAdd(&s_nfix2, 1)] = (static_cast<uint32_t>(y) << 8) | static_cast<uint32_t>(2 * lane + e);
                                }
                        }
                    }
                }
            }
        }
        if (dense_rows) {                                           // wave-uniform: the exact sweep (see resize_v_guard_kernel)
            const uint32_t rowsel = __builtin_amdgcn_readfirstlane(dense_rows) * 0x1111u;
            double r[VG][6];
#pragma unroll
            for (int j = 0; j < VG; j++)
#pragma unroll
                for (int q = 0; q < 6; q++) r[j][q] = 0.0;
            uint32_t and0 = 0xffffffffu, and1 = 0xffffffffu;
            uint32_t tmn = s_tm[0];
            const double2 *wrow = reinterpret_cast<const double2 *>(s_awv);
            // The 2:1 form of the sweep (see phase 1): the group's four outputs have the same twelve weights, each two tmp rows
            // further down -- 18 union rows, entry (i, j) = W[i - 2 j].  Checked against the staged table, then straight-line
            // code with scalar weights (no masks, no weight reads).  Groups at the image's top and bottom keep the masked form.
            constexpr int UT = 12, USH = 2;
            bool uni = false;
            if constexpr (NV == 4 && FAST) {
                if (nr == USH * (VG - 1) + UT && dense_rows == (1u << VG) - 1u) {   // wave-uniform
                    bool same = true;
#pragma unroll
                    for (int k = 0; k < 5; k++) {
                        const int idx = lane + 64 * k, i = idx >> 2, j = idx & 3, t = i - USH * j;
                        if (idx < 68 * VG) {
                            const double want = (t >= 0 && t < UT && i < nr) ? s_awv[t * VG] : 0.0;
                            same = same && s_awv[idx] == want;
                        }
                    }
                    uni = __all(same) && s_awv[0] != 0.0;
                }
            }
            if (uni) {                                              // wave-uniform
                double W[UT];
#pragma unroll
                for (int t = 0; t < UT; t++) {
                    const double wv = s_awv[t * VG];
                    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(__double2loint(wv)));
                    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(__double2hiint(wv)));
                    W[t] = __hiloint2double(static_cast<int>(hi), static_cast<int>(lo));
                }
#pragma unroll
                for (int i = 0; i < USH * (VG - 1) + UT; i++) {
                    const u32x2 u = *reinterpret_cast<const u32x2 *>(tg + i * RF_TW);
                    const uint32_t q0 = u[0], q1 = u[1];
                    and0 &= q0; and1 &= q1;
                    const double f0 = u8_to_f64(q0 & 0xffu), f1 = u8_to_f64((q0 >> 8) & 0xffu), f2 = u8_to_f64((q0 >> 16) & 0xffu);
                    const double f3 = u8_to_f64(q1 & 0xffu), f4 = u8_to_f64((q1 >> 8) & 0xffu), f5 = u8_to_f64((q1 >> 16) & 0xffu);
#pragma unroll
                    for (int j = 0; j < VG; j++) {
                        const int t = i - USH * j;
                        if (t >= 0 && t < UT) {
                            const double aw = W[t];
                            r[j][0] = r[j][0] + f0 * aw; r[j][1] = r[j][1] + f1 * aw; r[j][2] = r[j][2] + f2 * aw;
                            r[j][3] = r[j][3] + f3 * aw; r[j][4] = r[j][4] + f4 * aw; r[j][5] = r[j][5] + f5 * aw;
                        }
                    }
                }
            }
#pragma unroll 1
            for (int i = uni ? nr : 0; i < nr; i += 4) {
                u32x2 u[4];
#pragma unroll
                for (int k = 0; k < 4; k++) u[k] = *reinterpret_cast<const u32x2 *>(tg + min(i + k, nr - 1) * RF_TW);
                const uint32_t tm = __builtin_amdgcn_readfirstlane(tmn) & rowsel;
                tmn = s_tm[(i >> 2) + 1];                           // (entries 16..19 are zero)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double2 wa = wrow[(i + k) * 2], wb = wrow[(i + k) * 2 + 1];
                    const double wj[VG] = {wa.x, wa.y, wb.x, wb.y};
                    const uint32_t q0 = u[k][0], q1 = u[k][1];
                    and0 &= q0; and1 &= q1;
                    if ((tm >> (4 * k)) & 0xfu) {                   // scalar
                        const double f0 = u8_to_f64(q0 & 0xffu), f1 = u8_to_f64((q0 >> 8) & 0xffu), f2 = u8_to_f64((q0 >> 16) & 0xffu);
                        const double f3 = u8_to_f64(q1 & 0xffu), f4 = u8_to_f64((q1 >> 8) & 0xffu), f5 = u8_to_f64((q1 >> 16) & 0xffu);
#pragma unroll
                        for (int j = 0; j < VG; j++) {
                            if ((tm >> (4 * k + j)) & 1u) {         // scalar
                                asm volatile("");
                                const double aw = wj[j];
                                r[j][0] = r[j][0] + f0 * aw; r[j][1] = r[j][1] + f1 * aw; r[j][2] = r[j][2] + f2 * aw;
                                r[j][3] = r[j][3] + f3 * aw; r[j][4] = r[j][4] + f4 * aw; r[j][5] = r[j][5] + f5 * aw;
                            }
                        }
                    }
                }
            }
            const bool q0ok = (and0 >> 24) == 0xffu, q1ok = (and1 >> 24) == 0xffu;
#pragma unroll
            for (int j = 0; j < VG; j++) {
                if (!((dense_rows >> j) & 1u) || !ncol) continue;
                const uint32_t al = alv[j];
                const double inv = invv[j];
                const int y = y0 + j;
                uint8_t *dp = v.dst + static_cast<size_t>(y) * v.dstride + 4 * static_cast<size_t>(x);
                const uint32_t o0 = clampF_fast64(r[j][0] * inv) | (clampF_fast64(r[j][1] * inv) << 8) | (clampF_fast64(r[j][2] * inv) << 16) | al;
                const uint32_t o1 = clampF_fast64(r[j][3] * inv) | (clampF_fast64(r[j][4] * inv) << 8) | (clampF_fast64(r[j][5] * inv) << 16) | al;
                if (q0ok && q1ok && ncol == 2 && st8) {
                    *(g_u32x2w *)dp = (u32x2){o0, o1};
                } else {
                    if (q0ok) *(g_u32w *)dp = o0;
                    else *(g_u32w *)dp = resize_exact_px_tile(v, s_tile, r0, 2 * lane, y);
                    if (ncol == 2) {
                        if (q1ok) *(g_u32w *)(dp + 4) = o1;
                        else *(g_u32w *)(dp + 4) = resize_exact_px_tile(v, s_tile, r0, 2 * lane + 1, y);
                    }
                }
            }
        }
    }
    __syncthreads();
    {   // exact outputs for the listed ones (never stored above)
        const int nfix = s_nfix2;
        for (int e = tid; e < nfix; e += 256) {
            const int y = static_cast<int>(s_fix[e] >> 8), col = static_cast<int>(s_fix[e] & 0xffu);
            *(g_u32w *)(v.dst + static_cast<size_t>(y) * v.dstride + 4 * static_cast<size_t>(bx * RF_TW + col)) =
                resize_exact_px_tile(v, s_tile, r0, col, y);
        }
    }
}

template <int NV, int RMAX>
__global__ __launch_bounds__(256, RMAX <= 32 ? 4 : 3) void resize_fused_kernel(ResizeFusedArgs fa)
{
    resize_pick_image(fa, blockIdx.z);
    resize_fused_tile<NV, RMAX>(fa, blockIdx.x, blockIdx.y);
}

template <int NV, int RMAX>
__global__ __launch_bounds__(256, 3) void resize_fused_dense_kernel(ResizeFusedArgs fa)
{
    resize_pick_image(fa, blockIdx.z);
    resize_fused_tile<NV, RMAX, true, true>(fa, blockIdx.x, blockIdx.y);
}

// ------------------------------------------------------------------------------------
// resize_dense21_kernel (r5): the exact 2:1 downscale of tie-dense content, and nothing else
// ------------------------------------------------------------------------------------
// What resize_fused_dense_kernel does for a plan in its tie-dense cool-down, as a kernel of its own: the straight-line forms
// of the exact loops (four outputs per lane and two rows per wave instruction in the H pass, the 18-row sweep in the V pass)
// with the twelve weights, 1 / sum and the alpha byte as kernel ARGUMENTS (the host has checked that every output but the
// image's edge groups shares them: build_dense21) -- no weight tables, no staging, no fp32 form, no lists.  What does not
// fit the form is recomputed per pixel from the tap lists, in the reference's order (resize_exact_px / _tile): the outputs
// of "odd" groups (clamped tap lists at the image's edges) and every output whose window holds a pixel that is not opaque.
// Tiles of up to 74 rows (eight V groups: 1.16 rows of H work per row used where the general kernel's 64 rows do 1.33, and two
// groups for each wave), 42 KB of LDS, three workgroups per CU (four: twelve registers spilled, 47 us against 45).
constexpr int D21_ROWS = 74, D21_NGMAX = 8;
struct Dense21Args {
    double wh[12], wv[12];           // aw = 255 w of the shared tap lists
    double inv_h, inv_v;             // 1 / sum aw
    uint32_t al_h, al_v;             // clampF(sum aw): the alpha byte of an opaque window
    int base_h, base_v;              // first tap of output d: 2 d + base
    const uint8_t *odd_h, *odd_v;    // per H group (2 outputs) / V group (4 outputs): 1 = not of the form
    unsigned long long *heavy;       // host-mapped: a workgroup that redid more than an eighth of its tile per pixel leaves `gen`
    unsigned gen;
};

#ifndef FNX_D21_OCC
#define FNX_D21_OCC 3
#endif
__global__ __launch_bounds__(256, FNX_D21_OCC) void resize_dense21_kernel(ResizeFusedArgs fa, Dense21Args dn)
{
    constexpr int VG = RG_VG, UT = 12, USH = 2;
    __shared__ __attribute__((aligned(16))) uint32_t s_tile[D21_ROWS * RF_TW];
    __shared__ uint16_t s_list[D21_ROWS * 32];                       // (row << 5 | lane): four outputs that await the per-pixel form
    __shared__ unsigned s_nlist;
    resize_pick_image(fa, blockIdx.z);
    const ResizeGuardArgs &a = fa.h;
    const ResizeGuardArgs &v = fa.v;
    const int bx = blockIdx.x, by = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) s_nlist = 0;
    const int gfirst = by * fa.ng, glast = min(gfirst + fa.ng, v.ngroups) - 1;
    const int r0 = v.s0[gfirst], r1 = v.s0[glast] + v.cnt[glast];   // the tile's tmp rows (monotone tables: the host checks)
    const int rpw = (r1 - r0 + 3) >> 2;
    const int yw = r0 + wave * rpw, y1 = min(yw + rpw, r1);          // this wave's rows (<= 19)
    __syncthreads();
    // ---------------- phase 1: resizeH (resize.go:77-117), four outputs per lane, two rows per pass ----------------
    {
        const int hl = lane & 31, half = lane >> 5;
        const int d0 = bx * RF_TW + 4 * hl;                          // this lane's first output column
        const bool act = d0 < a.nout;
        const int sx0 = 2 * d0 + dn.base_h;                          // the first of this lane's 18 tap pixels (20 are loaded)
        const bool lane_odd = act && (dn.odd_h[min(d0 >> 1, a.ngroups - 1)] | dn.odd_h[min((d0 >> 1) + 1, a.ngroups)] | (d0 + 3 >= a.nout) |
                                      (sx0 < 0) | (sx0 + 20 > a.srcN));
        const int sx = clampi(sx0, 0, a.srcN - 20);                  // (an odd lane's 20 pixels need not be its own: it is redone)
        auto load20 = [&](int y, u32x4 (&w)[5]) {
            const uint8_t *row = a.src + static_cast<size_t>(y) * a.sstride + 4 * static_cast<size_t>(sx);
#pragma unroll
            for (int q = 0; q < 5; q++) w[q] = *(g_u32x4 *)(row + 16 * q);
        };
        const int npairs = (y1 - yw + 1) >> 1;                       // wave-uniform; <= 10
        u32x4 ra[5];                                                 // the next pair's pixels (two pairs ahead: 12 registers spilled)
        auto row_of = [&](int k) { return min(yw + 2 * min(k, max(npairs - 1, 0)) + half, max(y1 - 1, yw)); };
        if (npairs > 0) load20(row_of(0), ra);
        for (int k = 0; k < npairs; k++) {
            const int y = yw + 2 * k + half;
            const bool mine = y < y1;                                // (a lone last row: the upper half computes the row before it again)
            u32x4 w[5];
#pragma unroll
            for (int q = 0; q < 5; q++) w[q] = ra[q];
            load20(row_of(k + 1), ra);                               // one pair (two rows) ahead
            uint32_t andp = 0xffffffffu;
#pragma unroll
            for (int q = 0; q < 5; q++) andp &= (w[q][0] & w[q][1]) & (w[q][2] & w[q][3]);
            double acc[4][3];
#pragma unroll
            for (int j = 0; j < 4; j++) acc[j][0] = acc[j][1] = acc[j][2] = 0.0;
#pragma unroll
            for (int i = 0; i < 3 * USH + UT; i++) {
                const uint32_t p = w[i / 4][i % 4];
                const double fr = u8_to_f64(p & 0xffu), fg = u8_to_f64((p >> 8) & 0xffu), fb = u8_to_f64((p >> 16) & 0xffu);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int t = i - USH * j;
                    if (t >= 0 && t < UT) {
                        acc[j][0] = acc[j][0] + fr * dn.wh[t]; acc[j][1] = acc[j][1] + fg * dn.wh[t]; acc[j][2] = acc[j][2] + fb * dn.wh[t];
                    }
                }
            }
            u32x4 o;
#pragma unroll
            for (int j = 0; j < 4; j++)
                o[j] = clampF_fast64(acc[j][0] * dn.inv_h) | (clampF_fast64(acc[j][1] * dn.inv_h) << 8) |
                       (clampF_fast64(acc[j][2] * dn.inv_h) << 16) | (dn.al_h << 24);
            if (mine && act) {
                *reinterpret_cast<u32x4 *>(s_tile + (min(y, y1 - 1) - r0) * RF_TW + 4 * hl) = o;
                if (lane_odd || (andp >> 24) != 0xffu) s_list[atomicAdd(&s_nlist, 1u)] = static_cast<uint16_t>(((y - r0) << 5) | hl);
            }
        }
    }
    __syncthreads();
    // the per-pixel form for what phase 1 could not take, one output per thread (a thread per lane's four took 70 us on the
    // image's left and right tiles, where one lane of every row is on the list)
    // (translucent content is all list: the host is told, and gives the plan's next calls to the general kernel)
    if (tid == 0 && s_nlist > 4u * static_cast<unsigned>(r1 - r0))
        __hip_atomic_store(dn.heavy, static_cast<unsigned long long>(dn.gen), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (unsigned e = tid, n = 4 * s_nlist; e < n; e += 256) {
        const unsigned ent = s_list[e >> 2];
        const int row = ent >> 5, col = 4 * (ent & 31u) + (e & 3u);
        if (bx * RF_TW + col < a.nout) s_tile[row * RF_TW + col] = resize_exact_px<false>(a, bx * RF_TW + col, r0 + row);
    }
    __syncthreads();
    // ---------------- phase 2: resizeV (resize.go:120-160) out of the tile ----------------
    const int x = bx * RF_TW + 2 * lane;
    const int ncol = x + 1 < v.other ? 2 : (x < v.other ? 1 : 0);
    const bool st8 = ((reinterpret_cast<uintptr_t>(v.dst) | static_cast<uintptr_t>(v.dstride)) & 7u) == 0;
    typedef __attribute__((address_space(1))) u32x2 g_u32x2w;
    for (int gi = gfirst + wave; gi <= glast; gi += 4) {            // wave-uniform
        const int y0 = gi * VG, sg = v.s0[gi];
        const bool form = !dn.odd_v[gi] && v.cnt[gi] == USH * (VG - 1) + UT && y0 + VG <= v.nout && sg == 2 * y0 + dn.base_v;   // wave-uniform
        uint32_t and0 = 0xffffffffu, and1 = 0xffffffffu;
        double r[VG][6];
        if (form) {
#pragma unroll
            for (int j = 0; j < VG; j++)
#pragma unroll
                for (int q = 0; q < 6; q++) r[j][q] = 0.0;
            const uint32_t *tg = s_tile + (sg - r0) * RF_TW + 2 * lane;
#pragma unroll
            for (int i = 0; i < USH * (VG - 1) + UT; i++) {
                const u32x2 u = *reinterpret_cast<const u32x2 *>(tg + i * RF_TW);
                const uint32_t q0 = u[0], q1 = u[1];
                and0 &= q0; and1 &= q1;
                const double f0 = u8_to_f64(q0 & 0xffu), f1 = u8_to_f64((q0 >> 8) & 0xffu), f2 = u8_to_f64((q0 >> 16) & 0xffu);
                const double f3 = u8_to_f64(q1 & 0xffu), f4 = u8_to_f64((q1 >> 8) & 0xffu), f5 = u8_to_f64((q1 >> 16) & 0xffu);
#pragma unroll
                for (int j = 0; j < VG; j++) {
                    const int t = i - USH * j;
                    if (t >= 0 && t < UT) {
                        const double aw = dn.wv[t];
                        r[j][0] = r[j][0] + f0 * aw; r[j][1] = r[j][1] + f1 * aw; r[j][2] = r[j][2] + f2 * aw;
                        r[j][3] = r[j][3] + f3 * aw; r[j][4] = r[j][4] + f4 * aw; r[j][5] = r[j][5] + f5 * aw;
                    }
                }
            }
        }
        const bool q0ok = form && (and0 >> 24) == 0xffu, q1ok = form && (and1 >> 24) == 0xffu;
#pragma unroll
        for (int j = 0; j < VG; j++) {
            const int y = y0 + j;
            if (y >= v.nout || !ncol) continue;
            uint8_t *dp = v.dst + static_cast<size_t>(y) * v.dstride + 4 * static_cast<size_t>(x);
            uint32_t o0 = 0, o1 = 0;
            if (form) {
                o0 = clampF_fast64(r[j][0] * dn.inv_v) | (clampF_fast64(r[j][1] * dn.inv_v) << 8) | (clampF_fast64(r[j][2] * dn.inv_v) << 16) | (dn.al_v << 24);
                o1 = clampF_fast64(r[j][3] * dn.inv_v) | (clampF_fast64(r[j][4] * dn.inv_v) << 8) | (clampF_fast64(r[j][5] * dn.inv_v) << 16) | (dn.al_v << 24);
            }
            if (q0ok && q1ok && ncol == 2 && st8) {
                *(g_u32x2w *)dp = (u32x2){o0, o1};
            } else {
                *(g_u32w *)dp = q0ok ? o0 : resize_exact_px_tile(v, s_tile, r0, 2 * lane, y);
                if (ncol == 2) *(g_u32w *)(dp + 4) = q1ok ? o1 : resize_exact_px_tile(v, s_tile, r0, 2 * lane + 1, y);
            }
        }
    }
}

// After resize_mfma_kernel: the tiles it handed back, and only those.  A fixed grid of workgroups walks the stamps (a launch
// of every tile's workgroup that looks and leaves took 4 us of an 18 us call); the first one also passes on to the host
// how many of the matrix kernel's workgroups gave up (a hint for the next call with these tables: resize_fused's cool-down).
template <int NV, int RMAX>
__global__ __launch_bounds__(256, RMAX <= 32 ? 4 : 3) void resize_fused_sparse_kernel(ResizeFusedArgs fa)
{
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        // how many of resize_mfma_kernel's workgroups gave up (it has finished: stream order), for the host's next call
        const unsigned total = atomicExch(fa.counter, 0u);
        *reinterpret_cast<volatile unsigned long long *>(fa.report) = (static_cast<unsigned long long>(fa.gen) << 32) | total;
    }
    resize_pick_image(fa, blockIdx.y);
    const uint32_t *todo = fa.todo + static_cast<size_t>(blockIdx.y) * fa.cells;        // (a batch: `cells` stamps per image)
    for (int t = blockIdx.x; t < fa.cells; t += gridDim.x) {
        if (todo[t] != fa.gen) continue;
        resize_fused_tile<NV, RMAX, false>(fa, t % fa.gx, t / fa.gx);
        __syncthreads();
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

}  // namespace fnx

// ------------------------------------------------------------------------------------
// resize plans: what one tap table turns into on the device, built once per (table, direction) and ctx
// ------------------------------------------------------------------------------------
struct fnx_resize_plan {
    // key
    uint64_t id = 0;                 // != 0: an immutable table of host_api's cache (no content compare)
    bool vertical = false;
    int nout = 0, srcN = 0;
    std::vector<int32_t> k_off, k_idx;   // id == 0: copy of the caller's table, compared on every call
    std::vector<double> k_wt;
    uint64_t last_use = 0;
    // device blob
    void *blob = nullptr;
    const int32_t *d_off = nullptr, *d_idx = nullptr;
    const double *d_wt = nullptr;
    int contig_taps = 0;
    // guard form
    bool guard_ok = false;
    int HO = 0, NV = 0, ngroups = 0, npx = 0;
    unsigned d21_fit = 0;            // vertical plans: bit ng = tiles of ng V groups hold at most D21_ROWS rows (resize_dense21_kernel)
    bool d21_heavy = false;          // H plans: the last resize_dense21_kernel call met translucent content (until the matrix kernel's next try)
    uint32_t d21_last_gen = 0;
    int fused_ng = 0, fused_ng32 = 0;   // vertical plans: V groups per 64-row / 32-row tile of resize_fused_kernel (0: not eligible)
    float guard = 0;
    const float *d_dense = nullptr;
    const int32_t *d_s0 = nullptr, *d_cnt = nullptr;
    const uint32_t *d_alpha = nullptr;
    const double *d_aw = nullptr, *d_inv = nullptr;
    fnx::RzMfTable mf;               // resize_mfma.hip: the matrix form (mf.ok: covered)
    // resize_dense21_kernel (exact 2:1): the twelve weights aw = 255 w every interior output shares, 1 / sum aw, the alpha byte,
    // and one flag per group (H: 2 outputs, V: 4) that says "not of that form" (clamped tap lists at the image's edges)
    bool d21_ok = false;
    double d21_w[12] = {}, d21_inv = 0;
    uint32_t d21_alpha = 0;
    int d21_base = 0;                // first tap of output d: 2 d + base
    const uint8_t *d_odd = nullptr;
    int mf_cool = 0;                 // H plans: calls left to skip the matrix kernel (it handed most of an image back)
    int mf_cool_len = 32;            // half the length of the next such pause
};

namespace fnx {

void free_resize_plans(fnx_ctx *ctx)
{
    for (fnx_resize_plan *p : ctx->rplans) {
        if (p->blob) (void)hipFree(p->blob);
        resize_mfma_free(&p->mf);
        delete p;
    }
    ctx->rplans.clear();
    if (ctx->rz_todo) (void)hipFree(ctx->rz_todo);
    ctx->rz_todo = nullptr;
    ctx->rz_todo_cap = 0;
    if (ctx->rz_report) (void)hipHostFree(ctx->rz_report);
    ctx->rz_report = nullptr;
    ctx->rz_last_gen = 0;
}

static bool resize_guard_disabled(const fnx_ctx *ctx)
{
    const char *e = form_value(ctx, FORM_RESIZE_FP64);      // A/B and tests: "1" keeps the round-1 fp64 kernels
    return e && e[0] == '1';
}

// Host side of the guard form.  Returns false when the table is outside what the guard kernels cover
// (gaps in the tap lists, windows too wide for the register matrix, a <= 0.5 somewhere, wild weights).
// The 2:1 form of a plan (resize_dense21_kernel): is there ONE list of twelve weights that the outputs share, each output two
// source pixels after the one before?  W, 1 / sum and the alpha byte are taken from the middle output; `odd[g]` marks the
// groups (H: 2 outputs, V: 4) with an output that differs in any bit or lies elsewhere.  ok: at least three quarters match.
static void build_dense21(const TapTable &t, int group, const std::vector<double> &invv, const std::vector<uint32_t> &abyte,
                          fnx_resize_plan &p, std::vector<uint8_t> &odd)
{
    p.d21_ok = false;
    odd.clear();
    const int nout = t.nout, ng = (nout + group - 1) / group, dm = nout / 2;
    if (nout < 2 * group || t.off[dm + 1] - t.off[dm] != 12) return;
    const int base = t.idx[t.off[dm]] - 2 * dm;                     // first tap of output d: 2 d + base
    p.d21_base = base;
    for (int k = 0; k < 12; k++) p.d21_w[k] = 255.0 * t.wt[t.off[dm] + k];
    p.d21_inv = invv[dm];
    p.d21_alpha = abyte[dm];
    if (p.d21_w[0] == 0.0) return;
    auto like = [&](int d) {
        if (d >= nout) return false;
        const int t0 = t.off[d];
        if (t.off[d + 1] - t0 != 12 || t.idx[t0] != 2 * d + base || invv[d] != p.d21_inv || abyte[d] != p.d21_alpha) return false;
        for (int k = 0; k < 12; k++)
            if (255.0 * t.wt[t0 + k] != p.d21_w[k]) return false;
        return true;
    };
    odd.assign(ng, 0);
    int nodd = 0;
    for (int g = 0; g < ng; g++) {
        bool all = true;
        for (int j = 0; j < group; j++) all = all && like(g * group + j);
        odd[g] = all ? 0 : 1;
        nodd += odd[g];
    }
    p.d21_ok = 4 * nodd <= ng;
}

static bool build_guard(const TapTable &t, int srcN, bool vertical, fnx_resize_plan &p, std::vector<float> &dense,
                        std::vector<int32_t> &s0v, std::vector<int32_t> &cntv, std::vector<uint32_t> &alphav,
                        std::vector<double> &awv, std::vector<double> &invv, std::vector<uint8_t> &odd)
{
    const int nout = t.nout;
    if (p.contig_taps <= 0) return false;
    // per output: start, count, normalised fp32 weights, alpha byte
    double smax = 0;
    std::vector<uint32_t> abyte(nout);
    invv.assign(nout, 0.0);
    for (int d = 0; d < nout; d++) {
        const int t0 = t.off[d], n = t.off[d + 1] - t0;
        if (n < 1) return false;
        double a = 0;
        for (int k = 0; k < n; k++) {
            const double aw = 255.0 * t.wt[t0 + k];                 // sa * w with sa = 255 (resize.go:95-96)
            if (!std::isfinite(aw)) return false;
            a += aw;
        }
        if (!(a > 0.5) || !std::isfinite(a)) return false;          // resize.go:107
        const double inv = 1.0 / a;
        invv[d] = inv;
        double sabs = 0;
        for (int k = 0; k < n; k++) sabs += std::fabs(255.0 * t.wt[t0 + k] * inv);
        if (sabs > smax) smax = sabs;
        double r = std::trunc(a);                                    // clampF(a)
        if (std::fabs(a - r) >= 0.5) r += 1.0;
        abyte[d] = static_cast<uint32_t>(std::fmin(std::fmax(r, 0.0), 255.0));
    }
    if (!(smax < 8.0)) return false;
    auto wnorm = [&](int d, int k) {                                 // W_k = aw_k * inv in fp64, then fp32
        const int t0 = t.off[d], n = t.off[d + 1] - t0;
        double a = 0;
        for (int q = 0; q < n; q++) a += 255.0 * t.wt[t0 + q];
        return static_cast<float>(255.0 * t.wt[t0 + k] * (1.0 / a));
    };
    int nfma = 0;
    if (!vertical) {
        // groups of HO adjacent outputs, union window of 4 NV pixels (NV <= 8: 64 weights per lane)
        constexpr int HO = RG_HO;
        int need = 0;
        for (int d0 = 0; d0 < nout; d0 += HO) {
            const int s0 = t.idx[t.off[d0]];
            int end = s0;
            for (int j = 0; j < HO && d0 + j < nout; j++) {
                const int tj = t.off[d0 + j], nj = t.off[d0 + j + 1] - tj;
                if (t.idx[tj] < s0) return false;                    // windows must not start before the group's first
                end = std::max(end, t.idx[tj] + nj);
            }
            need = std::max(need, end - s0);
        }
        int NV = std::max(2, (need + 3) / 4);
        if (NV > 8 || srcN < 4 * NV) return false;                   // (tiny sources keep the fp64 kernels)
        const int NPX = 4 * NV, ng = (nout + HO - 1) / HO;
        dense.assign(static_cast<size_t>(HO) * NPX * ng, 0.0f);
        awv.assign(static_cast<size_t>(HO) * NPX * ng, 0.0);
        s0v.assign(ng, 0);
        alphav.assign(ng, 0);
        for (int g = 0; g < ng; g++) {
            // the kernels always read 4 NV pixels from s0: a window that would stick out of the row starts earlier
            // (its extra leading pixels get zero weights)
            const int d0 = g * HO, s0 = std::min(t.idx[t.off[d0]], srcN - NPX);
            s0v[g] = s0;
            for (int j = 0; j < HO && d0 + j < nout; j++) {
                const int tj = t.off[d0 + j], nj = t.off[d0 + j + 1] - tj;
                for (int k = 0; k < nj; k++) {
                    dense[static_cast<size_t>(j * NPX + (t.idx[tj] - s0 + k)) * ng + g] = wnorm(d0 + j, k);
                    awv[static_cast<size_t>(j * NPX + (t.idx[tj] - s0 + k)) * ng + g] = 255.0 * t.wt[tj + k];
                }
                alphav[g] |= abyte[d0 + j] << (8 * j);
            }
        }
        p.HO = HO; p.NV = NV; p.ngroups = ng; p.npx = NPX;
        nfma = NPX;
    } else {
        constexpr int VG = RG_VG;
        const int ng = (nout + VG - 1) / VG;
        int need = 0;
        s0v.assign(ng, 0);
        cntv.assign(ng, 0);
        for (int g = 0; g < ng; g++) {
            const int d0 = g * VG;
            int s0 = 1 << 30, end = 0;
            for (int j = 0; j < VG && d0 + j < nout; j++) {
                const int tj = t.off[d0 + j], nj = t.off[d0 + j + 1] - tj;
                s0 = std::min(s0, t.idx[tj]);
                end = std::max(end, t.idx[tj] + nj);
            }
            s0v[g] = s0;
            cntv[g] = end - s0;
            need = std::max(need, end - s0);
        }
        if (need > 64) return false;                                 // a lane per union row (resize_v_guard_kernel)
        dense.assign(static_cast<size_t>(ng) * need * VG, 0.0f);
        awv.assign(static_cast<size_t>(ng) * need * VG, 0.0);
        for (int g = 0; g < ng; g++)
            for (int j = 0; j < VG && g * VG + j < nout; j++) {
                const int d = g * VG + j, tj = t.off[d], nj = t.off[d + 1] - tj;
                for (int k = 0; k < nj; k++) {
                    dense[(static_cast<size_t>(g) * need + (t.idx[tj] - s0v[g] + k)) * VG + j] = wnorm(d, k);
                    awv[(static_cast<size_t>(g) * need + (t.idx[tj] - s0v[g] + k)) * VG + j] = 255.0 * t.wt[tj + k];
                }
            }
        alphav.resize(nout);
        for (int d = 0; d < nout; d++) alphav[d] = abyte[d] << 24;
        p.HO = VG; p.NV = 0; p.ngroups = ng; p.npx = need;
        nfma = need;
        // resize_fused_kernel: a tile's tmp rows are [s0 of its first group, end of its last) -- windows must move down
        // monotonically -- and at most RF_RMAX of them; the largest group count that fits every tile
        bool mono = true;
        for (int g = 1; g < ng; g++)
            if (s0v[g] < s0v[g - 1] || s0v[g] + cntv[g] < s0v[g - 1] + cntv[g - 1]) mono = false;
        auto groups_that_fit = [&](int rmax) {
            for (int cand = RF_NGMAX; mono && cand >= 1; cand--) {
                bool fits = true;
                for (int g0 = 0; g0 < ng && fits; g0 += cand) {
                    const int gl = std::min(g0 + cand, ng) - 1;
                    if (s0v[gl] + cntv[gl] - s0v[g0] > rmax) fits = false;
                }
                if (fits) return cand;
            }
            return 0;
        };
        p.fused_ng = groups_that_fit(RF_RMAX);
        p.fused_ng32 = groups_that_fit(32);
        p.d21_fit = 0;
        for (int cand = 1; mono && cand <= D21_NGMAX; cand++) {
            bool fits = true;
            for (int g0 = 0; g0 < ng && fits; g0 += cand) {
                const int gl = std::min(g0 + cand, ng) - 1;
                if (s0v[gl] + cntv[gl] - s0v[g0] > D21_ROWS) fits = false;
            }
            if (fits) p.d21_fit |= 1u << cand;
        }
    }
    // the bound of the header comment
    const double top = 255.0 * smax + 1.0;
    const double half_ulp = std::ldexp(1.0, static_cast<int>(std::ceil(std::log2(top))) - 24);
    const double E = 255.0 * smax * std::ldexp(1.0, -24) + (nfma + 1) * half_ulp;
    const double G = E + half_ulp + 1e-6;
    if (!(G < 0.05)) return false;
    float gf = static_cast<float>(G);
    if (static_cast<double>(gf) < G) gf = std::nextafterf(gf, 1.0f);
    p.guard = gf;
    (void)srcN;
    build_dense21(t, vertical ? RG_VG : RG_HO, invv, abyte, p, odd);
    return true;
}

static int get_resize_plan(fnx_ctx *ctx, const TapTable &t, int srcN, bool vertical, fnx_resize_plan **out)
{
    FNX_REQUIRE(t.off && t.idx && t.wt && t.nout > 0, "tap table is null");
    const int ntaps = t.off[t.nout];
    FNX_REQUIRE(ntaps >= 0, "tap table offsets");
    static std::atomic<uint64_t> tick{1};
    for (fnx_resize_plan *p : ctx->rplans) {
        if (p->vertical != vertical || p->nout != t.nout || p->srcN != srcN) continue;
        bool same;
        if (t.id != 0 || p->id != 0) {
            same = t.id == p->id;
        } else {
            same = static_cast<int>(p->k_idx.size()) == ntaps &&
                   std::memcmp(p->k_off.data(), t.off, sizeof(int32_t) * (t.nout + 1)) == 0 &&
                   std::memcmp(p->k_idx.data(), t.idx, sizeof(int32_t) * ntaps) == 0 &&
                   std::memcmp(p->k_wt.data(), t.wt, sizeof(double) * ntaps) == 0;
        }
        if (same) {
            p->last_use = tick++;
            *out = p;
            return FNX_OK;
        }
    }
    // indices must address the source (the kernels do not check per tap)
    for (int i = 0; i < ntaps; i++) FNX_REQUIRE(t.idx[i] >= 0 && t.idx[i] < srcN, "tap index outside the source");
    for (int d = 0; d < t.nout; d++) FNX_REQUIRE(t.off[d + 1] >= t.off[d], "tap table offsets");
    if (ctx->rplans.size() >= 8) {                                   // evict the least recently used
        size_t victim = 0;
        for (size_t i = 1; i < ctx->rplans.size(); i++)
            if (ctx->rplans[i]->last_use < ctx->rplans[victim]->last_use) victim = i;
        FNX_HIP(hipStreamSynchronize(ctx->stream));                  // queued kernels may still read its tables
        if (ctx->rplans[victim]->blob) FNX_HIP(hipFree(ctx->rplans[victim]->blob));
        resize_mfma_free(&ctx->rplans[victim]->mf);
        delete ctx->rplans[victim];
        ctx->rplans.erase(ctx->rplans.begin() + victim);
    }
    std::unique_ptr<fnx_resize_plan> p(new fnx_resize_plan());
    p->id = t.id; p->vertical = vertical; p->nout = t.nout; p->srcN = srcN;
    if (t.id == 0) {
        p->k_off.assign(t.off, t.off + t.nout + 1);
        p->k_idx.assign(t.idx, t.idx + ntaps);
        p->k_wt.assign(t.wt, t.wt + ntaps);
    }
    p->contig_taps = resize_contiguous_taps(t.off, t.idx, t.nout);
    std::vector<float> dense;
    std::vector<int32_t> s0v, cntv;
    std::vector<uint32_t> alphav;
    std::vector<double> awv, invv;
    std::vector<uint8_t> odd;
    p->guard_ok = build_guard(t, srcN, vertical, *p, dense, s0v, cntv, alphav, awv, invv, odd);
    if (!p->guard_ok) { dense.clear(); s0v.clear(); cntv.clear(); alphav.clear(); awv.clear(); invv.clear(); odd.clear(); p->d21_ok = false; }
    // one blob: wt | off | idx | dense | s0 | cnt | alpha | aw | inv, each 16-byte aligned
    auto al16 = [](size_t n) { return (n + 15) & ~size_t(15); };
    const size_t b_wt = al16(sizeof(double) * std::max(ntaps, 1)), b_off = al16(sizeof(int32_t) * (t.nout + 1)),
                 b_idx = al16(sizeof(int32_t) * std::max(ntaps, 1)), b_dense = al16(sizeof(float) * dense.size()),
                 b_s0 = al16(sizeof(int32_t) * s0v.size()), b_cnt = al16(sizeof(int32_t) * cntv.size()),
                 b_alpha = al16(sizeof(uint32_t) * alphav.size()), b_aw = al16(sizeof(double) * awv.size()),
                 b_inv = al16(sizeof(double) * invv.size()), b_odd = al16(odd.size() + 64);   // (+ 64: lanes past the last group read on)
    const size_t total = b_wt + b_off + b_idx + b_dense + b_s0 + b_cnt + b_alpha + b_aw + b_inv + b_odd + 16;
    std::vector<unsigned char> host(total, 0);
    size_t o = 0;
    std::memcpy(host.data() + o, t.wt, sizeof(double) * ntaps); const size_t o_wt = o; o += b_wt;
    std::memcpy(host.data() + o, t.off, sizeof(int32_t) * (t.nout + 1)); const size_t o_off = o; o += b_off;
    std::memcpy(host.data() + o, t.idx, sizeof(int32_t) * ntaps); const size_t o_idx = o; o += b_idx;
    if (!dense.empty()) std::memcpy(host.data() + o, dense.data(), sizeof(float) * dense.size());
    const size_t o_dense = o; o += b_dense;
    if (!s0v.empty()) std::memcpy(host.data() + o, s0v.data(), sizeof(int32_t) * s0v.size());
    const size_t o_s0 = o; o += b_s0;
    if (!cntv.empty()) std::memcpy(host.data() + o, cntv.data(), sizeof(int32_t) * cntv.size());
    const size_t o_cnt = o; o += b_cnt;
    if (!alphav.empty()) std::memcpy(host.data() + o, alphav.data(), sizeof(uint32_t) * alphav.size());
    const size_t o_alpha = o; o += b_alpha;
    if (!awv.empty()) std::memcpy(host.data() + o, awv.data(), sizeof(double) * awv.size());
    const size_t o_aw = o; o += b_aw;
    if (!invv.empty()) std::memcpy(host.data() + o, invv.data(), sizeof(double) * invv.size());
    const size_t o_inv = o; o += b_inv;
    std::memset(host.data() + o, 1, b_odd);                          // (what lies past the last group is "odd")
    if (!odd.empty()) std::memcpy(host.data() + o, odd.data(), odd.size());
    const size_t o_odd = o;
    FNX_HIP(hipMalloc(&p->blob, total));
    // synchronous copy: the host vector dies with this call (plans are built once per table and ctx)
    hipError_t e = hipMemcpy(p->blob, host.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(p->blob);
        set_error("hipMemcpy of a resize plan failed: %s", hipGetErrorString(e));
        return FNX_ERR_HIP;
    }
    const unsigned char *base = static_cast<const unsigned char *>(p->blob);
    p->d_wt = reinterpret_cast<const double *>(base + o_wt);
    p->d_off = reinterpret_cast<const int32_t *>(base + o_off);
    p->d_idx = reinterpret_cast<const int32_t *>(base + o_idx);
    p->d_dense = reinterpret_cast<const float *>(base + o_dense);
    p->d_s0 = reinterpret_cast<const int32_t *>(base + o_s0);
    p->d_cnt = reinterpret_cast<const int32_t *>(base + o_cnt);
    p->d_alpha = reinterpret_cast<const uint32_t *>(base + o_alpha);
    p->d_aw = reinterpret_cast<const double *>(base + o_aw);
    p->d_inv = reinterpret_cast<const double *>(base + o_inv);
    p->d_odd = base + o_odd;
    if (p->guard_ok) (void)resize_mfma_build(ctx, t, srcN, vertical, invv.data(), &p->mf);
    p->last_use = tick++;
    *out = p.get();
    ctx->rplans.push_back(p.release());
    return FNX_OK;
}

template <int NV>
static void launch_h_guard(fnx_ctx *ctx, const ResizeGuardArgs &ga, dim3 grid)
{
    // the next row's window is prefetched while the registers allow it
    // NV <= 4: the fp32 weight pairs ride in registers (WREG); the LDS holds the exact loop's fp64 weights only
    const size_t lds = (sizeof(double) * RG_HO + (NV <= 4 ? 0 : 2 * sizeof(float))) * 4 * NV * 64;
    hipLaunchKernelGGL((resize_h_guard_kernel<NV, (NV <= 5)>), grid, dim3(256), lds, ctx->stream, ga);
}

// lanczosResize in one launch (resize_fused_kernel) when both tables take the guard form, the H windows fit the
// register matrix and the V windows a tile.  FNX_NOOP: not covered -- the caller runs the two passes.
int resize_fused(fnx_ctx *ctx, const TapTable &th, const TapTable &tv, const uint8_t *src, int sstride, int srcW, int srcH,
                 uint8_t *dst, int dstride, int nimg, const uint8_t *const *d_srcs, uint8_t *const *d_dsts)
{
    const unsigned nz = static_cast<unsigned>(nimg > 1 ? nimg : 1);
    const char *fe = form_value(ctx, FORM_RESIZE_FUSED);                     // "0": A/B and tests (the two-pass kernels)
    const bool off = (fe && fe[0] == '0') || (fe && fe[0] == '2' && th.nout < srcW);   // "2": upscales only (experiments)
    if (off || resize_guard_disabled(ctx) || th.nout <= 0 || tv.nout <= 0 || srcW <= 0 || srcH <= 0) return FNX_NOOP;
    fnx_resize_plan *ph = nullptr, *pv = nullptr;
    FNX_TRY(get_resize_plan(ctx, th, srcW, false, &ph));
    FNX_TRY(get_resize_plan(ctx, tv, srcH, true, &pv));
    if (!ph->guard_ok || !pv->guard_ok || ph->NV > 4 || pv->fused_ng < 1 || th.nout < 2) return FNX_NOOP;
    ResizeFusedArgs fa{};
    fa.srcs = nz > 1 ? d_srcs : nullptr; fa.dsts = nz > 1 ? d_dsts : nullptr;
    ResizeGuardArgs &h = fa.h, &v = fa.v;
    h.src = src; h.sstride = sstride; h.srcN = srcW; h.nout = th.nout; h.other = srcH;
    h.ngroups = ph->ngroups; h.npx = ph->npx; h.guard = ph->guard;
    h.dense = ph->d_dense; h.s0 = ph->d_s0; h.cnt = ph->d_cnt; h.alpha = ph->d_alpha;
    h.off = ph->d_off; h.idx = ph->d_idx; h.wt = ph->d_wt; h.aw = ph->d_aw; h.inv = ph->d_inv;
    v.dst = dst; v.dstride = dstride; v.srcN = srcH; v.nout = tv.nout; v.other = th.nout;
    v.ngroups = pv->ngroups; v.npx = pv->npx; v.guard = pv->guard;
    v.dense = pv->d_dense; v.s0 = pv->d_s0; v.cnt = pv->d_cnt; v.alpha = pv->d_alpha;
    v.off = pv->d_off; v.idx = pv->d_idx; v.wt = pv->d_wt; v.aw = pv->d_aw; v.inv = pv->d_inv;
    // Tile height: 32 rows (four workgroups per CU) where they hold at least eight V groups -- upscales -- else 64 (three).
    // V groups per tile: a workgroup lives for tens of microseconds, so what matters is how full the LAST round of
    // workgroups is (810 workgroups on 768 slots take two rounds, 675 one): among the counts from what the tile holds
    // down to half of it (more groups = less row halo in phase 1), the one whose rounds are fullest.
    const int gx = (ph->ngroups + 63) / 64;
    static const bool no32 = [] { const char *e = dev_env("FNX_RF_NO32"); return e && e[0] == '1'; }();   // experiments
    const bool low = !no32 && ph->NV <= 2 && pv->fused_ng32 >= 8;
    const int ngmax = low ? pv->fused_ng32 : pv->fused_ng;
    int ng = ngmax;
    {
        const double slots = (low ? 4.0 : 3.0) * ctx->num_cus;
        double best = -1.0;
        for (int cand = ngmax; cand >= std::max(2, ngmax / 2); cand--) {
            const double rounds = static_cast<double>(gx) * ((pv->ngroups + cand - 1) / cand) * nz / slots;
            const double fill = rounds / std::ceil(rounds);
            if (fill > best + 0.02) { best = fill; ng = cand; }
        }
    }
    if (const char *e = dev_env("FNX_RF_NG")) ng = std::max(1, std::min(ngmax, atoi(e)));   // experiments
    fa.ng = ng;
    const dim3 grid(gx, (pv->ngroups + ng - 1) / ng, nz);
    FNX_TRY(prof_begin(ctx, FNX_PROF_RESIZE));
    // The matrix kernel first (resize_mfma.hip); resize_fused_sparse_kernel then redoes the tiles it gave up (translucent
    // or tie-dense regions).  When it gave up most of an image -- the count comes back through host-mapped memory, read
    // here one call later -- the next 64 .. 4096 calls with this H plan skip it: a heuristic about cost, both kernels are exact.
    bool use_mf = ph->mf.ok && pv->mf.ok;
    static const bool adapt = [] { const char *e = dev_env("FNX_RM_ADAPT"); return !(e && e[0] == '0'); }();
    if (use_mf && adapt && ctx->rz_report && ctx->rz_last_gen != 0) {
        const unsigned long long repv = *reinterpret_cast<volatile unsigned long long *>(ctx->rz_report);
        if (static_cast<uint32_t>(repv >> 32) == ctx->rz_last_gen) {
            if (ctx->rz_last_h == ph) {
                // 64 calls off, twice as many each time it happens again in a row (a stream of such images), up to 4096
                if (2 * (repv & 0xffffffffull) > ctx->rz_last_cells) ph->mf_cool = ph->mf_cool_len = std::min(4096, 2 * ph->mf_cool_len);
                else ph->mf_cool_len = 32;
            }
            ctx->rz_last_gen = 0;
        }
    }
    // in the cool-down the plan's recent images were tie-dense (or translucent): the form without the fp32 passes (NV = 4: the
    // 2:1 downscales whose exact loops have the straight-line forms)
    static const bool dense_off = [] { const char *e = dev_env("FNX_RF_DENSE"); return e && e[0] == '0'; }();   // A/B and tests
    // (from the second cool-down in a row on -- mf_cool_len doubles each time the matrix kernel's retry comes back dense again --,
    // so that one synthetic image in a stream of photographs does not send the next 64 of them through fp64 loops)
    const bool dense_form = use_mf && ph->mf_cool > 0 && ph->mf_cool_len >= 128 && !low && ph->NV == 4 && !dense_off;
    if (use_mf && ph->mf_cool > 0) { ph->mf_cool--; use_mf = false; }
    if (ph->d21_last_gen != 0 && ctx->rz_report &&
        static_cast<uint32_t>(*reinterpret_cast<volatile unsigned long long *>(ctx->rz_report + 1)) == ph->d21_last_gen) {
        ph->d21_heavy = true;
        ph->d21_last_gen = 0;
    }
    if (use_mf) {
        ph->d21_heavy = false;
        const size_t cells = static_cast<size_t>(grid.x) * grid.y;
        if (cells * nz + 2 > ctx->rz_todo_cap) {
            FNX_HIP(hipStreamSynchronize(ctx->stream));
            if (ctx->rz_todo) FNX_HIP(hipFree(ctx->rz_todo));
            ctx->rz_todo = nullptr;
            ctx->rz_todo_cap = 0;
            FNX_HIP(hipMalloc(reinterpret_cast<void **>(&ctx->rz_todo), sizeof(uint32_t) * (cells * nz + 2)));
            FNX_HIP(hipMemsetAsync(ctx->rz_todo, 0, sizeof(uint32_t) * (cells * nz + 2), ctx->stream));
            ctx->rz_todo_cap = cells * nz + 2;
            ctx->rz_gen = 0;
        }
        if (!ctx->rz_report) {
            FNX_HIP(hipHostMalloc(reinterpret_cast<void **>(&ctx->rz_report), 2 * sizeof(unsigned long long), hipHostMallocMapped));
            ctx->rz_report[0] = ctx->rz_report[1] = 0;               // [1]: resize_dense21_kernel's word
        }
        if (++ctx->rz_gen == 0) {                                   // 2^32 calls later: start the stamps over
            FNX_HIP(hipMemsetAsync(ctx->rz_todo, 0, sizeof(uint32_t) * ctx->rz_todo_cap, ctx->stream));
            ctx->rz_gen = 1;
        }
        int mf_wgs = 0;
        FNX_TRY(resize_mfma_launch(ctx, ph->mf, pv->mf, src, sstride, srcW, srcH, dst, dstride, th.nout, tv.nout,
                                   ctx->rz_todo + 2, ctx->rz_todo, ctx->rz_gen, 64 * RG_HO, RG_VG * ng, static_cast<int>(grid.x), &mf_wgs,
                                   static_cast<int>(nz), d_srcs, d_dsts, static_cast<uint32_t>(cells)));
        fa.todo = ctx->rz_todo + 2;
        fa.gen = ctx->rz_gen;
        fa.cells = static_cast<int>(cells);
        fa.gx = static_cast<int>(grid.x);
        fa.counter = ctx->rz_todo;
        void *dev_report = nullptr;
        FNX_HIP(hipHostGetDevicePointer(&dev_report, ctx->rz_report, 0));
        fa.report = static_cast<unsigned long long *>(dev_report);
        ctx->rz_last_gen = ctx->rz_gen;
        ctx->rz_last_cells = static_cast<size_t>(mf_wgs);
        ctx->rz_last_h = ph;
#ifdef FNX_DEVELOP                     // a development build only (make DEVELOP=1): "2" stops before the handed-back tiles are redone
        static const char *const rm_stats = dev_env("FNX_RM_STATS");
        if (const char *e = rm_stats) {                              // how many tiles came back
            std::vector<uint32_t> cells_h(cells);
            FNX_HIP(hipStreamSynchronize(ctx->stream));
            FNX_HIP(hipMemcpy(cells_h.data(), ctx->rz_todo + 2, sizeof(uint32_t) * cells, hipMemcpyDeviceToHost));
            size_t back = 0;
            for (uint32_t c : cells_h) back += c == ctx->rz_gen;
            fprintf(stderr, "resize_mfma: S %d / %d, NC %d, %d + %d matrices, G %d / %d, %zu of %zu tiles handed back\n", ph->mf.S, pv->mf.S, ph->mf.NC,
                    ph->mf.nmat, pv->mf.nmat, ph->mf.thr / 2, pv->mf.thr / 2, back, cells);
            if (e[0] == '2') return prof_end(ctx);
        }
#endif
        note_route(ctx, FNX_PROF_RESIZE, "resize_mfma_kernel + resize_fused_sparse_kernel");
        const dim3 sgrid(static_cast<unsigned>(std::min<size_t>(cells, static_cast<size_t>(2) * ctx->num_cus)), nz);
        if (low) {
            hipLaunchKernelGGL((resize_fused_sparse_kernel<2, 32>), sgrid, dim3(256), 0, ctx->stream, fa);
        } else {
            switch (ph->NV) {
            case 2: hipLaunchKernelGGL((resize_fused_sparse_kernel<2, RF_RMAX>), sgrid, dim3(256), 0, ctx->stream, fa); break;
            case 3: hipLaunchKernelGGL((resize_fused_sparse_kernel<3, RF_RMAX>), sgrid, dim3(256), 0, ctx->stream, fa); break;
            default: hipLaunchKernelGGL((resize_fused_sparse_kernel<4, RF_RMAX>), sgrid, dim3(256), 0, ctx->stream, fa); break;
            }
        }
        FNX_HIP(hipGetLastError());
        return prof_end(ctx);
    }
    static const bool d21_off = [] { const char *e = dev_env("FNX_RF_DENSE21"); return e && e[0] == '0'; }();   // A/B and tests
    if (dense_form && ph->d21_ok && pv->d21_ok && srcW >= 20 && !d21_off && !ph->d21_heavy && ctx->rz_report && (pv->d21_fit & 2u)) {
        // V groups per tile: the count whose busiest CU has least to do.  A CU's workgroups share its SIMDs, so its time goes
        // with (workgroups on it) x (instructions of a workgroup's longest wave: ~534 per pair of tmp rows, ~1100 per V group)
        int ng21 = 1;
        double best = 0;
        for (int cand = 1; cand <= D21_NGMAX; cand++) {
            if (!(pv->d21_fit >> cand & 1u)) continue;
            const double wgs = static_cast<double>(gx) * ((pv->ngroups + cand - 1) / cand) * nz;
            const int rows = 2 * RG_VG * cand + 10, pairs = ((rows + 3) / 4 + 1) / 2;
            const double cost = std::ceil(wgs / ctx->num_cus) * (534.0 * pairs + 1100.0 * ((cand + 3) / 4));
            if (best == 0 || cost < best) { best = cost; ng21 = cand; }
        }
        if (const char *e = dev_env("FNX_RF_NG21")) {                 // experiments
            const int want = atoi(e);
            if (want >= 1 && want <= D21_NGMAX && (pv->d21_fit >> want & 1u)) ng21 = want;
        }
        fa.ng = ng21;
        const dim3 grid21(gx, (pv->ngroups + ng21 - 1) / ng21, nz);
        Dense21Args dn{};
        for (int k = 0; k < 12; k++) { dn.wh[k] = ph->d21_w[k]; dn.wv[k] = pv->d21_w[k]; }
        dn.inv_h = ph->d21_inv; dn.inv_v = pv->d21_inv;
        dn.al_h = ph->d21_alpha; dn.al_v = pv->d21_alpha;
        dn.base_h = ph->d21_base; dn.base_v = pv->d21_base;
        dn.odd_h = ph->d_odd; dn.odd_v = pv->d_odd;
        void *dev_report = nullptr;
        FNX_HIP(hipHostGetDevicePointer(&dev_report, ctx->rz_report, 0));
        dn.heavy = static_cast<unsigned long long *>(dev_report) + 1;
        if (++ctx->d21_gen == 0) ctx->d21_gen = 1;
        dn.gen = ctx->d21_gen;
        ph->d21_last_gen = ctx->d21_gen;
        note_route(ctx, FNX_PROF_RESIZE, "resize_dense21_kernel");
        hipLaunchKernelGGL(resize_dense21_kernel, grid21, dim3(256), 0, ctx->stream, fa, dn);
        FNX_HIP(hipGetLastError());
        return prof_end(ctx);
    }
    if (dense_form) {
        note_route(ctx, FNX_PROF_RESIZE, "resize_fused_dense_kernel");
        hipLaunchKernelGGL((resize_fused_dense_kernel<4, RF_RMAX>), grid, dim3(256), 0, ctx->stream, fa);
        FNX_HIP(hipGetLastError());
        return prof_end(ctx);
    }
    note_route(ctx, FNX_PROF_RESIZE, "resize_fused_kernel");
    if (low) {
        hipLaunchKernelGGL((resize_fused_kernel<2, 32>), grid, dim3(256), 0, ctx->stream, fa);
    } else {
        switch (ph->NV) {
        case 2: hipLaunchKernelGGL((resize_fused_kernel<2, RF_RMAX>), grid, dim3(256), 0, ctx->stream, fa); break;
        case 3: hipLaunchKernelGGL((resize_fused_kernel<3, RF_RMAX>), grid, dim3(256), 0, ctx->stream, fa); break;
        default: hipLaunchKernelGGL((resize_fused_kernel<4, RF_RMAX>), grid, dim3(256), 0, ctx->stream, fa); break;
        }
    }
    FNX_HIP(hipGetLastError());
    return prof_end(ctx);
}

// one pass of lanczosResize: resizeH (vertical == false: src is srcW x srcH, dst outN x srcH) or resizeV
// (src is srcW x srcH, dst srcW x outN); t.nout == outN
int resize_pass(fnx_ctx *ctx, bool vertical, const TapTable &t, const uint8_t *src, int sstride, int srcW, int srcH,
                uint8_t *dst, int dstride, ResizeHint *hint)
{
    if (t.nout <= 0 || srcW <= 0 || srcH <= 0) return FNX_OK;
    fnx_resize_plan *p = nullptr;
    FNX_TRY(get_resize_plan(ctx, t, vertical ? srcH : srcW, vertical, &p));
    const int other = vertical ? srcW : srcH;
    FNX_TRY(prof_begin(ctx, FNX_PROF_RESIZE));
    if (p->guard_ok && !resize_guard_disabled(ctx) && (!vertical || other >= RG_VPX)) {
        ResizeGuardArgs ga{};
        ga.src = src; ga.dst = dst; ga.sstride = sstride; ga.dstride = dstride;
        ga.srcN = vertical ? srcH : srcW; ga.nout = t.nout; ga.other = other;
        ga.ngroups = p->ngroups; ga.npx = p->npx; ga.guard = p->guard;
        ga.dense = p->d_dense; ga.s0 = p->d_s0; ga.cnt = p->d_cnt; ga.alpha = p->d_alpha;
        ga.off = p->d_off; ga.idx = p->d_idx; ga.wt = p->d_wt;
        ga.aw = p->d_aw; ga.inv = p->d_inv;
        note_route(ctx, FNX_PROF_RESIZE, vertical ? "resize_v_guard_kernel" : "resize_h_guard_kernel");
        if (vertical) {
            if (hint && hint->valid) {                               // the H pass of this call left its verdicts
                ga.hint = hint->cells; ga.hint_gx = hint->gx; ga.hint_gy = hint->gy; ga.hint_rows = hint->rows;
            }
            const dim3 grid((other + 256 * RG_VPX - 1) / (256 * RG_VPX), p->ngroups);
            const bool al = (other % RG_VPX) == 0 && ((reinterpret_cast<uintptr_t>(src) | static_cast<uintptr_t>(sstride)) & 15u) == 0;
            if (al) hipLaunchKernelGGL(resize_v_guard_kernel<true>, grid, dim3(256), 0, ctx->stream, ga);
            else hipLaunchKernelGGL(resize_v_guard_kernel<false>, grid, dim3(256), 0, ctx->stream, ga);
        } else {
            // rows per lane: the weight matrix is loaded once per lane, so long walks amortise it; short
            // ones fill the chip (>= ~2 workgroups per CU)
            int rows = 16;
            const int gx = (p->ngroups + 63) / 64;
            while (rows > 2 && static_cast<long>(gx) * ((other + 4 * rows - 1) / (4 * rows)) < 2L * ctx->num_cus) rows >>= 1;
            if (const char *e = dev_env("FNX_RH_ROWS")) rows = std::max(1, std::min(16, atoi(e)));   // experiments
            ga.rows = rows;
            const dim3 grid(gx, (other + 4 * rows - 1) / (4 * rows));
            if (hint) {
                hint->valid = hint->cells && static_cast<size_t>(grid.x) * grid.y <= hint->cap;
                if (hint->valid) {
                    hint->gx = static_cast<int>(grid.x); hint->gy = static_cast<int>(grid.y); hint->rows = 4 * rows;
                    ga.hint = hint->cells; ga.hint_gx = hint->gx;
                }
            }
            switch (p->NV) {
            case 2: launch_h_guard<2>(ctx, ga, grid); break;
            case 3: launch_h_guard<3>(ctx, ga, grid); break;
            case 4: launch_h_guard<4>(ctx, ga, grid); break;
            case 5: launch_h_guard<5>(ctx, ga, grid); break;
            case 6: launch_h_guard<6>(ctx, ga, grid); break;
            case 7: launch_h_guard<7>(ctx, ga, grid); break;
            case 8: launch_h_guard<8>(ctx, ga, grid); break;
            default: set_error("resize plan: no kernel for NV=%d", p->NV); return FNX_ERR_INVALID;
            }
        }
        FNX_HIP(hipGetLastError());
        return prof_end(ctx);
    }
    // fp64 kernels of round 1
    note_route(ctx, FNX_PROF_RESIZE, "fp64 resize kernels (round 1)");
    if (hint && !vertical) hint->valid = false;
    int rc;
    if (vertical) rc = launch_resize_v(ctx, src, sstride, srcW, srcH, p->d_off, p->d_idx, p->d_wt, dst, dstride, t.nout, p->contig_taps);
    else rc = launch_resize_h(ctx, src, sstride, srcW, srcH, p->d_off, p->d_idx, p->d_wt, dst, dstride, t.nout, p->contig_taps);
    if (rc < 0) return rc;
    return prof_end(ctx);
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
