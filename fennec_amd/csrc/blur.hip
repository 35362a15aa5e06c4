// GaussianBlur (effects.go:146-220) on gfx950.
//
//  * blur_pass_kernel<T,VERT>: one separable pass, any radius, thread per pixel.
//    T=double is the EXACT mode for radii > 8: fp64, unfused mul+add in the reference's tap order
//    (this TU is built with -ffp-contract=off), so H(tmp uint8) then V is bit-exact.
//    T=float is the generic fast mode for radii the fused kernel is not built for.
//  * blur_exact_kernel<R>: the EXACT mode for radii <= 8, one launch, the direct kernel's tile
//    scheme with double accumulators.
//  * blur_direct_kernel<R>: the fast path for radii <= 8 (sigma=2 -> R=6): one launch, both
//    passes in one 64 x 52 (128 lanes) or 64 x 104 (256 lanes) tile, horizontal pass fed straight
//    from global memory into a uint8 LDS intermediate (the reference rounds the intermediate to
//    uint8, effects.go:186-188), vertical pass from LDS.  HBM traffic is read-once/write-once
//    (2*S, halo re-reads hit L2); fp32 FMA accumulation (<=1 LSB off on <=0.1% samples).
//  * blur_direct_kernel<R, ..., SCORE=true> + box_from_slabs_kernel: the same blur that also
//    gathers SSIMFast's boxDownsample sums of the source and of the blurred image, so
//    SSIMFast(src, blurred) needs no second pass over either (launch_blur_scored).
#include <hip/hip_ext.h>

#include "common.hpp"
#include "devutil.hpp"

#include <algorithm>
#include <cmath>

namespace fnx {

// ------------------------------------------------------------------------------------
// generic single pass
// ------------------------------------------------------------------------------------
template <typename T>
struct PassArgs {
    const uint8_t *src;
    const uint8_t *const *srcs;   // batched: device array of n pointers (else null)
    const uint8_t *alpha;         // image the alpha byte is copied from
    const uint8_t *const *alphas;
    uint8_t *dst;
    uint8_t *const *dsts;
    int sstride, astride, dstride, w, h, radius;
    const T *kern;                // device, 2*radius+1
};

template <typename T> __device__ __forceinline__ T acc_tap(T acc, uint32_t v, T wt);
template <> __device__ __forceinline__ double acc_tap<double>(double acc, uint32_t v, double wt)
{
    return acc + u8_to_f64(v) * wt;   // r += float64(pix) * wt  (effects.go:181); the convert as 2^52-trick: v_cvt_f64_u32 is slow
}
template <> __device__ __forceinline__ float acc_tap<float>(float acc, uint32_t v, float wt)
{
    return fmaf(static_cast<float>(v), wt, acc);
}
template <typename T> __device__ __forceinline__ uint32_t round_u8(T v);
template <> __device__ __forceinline__ uint32_t round_u8<double>(double v) { return clampF_dev(v); }
template <> __device__ __forceinline__ uint32_t round_u8<float>(float v) { return pack_u8(v, 0, 0); }

template <typename T, bool VERT>
__global__ __launch_bounds__(256) void blur_pass_kernel(PassArgs<T> a)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.w || y >= a.h) return;
    const int z = blockIdx.z;
    const uint8_t *src = a.srcs ? a.srcs[z] : a.src;
    const uint8_t *alpha = a.alphas ? a.alphas[z] : a.alpha;
    uint8_t *dst = a.dsts ? a.dsts[z] : a.dst;
    T r = 0, g = 0, b = 0;
    const int n = 2 * a.radius + 1;
    for (int k = 0; k < n; k++) {
        uint32_t p;
        if (VERT) {
            int sy = clampi(y + k - a.radius, 0, a.h - 1);
            p = ld_px(src + static_cast<size_t>(sy) * a.sstride, x);
        } else {
            int sx = clampi(x + k - a.radius, 0, a.w - 1);
            p = ld_px(src + static_cast<size_t>(y) * a.sstride, sx);
        }
        const T wt = a.kern[k];
        r = acc_tap<T>(r, p & 0xffu, wt);
        g = acc_tap<T>(g, (p >> 8) & 0xffu, wt);
        b = acc_tap<T>(b, (p >> 16) & 0xffu, wt);
    }
    const uint32_t al = alpha[static_cast<size_t>(y) * a.astride + 4 * static_cast<size_t>(x) + 3];
    const uint32_t out = round_u8<T>(r) | (round_u8<T>(g) << 8) | (round_u8<T>(b) << 16) | (al << 24);
    *reinterpret_cast<uint32_t *>(dst + static_cast<size_t>(y) * a.dstride + 4 * static_cast<size_t>(x)) = out;
}

// ------------------------------------------------------------------------------------
// fused fast path
// ------------------------------------------------------------------------------------
constexpr int FUSED_RMAX = 16;    // radius <= 16 (sigma <= 5.33): the tile kernel; r3: 9..16 added (the H window streams through, see WIDE)
constexpr int SCORE_RMAX = 8;     // the one-pass GaussianBlur + SSIMFast form and the 128-lane tile stop here

struct FusedArgs {
    const uint8_t *src;
    const uint8_t *const *srcs;
    uint8_t *dst;
    uint8_t *const *dsts;
    int sstride, dstride, w, h;
    int tiles_x, tiles;   // per image
    float wt[2 * FUSED_RMAX + 1];
    double wd[2 * SCORE_RMAX + 1];   // GUARD variant only: the caller's fp64 weights for the exact fix-ups (R <= 8: kernel arguments)
    const double *wdp;               // ... R > 8: a device table (33 doubles next to 33 floats would not fit the scalar registers)
    // SCORE variant only (launch_blur_scored): boxDownsample partial sums of src and dst
    const int32_t *bx, *by;       // box column / row of each source column / row (-1: in no box)
    unsigned long long *slabs;    // [image][tile][2][slabn] packed 4 x u16 channel sums
    int nbx, nby;                 // most box columns / rows any tile touches
    int cstride;                  // entries between the LDS copies of a box table: >= (nbx+1)(nby+1) and
                                  // = 4 mod 16, so that copy c of an entry sits 8c banks away (32 x 4-byte banks)
};

// most entries of a per-tile box table, (nbx+1) * (nby+1): the two tables live in dynamic LDS sized
// per launch (4K: 2 x 176 x 8 B = 2.8 KB); the cap keeps the kernel at >= 3 workgroups per CU
constexpr int SCORE_NBOX = 512;

// channel sums of one pixel into the box table entry at byte offset `off`: the entry is
// R | G<<16 | B<<32 | A<<48, a box of <= 256 px cannot carry out of a 16-bit field.  (Two
// v_perm_b32 at 4 clocks each; `px & 0x00ff00ff`, `(px >> 8) & 0x00ff00ff` -- three 2-clock
// instructions by experiments/valurate.hip -- measured 2 % SLOWER in this kernel: with 4 waves per
// SIMD it is the instruction count that matters, not the per-class issue cost.)
__device__ __forceinline__ void box_add(unsigned long long *table, uint32_t off, uint32_t px)
{
    const uint32_t rg = __builtin_amdgcn_perm(0u, px, 0x0c010c00u);
    const uint32_t ba = __builtin_amdgcn_perm(0u, px, 0x0c030c02u);
    unsigned long long *e = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(table) + off);
    __hip_atomic_fetch_add(e, (static_cast<unsigned long long>(ba) << 32) | rg, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
}

// acc += f * w on two lanes at once (v_pk_fma_f32)
__device__ __forceinline__ v2f fma2(v2f f, float w, v2f acc)
{
    return __builtin_elementwise_fma(f, (v2f){w, w}, acc);
}
// ------------------------------------------------------------------------------------
// fused fast path: blur_direct_kernel
// ------------------------------------------------------------------------------------
// The kernel is VALU-issue bound (every VALU instruction, packed FMA included, costs its SIMD one
// quad-cycle), so this variant is cut for instructions per pixel AND for
// LDS bytes per wave (staging the input tile in LDS capped its predecessor at 4 waves per SIMD
// and cost a barrier plus ~18 VALU ops per pixel of index arithmetic):
//   * no staged input tile: the H pass reads its 2 rows x (8+2R) px window straight from
//     global memory (16-byte loads at 4-byte alignment; neighbours' overlaps hit L1/L2), so LDS
//     holds only the uint8 intermediate (16 KB per 64x64 tile) and there is ONE barrier;
//   * the source alpha rides in byte 3 of the intermediate, so the V pass needs nothing else;
//   * 128- or 256-lane workgroups, 64 x TH output tile (TH = 52 or 104 at R = 6): an H item is
//     2 rows x 8 outputs (~2 per lane), a V item is 2 columns x Q = TH/(lanes/32) = 13 output rows
//     (1 per lane): each intermediate pixel is converted (Q+2R)/Q ~ 1.9 times instead of 4.
//
// SCORE = true additionally accumulates, per tile, the boxDownsample (ssim.go:244-309) channel
// sums of the source pixels (H pass: each item holds its 16 centre pixels) and of the blurred
// pixels (V pass: the packed outputs) with LDS atomics, and writes them to the tile's slab:
// SSIMFast(src, blurred) then needs no second pass over either image (launch_blur_scored).
//
// GUARD = true makes the fp32 kernel BIT-EXACT.  Its accumulator is within E = 255*2^-24 + (2R+1)*2^-17
// (<= 1.45e-4 at R = 8) of the exact sum plus the seed, while the reference's fp64 chain is within 4e-13
// of it.  Seeded with 0.5 - G (G = 2e-4 > E + the slack of one more fp32 add), a sample is packed twice,
// as floor(acc) and as floor(acc + 2G): when the two agree no integer lies within G of the true
// sum + 0.5, so floor() of it is that same number and so is the reference's clampF -- proven, not
// sampled.  The ~0.1 % of pixels where they differ are pushed onto an LDS list and recomputed in fp64
// in the reference's order after the pass (H: patched in the intermediate; V: stored again); doing it
// in place would run the fp64 code in almost every wave.  A list overflow recomputes the whole tile.
// With SCORE the flagged blurred pixels add into a spare table entry in the V pass and into their box
// after the recompute, so the box sums are those of the exact image (an overflow clears and rebuilds them).
// E at R = 16 is 255*2^-24 + 33*2^-17 = 2.67e-4; with the slack of the second add (2^-17) G = 3e-4 covers radii 9..16
constexpr float guard_g(int R) { return R <= 8 ? 2.0e-4f : 3.0e-4f; }
constexpr int guard_fix_cap(bool score) { return score ? 512 : 2048; }   // SCORE: LDS is shared with the tables

template <int R, int NTH, int IH, bool SCORE, int RA = 1, int RB = 1, bool GUARD = false>
// (guarded at R > 8: two waves per SIMD -- at four the compiler spilled 200-540 bytes per lane and sigma = 5 took 369 us
// per 4K image; at 200+ registers it takes ~85)
__global__ __launch_bounds__(NTH, (GUARD && R > 8) ? 2 : ((GUARD && (NTH == 128 || R >= 7)) || (!GUARD && R >= 15)) ? 3 : 4) void blur_direct_kernel(FusedArgs a)
{
    constexpr int FIX_CAP = guard_fix_cap(SCORE);
    constexpr float GUARD_G = guard_g(R);
    constexpr float SEED = GUARD ? 0.5f - GUARD_G : 0.5f;
    // WIDE (R > 8): the H window is 8 + 2R <= 40 pixels x 2 rows -- held whole it is 80 registers next to the 48 accumulators.
    // It streams through instead, four pixels at a time with two chunks in flight, and the eight centre pixels (the pack
    // chains' alpha seeds) are set aside as they pass.  R <= 8 keeps the whole-window form its schedule was tuned for.
    constexpr bool WIDE = R > SCORE_RMAX;
    static_assert(!(WIDE && SCORE), "the one-pass form stops at R = 8");
    constexpr int TW = 64;
    constexpr int RG = NTH / 32;                    // row groups of the V pass (32 column pairs each)
    constexpr int TH = ((IH - 2 * R) / RG) * RG;    // output rows per tile
    constexpr int Q = TH / RG;                      // output rows per V item
    constexpr int SR = TH + 2 * R;                  // staged (H-filtered) rows actually needed
    constexpr int NT = 2 * R + 1;
    constexpr int HO = 8;
    constexpr int NPX = HO + 2 * R;
    constexpr int NV = (NPX + 3) / 4;
    constexpr int HGROUPS = TW / HO;                // 8
    constexpr int HITEMS = ((SR + 1) / 2) * HGROUPS;
    static_assert(Q * RG == TH && SR <= IH && SR % 2 == 0, "tile shape");

    __shared__ __attribute__((aligned(16))) uint32_t s_tmp[SR * TW];   // H pass: R,G,B rounded + source alpha
    // SCORE: box tables in dynamic LDS, RA copies for the source side then RB for the blurred side, each
    // slabn entries.  Lanes that add into the same box in the same instruction -- row pairs of one
    // column group in the H pass, neighbouring column pairs in the V pass -- are spread over the
    // copies: same-address LDS atomics serialise (PMC: the LDS data FIFO was full 18 % of the time
    // and address conflicts cost 8 cycles per atomic with single tables).
    // As many copies as fit without costing a resident workgroup (launch_direct_cfg picks RA, RB; they are
    // template parameters because run-time counts cost 3.5 % in SGPR pressure).
    extern __shared__ unsigned long long s_box[];
    __shared__ __attribute__((aligned(16))) uint32_t s_coloff[SCORE ? TW : 4];   // byte offset of a tile column's box
    __shared__ uint32_t s_rowoff[SCORE ? TH : 1];                      // ... of a tile row's box row
    __shared__ uint32_t s_fix[GUARD ? FIX_CAP : 1];                    // GUARD: (row << 8 | column) of pixels to recompute
    __shared__ int s_nfix[2];

    const int tile = xcd_tile(blockIdx.x, a.tiles);
    if (tile < 0) return;
    const int z = blockIdx.y;
    const uint8_t *src = a.srcs ? a.srcs[z] : a.src;
    uint8_t *dst = a.dsts ? a.dsts[z] : a.dst;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int x0 = tx * TW, y0 = ty * TH;
    const int tid = threadIdx.x;
    // a window may over-read up to 3 px past its last tap: interior = no clamp needed anywhere
    const bool interior = x0 - R >= 0 && x0 + TW + R + 3 < a.w && y0 - R >= 0 && y0 + TH + R <= a.h;
    if constexpr (GUARD) {
        if (tid < 2) s_nfix[tid] = 0;
        __syncthreads();
    }

    if constexpr (SCORE) {
        // table layout: (nby+1) rows of (nbx+1) entries; the last column / row collect pixels that
        // belong to no box (outside the image, or the source's unboxed tail columns / rows)
        for (int e = tid; e < (RA + RB) * a.cstride; e += NTH) s_box[e] = 0;
        if (tid < TW) {
            const int b0 = a.bx[x0], v = x0 + tid < a.w ? a.bx[x0 + tid] : -1;
            s_coloff[tid] = (v >= 0 && b0 >= 0) ? 8u * (v - b0) : 8u * a.nbx;
        }
        for (int r = tid; r < TH; r += NTH) {
            const int b0 = a.by[y0], v = y0 + r < a.h ? a.by[y0 + r] : -1;
            s_rowoff[r] = 8u * (a.nbx + 1) * ((v >= 0 && b0 >= 0) ? v - b0 : a.nby);
        }
        __syncthreads();
    }

    // ---- horizontal pass (effects.go:169-191): item = 2 rows x 8 outputs, window from global ----
    for (int item = tid; item < HITEMS; item += NTH) {
        const int rp = item / HGROUPS, g = item - rp * HGROUPS;
        const int xs = x0 + HO * g - R;                          // first px of the window
        u32x4 t0[WIDE ? 1 : NV], t1[WIDE ? 1 : NV];
        if constexpr (WIDE) {
            (void)t0; (void)t1;
        } else if (interior) {
            const uint8_t *p0 = src + static_cast<size_t>(y0 - R + 2 * rp) * a.sstride + 4 * static_cast<size_t>(xs);
            const uint8_t *p1 = p0 + a.sstride;
#pragma unroll
            for (int q = 0; q < NV; q++) {
                t0[q] = *(g_u32x4 *)(p0 + 16 * q);
                t1[q] = *(g_u32x4 *)(p1 + 16 * q);
            }
        } else {   // clamp-to-edge (effects.go:174-178), rows and columns
            const uint8_t *p0 = src + static_cast<size_t>(clampi(y0 - R + 2 * rp, 0, a.h - 1)) * a.sstride;
            const uint8_t *p1 = src + static_cast<size_t>(clampi(y0 - R + 2 * rp + 1, 0, a.h - 1)) * a.sstride;
#pragma unroll
            for (int q = 0; q < NV; q++) {
                const int x = xs + 4 * q;
                if (x >= 0 && x + 3 < a.w) {
                    t0[q] = *(g_u32x4 *)(p0 + 4 * static_cast<size_t>(x));
                    t1[q] = *(g_u32x4 *)(p1 + 4 * static_cast<size_t>(x));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int xc = clampi(x + e, 0, a.w - 1);
                        t0[q][e] = ld_px(p0, xc);
                        t1[q][e] = ld_px(p1, xc);
                    }
                }
            }
        }
        if constexpr (SCORE) {   // source side of the box sums: this item's 2 x 8 centre pixels
            const int r0 = 2 * rp - R;
            unsigned long long *s_box_a = s_box + (rp & (RA - 1)) * a.cstride;
            const u32x4 ca = *reinterpret_cast<const u32x4 *>(s_coloff + HO * g);
            const u32x4 cb = *reinterpret_cast<const u32x4 *>(s_coloff + HO * g + 4);
            if (r0 >= 0 && r0 < TH) {
                const uint32_t ro = s_rowoff[r0];
#pragma unroll
                for (int j = 0; j < HO; j++) box_add(s_box_a, ro + (j < 4 ? ca[j & 3] : cb[j & 3]), t0[(j + R) / 4][(j + R) % 4]);
            }
            if (r0 + 1 >= 0 && r0 + 1 < TH) {
                const uint32_t ro = s_rowoff[r0 + 1];
#pragma unroll
                for (int j = 0; j < HO; j++) box_add(s_box_a, ro + (j < 4 ? ca[j & 3] : cb[j & 3]), t1[(j + R) / 4][(j + R) % 4]);
            }
        }
        v2f acc[HO][3];
#pragma unroll
        for (int j = 0; j < HO; j++) acc[j][0] = acc[j][1] = acc[j][2] = (v2f){SEED, SEED};
        uint32_t cw0[WIDE ? HO : 1], cw1[WIDE ? HO : 1];          // WIDE: the centre pixels of the two rows
        if constexpr (WIDE) {
            const int yr = y0 - R + 2 * rp;
            const uint8_t *q0 = src + static_cast<size_t>(interior ? yr : clampi(yr, 0, a.h - 1)) * a.sstride;
            const uint8_t *q1 = src + static_cast<size_t>(interior ? yr + 1 : clampi(yr + 1, 0, a.h - 1)) * a.sstride;
            // one chunk of both rows; the clamped form (effects.go:174-178) only on tiles that touch the image's edge
            // (`interior` is uniform over the workgroup: the two forms are separate loops, no branch sits between a
            // load and the chunk that waits for it)
            auto chunk = [&](auto edge, int q, u32x4 &c0, u32x4 &c1) {
                const int x = xs + 4 * q;
                if constexpr (!decltype(edge)::value) {
                    c0 = *(g_u32x4 *)(q0 + 4 * static_cast<size_t>(x));
                    c1 = *(g_u32x4 *)(q1 + 4 * static_cast<size_t>(x));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int xc = clampi(x + e, 0, a.w - 1);
                        c0[e] = ld_px(q0, xc);
                        c1[e] = ld_px(q1, xc);
                    }
                }
            };
            auto stream = [&](auto edge) {
                u32x4 a0, a1, b0, b1;
                chunk(edge, 0, a0, a1);
                chunk(edge, 1, b0, b1);
#pragma unroll
                for (int q = 0; q < NV; q++) {
                    const u32x4 c0 = a0, c1 = a1;
                    a0 = b0; a1 = b1;
                    if (q + 2 < NV) chunk(edge, q + 2, b0, b1);
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int i = 4 * q + e;
                        if (i < NPX) {
                            const uint32_t p0 = c0[e], p1 = c1[e];
                            if (i >= R && i < R + HO) { cw0[i - R] = p0; cw1[i - R] = p1; }
                            const v2f f0 = {static_cast<float>(p0 & 0xffu), static_cast<float>((p0 >> 8) & 0xffu)};
                            const v2f f1 = {static_cast<float>((p0 >> 16) & 0xffu), static_cast<float>(p1 & 0xffu)};
                            const v2f f2 = {static_cast<float>((p1 >> 8) & 0xffu), static_cast<float>((p1 >> 16) & 0xffu)};
#pragma unroll
                            for (int j = 0; j < HO; j++) {
                                const int k = i - j;
                                if (k >= 0 && k < NT) {
                                    acc[j][0] = fma2(f0, a.wt[k], acc[j][0]);
                                    acc[j][1] = fma2(f1, a.wt[k], acc[j][1]);
                                    acc[j][2] = fma2(f2, a.wt[k], acc[j][2]);
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            };
            if (interior) stream(std::false_type{}); else stream(std::true_type{});
        }
#pragma unroll
        for (int q = 0; q < (WIDE ? 0 : NV); q++) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int i = 4 * q + e;
                if (i < NPX) {
                    const uint32_t p0 = t0[q][e], p1 = t1[q][e];
                    const v2f f0 = {static_cast<float>(p0 & 0xffu), static_cast<float>((p0 >> 8) & 0xffu)};
                    const v2f f1 = {static_cast<float>((p0 >> 16) & 0xffu), static_cast<float>(p1 & 0xffu)};
                    const v2f f2 = {static_cast<float>((p1 >> 8) & 0xffu), static_cast<float>((p1 >> 16) & 0xffu)};
#pragma unroll
                    for (int j = 0; j < HO; j++) {
                        const int k = i - j;
                        if (k >= 0 && k < NT) {
                            acc[j][0] = fma2(f0, a.wt[k], acc[j][0]);
                            acc[j][1] = fma2(f1, a.wt[k], acc[j][1]);
                            acc[j][2] = fma2(f2, a.wt[k], acc[j][2]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);   // keep the converts next to their FMAs
                }
            }
        }
        // round to the uint8 intermediate (effects.go:186-188); the pack chain is seeded with the
        // centre source pixel so its alpha lands in byte 3
        uint32_t o0[HO], o1[HO];
        fp32_round_toward_zero();
#pragma unroll
        for (int j = 0; j < HO; j++) {
            constexpr int c = 0;
            uint32_t sd0, sd1;                                   // the centre source pixels: their alpha stays in byte 3
            if constexpr (WIDE) {
                sd0 = cw0[j]; sd1 = cw1[j];
            } else {
                sd0 = t0[(j + R) / 4][(j + R) % 4]; sd1 = t1[(j + R) / 4][(j + R) % 4];
            }
            (void)c;
            o0[j] = pk8(acc[j][1].x, 2, pk8(acc[j][0].y, 1, pk8(acc[j][0].x, 0, sd0)));
            o1[j] = pk8(acc[j][2].y, 2, pk8(acc[j][2].x, 1, pk8(acc[j][1].y, 0, sd1)));
            if constexpr (GUARD) {   // second pack at acc + 2G: a differing pixel is within G of a rounding boundary
                const v2f g2 = {2.0f * GUARD_G, 2.0f * GUARD_G};
                const v2f h0 = acc[j][0] + g2, h1 = acc[j][1] + g2, h2 = acc[j][2] + g2;
                const uint32_t p0 = pk8(h1.x, 2, pk8(h0.y, 1, pk8(h0.x, 0, sd0)));
                const uint32_t p1 = pk8(h2.y, 2, pk8(h2.x, 1, pk8(h1.y, 0, sd1)));
                if (p0 != o0[j]) {
                    const int e = atomicAdd(&s_nfix[0], 1);
                    if (e < FIX_CAP) s_fix[e] = ((2 * rp) << 8) | (HO * g + j);
                }
                if (p1 != o1[j]) {
                    const int e = atomicAdd(&s_nfix[0], 1);
                    if (e < FIX_CAP) s_fix[e] = ((2 * rp + 1) << 8) | (HO * g + j);
                }
            }
        }
        fp32_round_nearest();
#pragma unroll
        for (int b = 0; b < HO / 4; b++) {
            *reinterpret_cast<u32x4 *>(s_tmp + (2 * rp) * TW + HO * g + 4 * b) = (u32x4){o0[4 * b], o0[4 * b + 1], o0[4 * b + 2], o0[4 * b + 3]};
            *reinterpret_cast<u32x4 *>(s_tmp + (2 * rp + 1) * TW + HO * g + 4 * b) = (u32x4){o1[4 * b], o1[4 * b + 1], o1[4 * b + 2], o1[4 * b + 3]};
        }
    }
    __syncthreads();
    if constexpr (GUARD) {   // exact H results for the flagged pixels (effects.go:169-191, fp64, taps ascending)
        const int nfix = s_nfix[0];
        const int total = nfix > FIX_CAP ? SR * TW : nfix;
        for (int e = tid; e < total; e += NTH) {
            const int row = nfix > FIX_CAP ? e / TW : static_cast<int>(s_fix[e] >> 8);
            const int col = nfix > FIX_CAP ? e - row * TW : static_cast<int>(s_fix[e] & 0xffu);
            const uint8_t *prow = src + static_cast<size_t>(clampi(y0 - R + row, 0, a.h - 1)) * a.sstride;
            double r = 0, g = 0, b = 0;
            auto tap = [&](int k) {
                const uint32_t p = ld_px(prow, clampi(x0 + col + k - R, 0, a.w - 1));
                const double wk = WIDE ? a.wdp[k] : a.wd[WIDE ? 0 : k];
                r = r + u8_to_f64(p & 0xffu) * wk;
                g = g + u8_to_f64((p >> 8) & 0xffu) * wk;
                b = b + u8_to_f64((p >> 16) & 0xffu) * wk;
            };
            if constexpr (WIDE) {      // a rolled loop: 33 taps unrolled keep 33 loads' registers alive in a kernel that has none to spare
#pragma unroll 2
                for (int k = 0; k < NT; k++) tap(k);
            } else {
#pragma unroll
                for (int k = 0; k < NT; k++) tap(k);
            }
            s_tmp[row * TW + col] = clampF_dev(r) | (clampF_dev(g) << 8) | (clampF_dev(b) << 16) |
                                    (s_tmp[row * TW + col] & 0xff000000u);
        }
        __syncthreads();
    }

    // ---- vertical pass (effects.go:195-217): item = 2 columns x Q output rows ----
    {
        const int cp = tid & 31, rg = tid >> 5;                  // column pair, row group
        const int x = x0 + 2 * cp;
        const uint32_t *colp = s_tmp + (rg * Q) * TW + 2 * cp;
        v2f acc[Q][3];                                           // (r0,g0) (b0,r1) (g1,b1)
#pragma unroll
        for (int j = 0; j < Q; j++) acc[j][0] = acc[j][1] = acc[j][2] = (v2f){SEED, SEED};
        constexpr bool KEEP_AL = GUARD && !WIDE;                 // GUARD at R <= 8: the centre rows' words stay in registers (see below)
        u32x2 alg[KEEP_AL ? Q : 1];
        u32x2 tn = *reinterpret_cast<const u32x2 *>(colp);
#pragma unroll
        for (int i = 0; i < Q + 2 * R; i++) {
            const u32x2 t = tn;
            if (i + 1 < Q + 2 * R) tn = *reinterpret_cast<const u32x2 *>(colp + (i + 1) * TW);   // prefetch next row
            if constexpr (KEEP_AL) { if (i >= R && i < R + Q) alg[i - R] = t; }
            const v2f f0 = {static_cast<float>(t.x & 0xffu), static_cast<float>((t.x >> 8) & 0xffu)};
            const v2f f1 = {static_cast<float>((t.x >> 16) & 0xffu), static_cast<float>(t.y & 0xffu)};
            const v2f f2 = {static_cast<float>((t.y >> 8) & 0xffu), static_cast<float>((t.y >> 16) & 0xffu)};
#pragma unroll
            for (int j = 0; j < Q; j++) {
                const int k = i - j;
                if (k >= 0 && k < NT) {
                    acc[j][0] = fma2(f0, a.wt[k], acc[j][0]);
                    acc[j][1] = fma2(f1, a.wt[k], acc[j][1]);
                    acc[j][2] = fma2(f2, a.wt[k], acc[j][2]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        u32x2 o[Q];
        uint32_t flagged = 0;                                    // GUARD && SCORE: bit 2j / 2j+1 = o[j].x / .y awaits its recompute
        static_assert(2 * Q <= 32, "flag bits");
        // The centre rows carry the source alpha in byte 3 (effects.go:215): they are read AGAIN here, as the seeds of
        // the pack chains, instead of being held through the accumulation -- 2 Q registers less (r3: the kernel fits
        // 104 VGPRs instead of 128, so that four resident waves per SIMD leave a fifth of the register file to the tail
        // kernels of the previous step, which then run beside the blur's waves instead of in place of one).  The guarded
        // variant packs every sample twice and is register-bound elsewhere: it keeps them (re-reading made it spill).
        u32x2 aln = {0, 0}, aln2 = {0, 0};
        if constexpr (!KEEP_AL) {
            aln = *reinterpret_cast<const u32x2 *>(colp + R * TW);
            aln2 = *reinterpret_cast<const u32x2 *>(colp + (R + 1) * TW);
        }
        fp32_round_toward_zero();
#pragma unroll
        for (int j = 0; j < Q; j++) {
            u32x2 al;
            if constexpr (KEEP_AL) {
                al = alg[j];
            } else {
                al = aln;                                        // two rows ahead: an LDS read is ~64 clocks
                aln = aln2;
                if (j + 2 < Q) aln2 = *reinterpret_cast<const u32x2 *>(colp + (j + 2 + R) * TW);
            }
            o[j].x = pk8(acc[j][1].x, 2, pk8(acc[j][0].y, 1, pk8(acc[j][0].x, 0, al.x)));
            o[j].y = pk8(acc[j][2].y, 2, pk8(acc[j][2].x, 1, pk8(acc[j][1].y, 0, al.y)));
            if constexpr (GUARD) {
                const v2f g2 = {2.0f * GUARD_G, 2.0f * GUARD_G};
                const v2f h0 = acc[j][0] + g2, h1 = acc[j][1] + g2, h2 = acc[j][2] + g2;
                const uint32_t p0 = pk8(h1.x, 2, pk8(h0.y, 1, pk8(h0.x, 0, al.x)));
                const uint32_t p1 = pk8(h2.y, 2, pk8(h2.x, 1, pk8(h1.y, 0, al.y)));
                if (p0 != o[j].x) {
                    const int e = atomicAdd(&s_nfix[1], 1);
                    if (e < FIX_CAP) s_fix[e] = ((rg * Q + j) << 8) | (2 * cp);
                    if constexpr (SCORE) flagged |= 1u << (2 * j);
                }
                if (p1 != o[j].y) {
                    const int e = atomicAdd(&s_nfix[1], 1);
                    if (e < FIX_CAP) s_fix[e] = ((rg * Q + j) << 8) | (2 * cp + 1);
                    if constexpr (SCORE) flagged |= 2u << (2 * j);
                }
            }
            asm volatile("" : "+v"(o[j].x), "+v"(o[j].y));   // keep the accumulation out of the store branches
        }
        fp32_round_nearest();
        if constexpr (SCORE) {   // blurred side of the box sums (columns / rows outside the image -> spare entries)
            unsigned long long *s_box_b = s_box + (RA + (cp & (RB - 1))) * a.cstride;
            const u32x2 co = *reinterpret_cast<const u32x2 *>(s_coloff + 2 * cp);
            if constexpr (GUARD) {
#pragma unroll
                for (int j = 0; j < Q; j++) {   // provisional values of flagged pixels go to the spare corner entry
                    const uint32_t ro = s_rowoff[rg * Q + j];
                    const uint32_t spare = 8u * ((a.nbx + 1) * a.nby + a.nbx);
                    box_add(s_box_b, (flagged >> (2 * j)) & 1u ? spare : ro + co.x, o[j].x);
                    box_add(s_box_b, (flagged >> (2 * j)) & 2u ? spare : ro + co.y, o[j].y);
                }
            } else {
                // An item's Q rows cross two or three box rows: the channel sums of a column's run of rows inside one box
                // row are taken in registers (16-bit fields: Q * 255 < 2^16) and go to the table once per run -- 4 to 6
                // LDS atomics per item instead of 2 Q.  The row -> box row map is the same for the 32 lanes of a row group.
                uint32_t cur = s_rowoff[rg * Q];
                uint32_t rg0 = 0, ba0 = 0, rg1 = 0, ba1 = 0;
                auto flush = [&](uint32_t ro) {
                    __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(s_box_b) + ro + co.x),
                                           (static_cast<unsigned long long>(ba0) << 32) | rg0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(s_box_b) + ro + co.y),
                                           (static_cast<unsigned long long>(ba1) << 32) | rg1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                };
#pragma unroll
                for (int j = 0; j < Q; j++) {
                    const uint32_t ro = s_rowoff[rg * Q + j];
                    if (ro != cur) {
                        flush(cur);
                        rg0 = ba0 = rg1 = ba1 = 0;
                        cur = ro;
                    }
                    rg0 += __builtin_amdgcn_perm(0u, o[j].x, 0x0c010c00u);
                    ba0 += __builtin_amdgcn_perm(0u, o[j].x, 0x0c030c02u);
                    rg1 += __builtin_amdgcn_perm(0u, o[j].y, 0x0c010c00u);
                    ba1 += __builtin_amdgcn_perm(0u, o[j].y, 0x0c030c02u);
                }
                flush(cur);
            }
        }
        if (x < a.w) {
#pragma unroll
            for (int j = 0; j < Q; j++) {
                const int y = y0 + rg * Q + j;
                if (y < a.h) {
                    uint8_t *dp = dst + static_cast<size_t>(y) * a.dstride + 4 * static_cast<size_t>(x);
                    if (x + 1 < a.w) {
                        *(__attribute__((address_space(1))) u32x2 *)(dp) = o[j];
                    } else {
                        *(g_u32w *)(dp) = o[j].x;
                    }
                }
            }
        }
    }
    if constexpr (GUARD) {   // exact V results for the flagged pixels, stored over the provisional ones: the
                             // barrier's vmcnt(0) has the first stores acknowledged before these are issued
        __syncthreads();
        const int nfix = s_nfix[1];
        const int total = nfix > FIX_CAP ? TH * TW : nfix;
        if constexpr (SCORE) {   // list overflow: every pixel is recomputed, so the blurred-side tables start over
            if (nfix > FIX_CAP) {
                for (int e = tid; e < RB * a.cstride; e += NTH) s_box[RA * a.cstride + e] = 0;
                __syncthreads();
            }
        }
        for (int e = tid; e < total; e += NTH) {
            const int row = nfix > FIX_CAP ? e / TW : static_cast<int>(s_fix[e] >> 8);
            const int col = nfix > FIX_CAP ? e - row * TW : static_cast<int>(s_fix[e] & 0xffu);
            const int x = x0 + col, y = y0 + row;
            if (x >= a.w || y >= a.h) continue;
            double r = 0, g = 0, b = 0;
            auto tap = [&](int k) {
                const uint32_t p = s_tmp[(row + k) * TW + col];
                const double wk = WIDE ? a.wdp[k] : a.wd[WIDE ? 0 : k];
                r = r + u8_to_f64(p & 0xffu) * wk;
                g = g + u8_to_f64((p >> 8) & 0xffu) * wk;
                b = b + u8_to_f64((p >> 16) & 0xffu) * wk;
            };
            if constexpr (WIDE) {
#pragma unroll 2
                for (int k = 0; k < NT; k++) tap(k);
            } else {
#pragma unroll
                for (int k = 0; k < NT; k++) tap(k);
            }
            const uint32_t px = clampF_dev(r) | (clampF_dev(g) << 8) | (clampF_dev(b) << 16) |
                                (s_tmp[(row + R) * TW + col] & 0xff000000u);
            *(g_u32w *)(dst + static_cast<size_t>(y) * a.dstride + 4 * static_cast<size_t>(x)) = px;
            if constexpr (SCORE) box_add(s_box + (RA + (e & (RB - 1))) * a.cstride, s_rowoff[row] + s_coloff[col], px);
        }
    }
    if constexpr (SCORE) {
        __syncthreads();
        const int slabn = (a.nbx + 1) * (a.nby + 1);
        unsigned long long *slab = a.slabs + (static_cast<size_t>(z) * a.tiles + tile) * 2 * slabn;
        for (int e = tid; e < 2 * slabn; e += NTH) {      // fold the copies: [0, slabn) source, [slabn, 2 slabn) blurred
            const bool src_side = e < slabn;
            const int ent = src_side ? e : e - slabn, first = src_side ? 0 : RA, copies = src_side ? RA : RB;
            unsigned long long v = 0;
            for (int c = 0; c < copies; c++) v += s_box[(first + c) * a.cstride + ent];
            slab[e] = v;
        }
    }
}

// ------------------------------------------------------------------------------------
// box tables from the slabs of a SCORE launch: the boxDownsample'd source (z < n) and blurred
// (z >= n) images, identical to box_tiled_kernel's output (integer sums are order-free)
// ------------------------------------------------------------------------------------
struct BoxRef {           // where one output column (or row) finds its partial sums
    int32_t part0, part1; // slab offsets of the (at most two) tiles the box spans; part1 < 0: one tile
    int32_t len, pad;     // box width (height) in source px
};

struct SlabArgs {
    const unsigned long long *slabs;
    const uint32_t *magic;    // magic[c] = floor(2^32 / 2c) + 1, c <= 256: (2 n + c) / 2c by one v_mul_hi_u32
    const uint32_t *tiedown;  // tiedown[8 c + (k >> 5)] bit (k & 31): the reference's fp64 product at the tie
                              // n / c == k + 0.5 falls BELOW it (clampF then gives k, not k + 1)
    const BoxRef *xref, *yref;
    uint8_t *dst;             // [src 0..n-1][blurred 0..n-1] tight dstW x dstH planes
    size_t plane, image_slabs;  // entries per image: tiles * 2 * slabn
    int n, dstW, dstH, slabn;
};

// clampF(fl(n * fl(1 / c))) (ssim.go:301-308) for an integer channel sum n <= 255 c, c <= 256, in integers.
// n / c is a tie only when 2 n == (2 k + 1) c; anywhere else it is at least 1 / 2c >= 2^-9 away from the next
// half-integer, far beyond the fp64 product's error (2^-44), so the reference's result is the exact quotient
// rounded half up: q = (2 n + c) / 2c.  AT a tie the product fl(n * fl(1 / c)) lands on k + 0.5 or an ulp
// below it depending on c and k: the host evaluates exactly that product for every (c, k) -- the same two IEEE
// operations the reference performs -- and hands over the ones that round DOWN as a bit table.
__device__ __forceinline__ uint32_t box_mean_u8(uint32_t n, uint32_t c, uint32_t magic, const uint32_t *tiedown)
{
    const uint32_t t = 2u * n + c;
    uint32_t q = __umulhi(t, magic);                 // t / 2c exactly: t * (magic * 2c - 2^32) < 2^32 for t <= 511 c
    if (t == q * 2u * c && q > 0) {                  // tie between q - 1 and q
        const uint32_t k = q - 1;
        if ((tiedown[8u * c + (k >> 5)] >> (k & 31u)) & 1u) q = k;
    }
    return q;
}

// One thread per FOUR adjacent boxes of an output row, both images.  The index arithmetic (box_edge, tile and entry
// of each part) is per column / per row and comes from host tables: what is left is <= 32 slab loads, ALL issued
// before the first use (parts that do not exist -- seven boxes in eight lie inside one tile -- are not loaded: read
// unconditionally, 8 loads per box made the kernel address-bound, 61 us alone), and the integer finish.  r3: the kernel runs under the next step's blur, where a wave that waits on memory holds a slot a blur
// wave wants -- one box per thread was 73.7 k waves of two dependent load levels each per 32-image step (36 us alone,
// 100-120 us under the blur); four boxes per thread is a quarter of the waves at the same depth.
// Alpha is not produced: both planes only feed toLuminance (ssim.go:207-220).
constexpr int BFS_BOXES = 4;
__global__ __launch_bounds__(256, 5) void box_from_slabs_kernel(SlabArgs a)   // <= 96 VGPRs: fits beside four blur waves per SIMD
{
    const int dx0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * BFS_BOXES;
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx0 >= a.dstW || dy >= a.dstH) return;
    const int i = blockIdx.z;
    const BoxRef yr = a.yref[dy];
    BoxRef xr[BFS_BOXES];
#pragma unroll
    for (int j = 0; j < BFS_BOXES; j++) xr[j] = a.xref[dx0 + j < a.dstW ? dx0 + j : a.dstW - 1];
    const unsigned long long *base = a.slabs + a.image_slabs * i;
    const bool y2 = yr.part1 >= 0;
    unsigned long long v[BFS_BOXES][2], vx[BFS_BOXES][2], vy[BFS_BOXES][2];   // [box][0: source, 1: blurred]: part 0; the second tile in x; in y (+ the corner)
    uint32_t magic[BFS_BOXES], cnt[BFS_BOXES];
#pragma unroll
    for (int j = 0; j < BFS_BOXES; j++) {
        cnt[j] = static_cast<uint32_t>(xr[j].len * yr.len);
        magic[j] = a.magic[cnt[j]];
#pragma unroll
        for (int img = 0; img < 2; img++) {
            v[j][img] = base[yr.part0 + xr[j].part0 + img * a.slabn];
            vx[j][img] = vy[j][img] = 0;
        }
    }
#pragma unroll
    for (int j = 0; j < BFS_BOXES; j++) {
        if (xr[j].part1 >= 0) {
#pragma unroll
            for (int img = 0; img < 2; img++) vx[j][img] = base[yr.part0 + xr[j].part1 + img * a.slabn];
        }
    }
    if (y2) {                                                    // uniform over the workgroup's row
#pragma unroll
        for (int j = 0; j < BFS_BOXES; j++) {
#pragma unroll
            for (int img = 0; img < 2; img++) vy[j][img] = base[yr.part1 + xr[j].part0 + img * a.slabn];
            if (xr[j].part1 >= 0) {
#pragma unroll
                for (int img = 0; img < 2; img++) vx[j][img] += base[yr.part1 + xr[j].part1 + img * a.slabn];
            }
        }
    }
    // 2 x u16 per word: the whole box is <= 256 px, so no field overflows
    uint32_t o[2][BFS_BOXES];
#pragma unroll
    for (int j = 0; j < BFS_BOXES; j++) {
#pragma unroll
        for (int img = 0; img < 2; img++) {
            const unsigned long long t = v[j][img] + vx[j][img] + vy[j][img];
            const uint32_t rg = static_cast<uint32_t>(t), b = static_cast<uint32_t>(t >> 32);
            // clampF(sum * (1.0 / count)) per channel (ssim.go:301-308)
            o[img][j] = box_mean_u8(rg & 0xffffu, cnt[j], magic[j], a.tiedown) | (box_mean_u8(rg >> 16, cnt[j], magic[j], a.tiedown) << 8) |
                        (box_mean_u8(b & 0xffffu, cnt[j], magic[j], a.tiedown) << 16);
        }
    }
#pragma unroll
    for (int img = 0; img < 2; img++) {
        uint8_t *dp = a.dst + a.plane * (static_cast<size_t>(img) * a.n + i) + (static_cast<size_t>(dy) * a.dstW + dx0) * 4;
        if (dx0 + BFS_BOXES <= a.dstW && !(reinterpret_cast<uintptr_t>(dp) & 15)) {
            *reinterpret_cast<u32x4 *>(dp) = (u32x4){o[img][0], o[img][1], o[img][2], o[img][3]};
        } else {
#pragma unroll
            for (int j = 0; j < BFS_BOXES; j++)
                if (dx0 + j < a.dstW) reinterpret_cast<uint32_t *>(dp)[j] = o[img][j];
        }
    }
}

template <int R, int NTH, int IH, bool SCORE, bool GUARD = false>
static int launch_direct_cfg(fnx_ctx *ctx, int n, FusedArgs &fa)
{
    constexpr int TW = 64, RG = NTH / 32, TH = ((IH - 2 * R) / RG) * RG;
    note_route(ctx, FNX_PROF_MAIN, SCORE ? (GUARD ? "blur_direct_kernel<SCORE, GUARD>" : "blur_direct_kernel<SCORE>")
                                         : (GUARD ? "blur_direct_kernel<GUARD>" : "blur_direct_kernel"));
    fa.tiles_x = (fa.w + TW - 1) / TW;
    fa.tiles = fa.tiles_x * ((fa.h + TH - 1) / TH);
    dim3 grid(8 * ((fa.tiles + 7) / 8), n);
    // the launch's events ride on its own packet (common.hpp: LaunchEvents): the profile pair when this class is
    // profiled, and for SCORE launches always a stop event -- the tail on the ctx's second stream waits for it
    LaunchEvents ev;
    FNX_TRY(prof_bind(ctx, FNX_PROF_MAIN, &ev));
    if constexpr (SCORE) {
        if (!ev.stop) ev.stop = ctx->ev_blur[ctx->parity];
        ctx->blur_done = ev.stop;
    }
    if constexpr (SCORE) {
        fa.cstride = (((fa.nbx + 1) * (fa.nby + 1) + 11) / 16) * 16 + 4;
        // LDS left per workgroup at 4 (256 lanes) / 7 (128 lanes) workgroups per CU, next to the uint8
        // intermediate: as many table copies as fit in it
        const size_t budget = (NTH == 256 ? 9600 : 6000) - (GUARD ? sizeof(uint32_t) * guard_fix_cap(true) : 0);
        const size_t per_copy = sizeof(unsigned long long) * fa.cstride;
        if (6 * per_copy <= budget)
            hipExtLaunchKernelGGL((blur_direct_kernel<R, NTH, IH, true, 2, 4, GUARD>), grid, dim3(NTH), 6 * per_copy, ctx->stream, ev.start, ev.stop, 0, fa);
        else if (3 * per_copy <= budget)
            hipExtLaunchKernelGGL((blur_direct_kernel<R, NTH, IH, true, 1, 2, GUARD>), grid, dim3(NTH), 3 * per_copy, ctx->stream, ev.start, ev.stop, 0, fa);
        else
            hipExtLaunchKernelGGL((blur_direct_kernel<R, NTH, IH, true, 1, 1, GUARD>), grid, dim3(NTH), 2 * per_copy, ctx->stream, ev.start, ev.stop, 0, fa);
    } else {
        hipExtLaunchKernelGGL((blur_direct_kernel<R, NTH, IH, false, 1, 1, GUARD>), grid, dim3(NTH), 0, ctx->stream, ev.start, ev.stop, 0, fa);
    }
    FNX_HIP(hipGetLastError());
    return FNX_OK;
}

// Two tile shapes (measured on MI355X, 4K: 23.6 vs 22.9 us; 1080p: 5.6 vs 6.0 us per image):
//   128 lanes x 64 intermediate rows  (TH = 52 at R=6, H-pass row halo 1.23x)
//   256 lanes x (2*TH+2R) rows         (TH = 104,      halo 1.12x) -- wins when the image height
//     does not leave a mostly empty last tile row and there are enough tiles to fill the chip.
// Cost model: H work ~ staged rows, V work ~ output rows (about 55 : 45 of the instructions).
constexpr int WIDE_IH = 128;      // staged rows of the R > 8 tile (256 lanes): TH = the multiple of 8 below 128 - 2R
static int direct_tile_rows(int R, bool tall)
{
    if (R > SCORE_RMAX && R <= FUSED_RMAX) return ((WIDE_IH - 2 * R) / 8) * 8;
    const int th0 = R >= 1 && R <= SCORE_RMAX ? ((64 - 2 * R) / 4) * 4 : 4;   // never 0: callers divide by it
    return tall ? 2 * th0 : th0;
}
static bool direct_tall(const fnx_ctx *ctx, int R, int n, int w, int h)
{
    if (R > SCORE_RMAX) return true;                            // one tile shape for the wide radii
    const int TH0 = direct_tile_rows(R, false), TH1 = direct_tile_rows(R, true);
    const long ty0 = (h + TH0 - 1) / TH0, ty1 = (h + TH1 - 1) / TH1;
    const double cost0 = ty0 * (0.55 * (TH0 + 2 * R) + 0.45 * TH0);
    const double cost1 = ty1 * (0.55 * (TH1 + 2 * R) + 0.45 * TH1);
    const long tiles1 = ty1 * ((w + 63) / 64) * n;
    return cost1 < 0.95 * cost0 && tiles1 >= 4L * ctx->num_cus;
}

// The guarded kernel's error bound assumes what GaussianBlur's own kernel guarantees (effects.go:155-165):
// weights >= 0 that sum to 1, so accumulators stay below 256.
static bool guard_kernel_ok(const double *kernel, int radius)
{
    double sum = 0;
    for (int i = 0; i < 2 * radius + 1; i++) {
        if (!(kernel[i] >= 0)) return false;
        sum += kernel[i];
    }
    return sum <= 1.0 + 1e-9;
}

template <int R, bool SCORE, bool GUARD = false>
static int launch_direct(fnx_ctx *ctx, int n, FusedArgs &fa, bool tall)
{
    if constexpr (R > SCORE_RMAX) {
        static_assert(!SCORE, "one-pass form: R <= 8");
        return launch_direct_cfg<R, 256, WIDE_IH, false, GUARD>(ctx, n, fa);
    } else {
        constexpr int TH0 = ((64 - 2 * R) / 4) * 4;
        if (tall) return launch_direct_cfg<R, 256, 2 * TH0 + 2 * R, SCORE, GUARD>(ctx, n, fa);
        return launch_direct_cfg<R, 128, 64, SCORE, GUARD>(ctx, n, fa);
    }
}

template <bool SCORE, bool GUARD = false>
static int launch_direct_radius(fnx_ctx *ctx, int radius, int n, FusedArgs &fa, bool tall)
{
    switch (radius) {
    case 1: return launch_direct<1, SCORE, GUARD>(ctx, n, fa, tall);
    case 2: return launch_direct<2, SCORE, GUARD>(ctx, n, fa, tall);
    case 3: return launch_direct<3, SCORE, GUARD>(ctx, n, fa, tall);
    case 4: return launch_direct<4, SCORE, GUARD>(ctx, n, fa, tall);
    case 5: return launch_direct<5, SCORE, GUARD>(ctx, n, fa, tall);
    case 6: return launch_direct<6, SCORE, GUARD>(ctx, n, fa, tall);
    case 7: return launch_direct<7, SCORE, GUARD>(ctx, n, fa, tall);
    case 8: return launch_direct<8, SCORE, GUARD>(ctx, n, fa, tall);
    }
    if constexpr (!SCORE) {
        switch (radius) {
        case 9: return launch_direct<9, false, GUARD>(ctx, n, fa, tall);
        case 10: return launch_direct<10, false, GUARD>(ctx, n, fa, tall);
        case 11: return launch_direct<11, false, GUARD>(ctx, n, fa, tall);
        case 12: return launch_direct<12, false, GUARD>(ctx, n, fa, tall);
        case 13: return launch_direct<13, false, GUARD>(ctx, n, fa, tall);
        case 14: return launch_direct<14, false, GUARD>(ctx, n, fa, tall);
        case 15: return launch_direct<15, false, GUARD>(ctx, n, fa, tall);
        case 16: return launch_direct<16, false, GUARD>(ctx, n, fa, tall);
        }
    }
    return FNX_ERR_INVALID;
}

// GaussianBlur (fast mode) of n images AND the boxDownsample'd dstW x dstH planes of every
// source and blurred image, for SSIMFast(src, blurred) (ssim.go:48-70), in one pass over the
// pixels.  Returns FNX_NOOP without launching anything when the shape is outside what the
// SCORE kernel is built for (the caller then runs the two ops separately).
// Geometry of a one-pass launch: tile shape, box tables.  Cached on the ctx (batches repeat it).
constexpr int NHEAD = 260 + 257 * 8;   // uint32 words at the head of the table blob: magic[c] (c <= 256, padded to 260), tiedown[8 c + w]
static bool build_score_geom(fnx::ScoreGeom &g, int w, int h, int radius, int dstW, int dstH, int th_fixed = 0)
{
    // (th_fixed: the matrix-pipe kernels' tile, whose rows do not depend on the radius -- radii up to 24 since r5)
    if (radius < 1 || radius > (th_fixed ? 24 : SCORE_RMAX) || w < dstW || h < dstH || dstW <= 0 || dstH <= 0 || w >= (1 << 24) ||
        h >= (1 << 24))
        return false;
    const double xr = static_cast<double>(w) / static_cast<double>(dstW);   // ssim.go:251-252
    const double yr = static_cast<double>(h) / static_cast<double>(dstH);
    // Where the one-pass kernel wins (measured at steady clocks, tools/time_onepass.py, us per image
    // one-pass / two-call): ratio 3.1 (1600x1200) 10.1 / 9.6, 3.75 (1080p) 9.1 / 9.3, 4.3 (2200 px)
    // 11.5 / 12.1, 5 (1440p) 13.8 / 15.2, 7.5 (4K) 26.9 / 31.9, 15 (8K) 98.3 / 120.1.  Small boxes mean
    // many table entries per tile to zero, store and (no room for copies) contend on; below 3.6 the
    // two ops run back to back.  Upwards the limit is the 16-bit sum fields (boxes of <= 256 px).
    if (std::fmin(xr, yr) < 3.6) return false;
    // source column / row -> box index (boxes of a downscale are disjoint and ascending)
    // one table blob: magic[c], tiedown[c][8] (box_mean_u8) | BoxRef per output column, per output row |
    // box index of every source column, every source row
    const size_t ref_words = 4 * (static_cast<size_t>(dstW) + dstH);
    std::vector<int32_t> &map = g.map;
    map.assign(NHEAD + ref_words + static_cast<size_t>(w) + h, -1);
    {
        uint32_t *magic = reinterpret_cast<uint32_t *>(map.data()), *tiedown = magic + 260;
        std::fill(magic, magic + NHEAD, 0u);
        for (uint32_t c = 1; c <= 256; c++) {
            magic[c] = static_cast<uint32_t>((1ull << 32) / (2ull * c)) + 1u;
            if (c & 1u) continue;                                    // odd counts have no ties
            const double inv = 1.0 / static_cast<double>(c);         // inv := 1.0 / count (ssim.go:301)
            for (uint32_t k = 0; k < 256; k++) {
                const double n = static_cast<double>(c / 2) * static_cast<double>(2 * k + 1);   // n / c == k + 0.5
                if (n > 255.0 * c) break;
                if (n * inv < static_cast<double>(k) + 0.5) tiedown[8 * c + (k >> 5)] |= 1u << (k & 31);
            }
        }
    }
    BoxRef *xref = reinterpret_cast<BoxRef *>(map.data() + NHEAD), *yref = xref + dstW;
    int32_t *bx = map.data() + NHEAD + ref_words, *by = bx + w;
    int maxbw = 0, maxbh = 0;
    for (int d = 0; d < dstW; d++) {
        int s0, s1;
        box_edge(d, xr, w, s0, s1);
        if (d + 1 < dstW) { int n0, n1; box_edge(d + 1, xr, w, n0, n1); if (n0 < s1) return false; }
        for (int x = s0; x < s1; x++) bx[x] = d;
        if (s1 - s0 > maxbw) maxbw = s1 - s0;
    }
    for (int d = 0; d < dstH; d++) {
        int s0, s1;
        box_edge(d, yr, h, s0, s1);
        if (d + 1 < dstH) { int n0, n1; box_edge(d + 1, yr, h, n0, n1); if (n0 < s1) return false; }
        for (int y = s0; y < s1; y++) by[y] = d;
        if (s1 - s0 > maxbh) maxbh = s1 - s0;
    }
    if (static_cast<long>(maxbw) * maxbh > 256) return false;   // 16-bit sum fields
    // most boxes a tile touches, for the preferred tile shape and then the other one
    bool tall = g.tall_pref;
    int th = 0, nbx = 0, nby = 0;
    for (int attempt = 0; attempt < (th_fixed ? 1 : 2); attempt++, tall = !tall) {
        th = th_fixed ? th_fixed : direct_tile_rows(radius, tall);
        nbx = nby = 1;
        auto span = [](const int32_t *m, int lo, int hi) {   // boxes touched by [lo, hi)
            int first = -1, last = -1;
            for (int i = lo; i < hi; i++)
                if (m[i] >= 0) { if (first < 0) first = m[i]; last = m[i]; }
            return first < 0 ? 0 : last - first + 1;
        };
        for (int x0 = 0; x0 < w; x0 += 64) nbx = std::max(nbx, span(bx, x0, std::min(w, x0 + 64)));
        for (int y0 = 0; y0 < h; y0 += th) nby = std::max(nby, span(by, y0, std::min(h, y0 + th)));
        if ((nbx + 1) * (nby + 1) <= (th_fixed > 272 ? 2 * SCORE_NBOX : SCORE_NBOX)) break;   // (th_fixed > 272: the wide matrix kernel's long segments)
        th = 0;
    }
    if (!th) return false;
    tall = th == direct_tile_rows(radius, true);
    if (th_fixed) {
        // blur_mfma_kernel's indicator matrix has five box columns per 16-px group of a tile
        for (int x0 = 0; x0 < w; x0 += 16) {
            int first = -1, last = -1;
            for (int x = x0; x < std::min(w, x0 + 16); x++)
                if (bx[x] >= 0) { if (first < 0) first = bx[x]; last = bx[x]; }
            if (first >= 0 && last - first > 4) return false;
        }
    }
    // a tile whose first column / row is in no box must not hold boxed pixels (kernel uses bx[x0])
    for (int x0 = 0; x0 < w; x0 += 64)
        if (bx[x0] < 0) for (int x = x0; x < std::min(w, x0 + 64); x++) if (bx[x] >= 0) return false;
    for (int y0 = 0; y0 < h; y0 += th)
        if (by[y0] < 0) for (int y = y0; y < std::min(h, y0 + th); y++) if (by[y] >= 0) return false;

    const int slabn = (nbx + 1) * (nby + 1);
    const int tiles_x = (w + 63) / 64, tiles = tiles_x * ((h + th - 1) / th);
    if (static_cast<size_t>(tiles) * 2 * slabn >= (1u << 30)) return false;   // 32-bit slab offsets
    for (int d = 0; d < dstW; d++) {
        int s0, s1;
        box_edge(d, xr, w, s0, s1);
        const int t0 = s0 >> 6, t1 = (s1 - 1) >> 6;
        if (t1 > t0 + 1) return false;
        xref[d].part0 = t0 * 2 * slabn + (d - bx[t0 << 6]);
        xref[d].part1 = t1 > t0 ? t1 * 2 * slabn + (d - bx[t1 << 6]) : -1;
        xref[d].len = s1 - s0; xref[d].pad = 0;
    }
    for (int d = 0; d < dstH; d++) {
        int s0, s1;
        box_edge(d, yr, h, s0, s1);
        const int t0 = s0 / th, t1 = (s1 - 1) / th;
        if (t1 > t0 + 1) return false;
        yref[d].part0 = t0 * tiles_x * 2 * slabn + (d - by[t0 * th]) * (nbx + 1);
        yref[d].part1 = t1 > t0 ? t1 * tiles_x * 2 * slabn + (d - by[t1 * th]) * (nbx + 1) : -1;
        yref[d].len = s1 - s0; yref[d].pad = 0;
    }

    g.th = th; g.nbx = nbx; g.nby = nby; g.tall = tall;
    return true;
}

int launch_blur_scored(fnx_ctx *ctx, int n, const uint8_t *const *srcs, int sstride, int w, int h,
                       const double *kernel, int radius, int flags, uint8_t *const *dsts, int dstride,
                       uint8_t *planes, size_t plane, int dstW, int dstH)
{
    if (n > 65535) return FNX_NOOP;   // grid.z
    const bool exact = flags & FNX_BLUR_EXACT;
    // radii 7 .. 24 (r5): GaussianBlur alone runs on the matrix pipe (blur_mfma_wide_kernel), whose fast-mode bytes are not this
    // file's fp32 kernel's; the one-pass form must return what the two calls return, so it is that kernel's SCORE form or none
    const bool wide = radius > 6 && blur_mfma_wide_scored_covers(kernel, radius, w, h, exact);
    if (!wide) {
        if (radius < 1 || radius > SCORE_RMAX) return FNX_NOOP;   // before any tile arithmetic (TH would be <= 0)
        if (exact && !guard_kernel_ok(kernel, radius)) return FNX_NOOP;
        if (radius > 6 && blur_mfma_takes(kernel, radius, w, h, exact)) return FNX_NOOP;
    }
    const bool tall_pref = wide ? false : direct_tall(ctx, radius, n, w, h);
    // the matrix-pipe kernel (blur_mfma.hip) where its table and its box geometry fit; its tile is 64 px x seg rows
    static const int wide_cap = [] { const char *e = dev_env("FNX_MFMA_WIDE_SEG"); return e ? atoi(e) : 544; }();   // development: rows per workgroup
    const int seg = wide ? blur_mfma_segment(ctx, n, w, h, wide_cap, 2)
                         : (blur_mfma_covers(kernel, radius, w, h) && (!exact || blur_mfma_exact_enabled()) ? blur_mfma_segment(ctx, n, w, h, 272, 3) : 0);
    ScoreGeom &g = ctx->score_geom;
    if (!(g.w == w && g.h == h && g.dstW == dstW && g.dstH == dstH && g.radius == radius && g.tall_pref == tall_pref && g.seg == seg)) {
        g.w = w; g.h = h; g.dstW = dstW; g.dstH = dstH; g.radius = radius; g.tall_pref = tall_pref; g.seg = seg;
        g.mfma = seg > 0 && build_score_geom(g, w, h, radius, dstW, dstH, seg);
        // where GaussianBlur alone runs on the matrix pipe but its one-pass geometry does not fit (boxes under 4 px: long side
        // 1843..2047), this file's fp32 kernel would return fast-mode bytes the two calls do not: the two calls take the step
        // (tools/fuzz_blur.py: 72 such shapes in 6 528)
        g.ok = g.mfma || (!wide && seg == 0 && build_score_geom(g, w, h, radius, dstW, dstH));
    }
    if (!g.ok) return FNX_NOOP;
    const std::vector<int32_t> &map = g.map;
    const int th = g.th, nbx = g.nbx, nby = g.nby;
    const bool tall = g.tall;
    const size_t ref_words = 4 * (static_cast<size_t>(dstW) + dstH);
    const int slabn = (nbx + 1) * (nby + 1);
    const int tiles_x = (w + 63) / 64, tiles = tiles_x * ((h + th - 1) / th);

    void *dmap = nullptr;
    FNX_TRY(upload_table(ctx, SLOT_BOXMAP, map.data(), sizeof(int32_t) * map.size(), &dmap));
    FusedArgs fa{};
    fa.srcs = srcs; fa.dsts = dsts;
    fa.sstride = sstride; fa.dstride = dstride; fa.w = w; fa.h = h;
    for (int i = 0; i < 2 * radius + 1 && !wide; i++) {
        fa.wt[i] = static_cast<float>(kernel[i]);
        fa.wd[i] = kernel[i];
    }
    fa.bx = static_cast<const int32_t *>(dmap) + NHEAD + ref_words; fa.by = fa.bx + w;
    fa.nbx = nbx; fa.nby = nby;
    void *slabs = nullptr;
    FNX_TRY(scratch(ctx, ctx->parity ? SLOT_SLABS1 : SLOT_SLABS, sizeof(unsigned long long) * 2 * slabn * static_cast<size_t>(tiles) * n, &slabs));
    fa.slabs = static_cast<unsigned long long *>(slabs);
    int st;
    if (g.mfma) {
        st = wide ? launch_blur_mfma_wide_scored(ctx, n, srcs, sstride, w, h, kernel, radius, flags, dsts, dstride, fa.bx, fa.by, fa.slabs, nbx, nby, th)
                  : launch_blur_mfma_scored(ctx, n, srcs, sstride, w, h, kernel, radius, flags, dsts, dstride, fa.bx, fa.by, fa.slabs, nbx, nby, th);
        if (st == FNX_NOOP) return FNX_NOOP;   // (blur_mfma_covers said yes: not reached) the caller runs the two ops back to back
    } else {
        st = exact ? launch_direct_radius<true, true>(ctx, radius, n, fa, tall)
                   : launch_direct_radius<true, false>(ctx, radius, n, fa, tall);
    }
    if (st < 0) return st;
    // the rest of the step runs on the ctx's second stream, behind this blur (api.cpp: one-pass enqueue) -- or, for a blur that
    // only KEEPS the box planes for the scoring call that follows (FNX_BLUR_KEEP_BOX_SUMS), right behind it on the same stream:
    // there is no next blur to hide a tail under, and the two cross-stream hand-overs cost a per-image pair 25 us
    hipStream_t box_stream = ctx->stream2;
    if (ctx->boxes_on_main) box_stream = ctx->stream;
    else {
        FNX_HIP(hipStreamWaitEvent(ctx->stream2, ctx->blur_done, 0));   // bound to the blur dispatch: no packet on `stream`
        ctx->stream2_used = true;
    }

    SlabArgs sa{};
    sa.slabs = fa.slabs; sa.dst = planes; sa.plane = plane;
    sa.magic = static_cast<const uint32_t *>(dmap);
    sa.tiedown = sa.magic + 260;
    sa.xref = reinterpret_cast<const BoxRef *>(static_cast<const int32_t *>(dmap) + NHEAD);
    sa.yref = sa.xref + dstW;
    sa.image_slabs = static_cast<size_t>(tiles) * 2 * slabn;
    sa.n = n; sa.dstW = dstW; sa.dstH = dstH; sa.slabn = slabn;
    hipLaunchKernelGGL(box_from_slabs_kernel, dim3((dstW + 64 * BFS_BOXES - 1) / (64 * BFS_BOXES), (dstH + 3) / 4, n), dim3(256), 0,
                       box_stream, sa);
    FNX_HIP(hipGetLastError());
    return FNX_OK;
}

template <typename T>
static int launch_generic(fnx_ctx *ctx, int n, const uint8_t *src, const uint8_t *const *srcs,
                          int sstride, int w, int h, const double *kernel, int radius,
                          uint8_t *dst, uint8_t *const *dsts, int dstride)
{
    const int nt = 2 * radius + 1;
    note_route(ctx, FNX_PROF_MAIN, sizeof(T) == 8 ? "blur_pass_kernel<double> x 2" : "blur_pass_kernel<float> x 2");
    std::vector<T> hk(nt);
    for (int i = 0; i < nt; i++) hk[i] = static_cast<T>(kernel[i]);
    void *dk = nullptr;
    FNX_TRY(upload_table(ctx, SLOT_TABLE0, hk.data(), sizeof(T) * nt, &dk));
    // uint8 intermediate (effects.go:168): tight, one per batched image
    const int tpitch = pitch16(w);
    const size_t timg = static_cast<size_t>(tpitch) * h;
    void *tmp = nullptr;
    FNX_TRY(scratch(ctx, SLOT_TMP0, timg * n + 16, &tmp));
    const uint8_t *const *tmps = nullptr;
    if (srcs) {   // device array of tmp pointers for the batched form
        std::vector<const uint8_t *> hp(n);
        for (int i = 0; i < n; i++) hp[i] = static_cast<const uint8_t *>(tmp) + timg * i;
        void *dp = nullptr;
        FNX_TRY(upload_table(ctx, SLOT_TABLE1, hp.data(), sizeof(void *) * n, &dp));
        tmps = static_cast<const uint8_t *const *>(dp);
    }
    dim3 grid((w + 63) / 64, (h + 3) / 4, n);
    PassArgs<T> ha{};
    ha.src = src; ha.srcs = srcs; ha.alpha = src; ha.alphas = srcs;
    ha.dst = static_cast<uint8_t *>(tmp); ha.dsts = const_cast<uint8_t *const *>(reinterpret_cast<const uint8_t *const *>(tmps));
    ha.sstride = sstride; ha.astride = sstride; ha.dstride = tpitch;
    ha.w = w; ha.h = h; ha.radius = radius; ha.kern = static_cast<const T *>(dk);
    hipLaunchKernelGGL((blur_pass_kernel<T, false>), grid, dim3(256), 0, ctx->stream, ha);
    FNX_HIP(hipGetLastError());
    PassArgs<T> va = ha;
    va.src = static_cast<const uint8_t *>(tmp); va.srcs = tmps; va.sstride = tpitch;
    va.dst = dst; va.dsts = dsts; va.dstride = dstride;
    hipLaunchKernelGGL((blur_pass_kernel<T, true>), grid, dim3(256), 0, ctx->stream, va);
    FNX_HIP(hipGetLastError());
    return FNX_OK;
}

// ------------------------------------------------------------------------------------
// exact mode for radii <= 8: the direct kernel's shape in fp64
// ------------------------------------------------------------------------------------
// Same tile scheme as blur_direct_kernel (H pass fed from global memory into the uint8 LDS
// intermediate with the source alpha in byte 3, one barrier, V pass from LDS), but every
// accumulator is a double updated with an UNFUSED multiply and add, taps in ascending order, and the
// rounding is clampF_dev -- the reference's arithmetic (effects.go:169-217), so the output is
// bit-exact.  An H item is 2 rows x 4 outputs, a V item 1 column x Q rows: the register budget
// (2 VGPRs per accumulator) halves the fast kernel's item sizes.  256 lanes, 64 x 4Q tile.
struct ExactArgs {
    const uint8_t *src;
    const uint8_t *const *srcs;
    uint8_t *dst;
    uint8_t *const *dsts;
    int sstride, dstride, w, h;
    int tiles_x, tiles;
    double wt[2 * FUSED_RMAX + 1];
};

template <int R>
__global__ __launch_bounds__(256) void blur_exact_kernel(ExactArgs a)
{
    constexpr int TW = 64, NTH = 256, RG = 4;
    constexpr int Q = (64 - 2 * R) / RG;            // output rows per V item
    constexpr int TH = Q * RG, SR = TH + 2 * R;     // SR <= 64
    constexpr int NT = 2 * R + 1;
    constexpr int HO = 4, NPX = HO + 2 * R, NV = (NPX + 3) / 4;
    constexpr int HGROUPS = TW / HO;                // 16
    constexpr int HITEMS = ((SR + 1) / 2) * HGROUPS;
    __shared__ __attribute__((aligned(16))) uint32_t s_tmp[(SR + 1) * TW];

    const int tile = xcd_tile(blockIdx.x, a.tiles);
    if (tile < 0) return;
    const int z = blockIdx.y;
    const uint8_t *src = a.srcs ? a.srcs[z] : a.src;
    uint8_t *dst = a.dsts ? a.dsts[z] : a.dst;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int x0 = tx * TW, y0 = ty * TH;
    const int tid = threadIdx.x;

    // ---- horizontal pass (effects.go:169-191) ----
    for (int item = tid; item < HITEMS; item += NTH) {
        const int rp = item / HGROUPS, g = item - rp * HGROUPS;
        const int xs = x0 + HO * g - R;
        const uint8_t *p0 = src + static_cast<size_t>(clampi(y0 - R + 2 * rp, 0, a.h - 1)) * a.sstride;
        const uint8_t *p1 = src + static_cast<size_t>(clampi(y0 - R + 2 * rp + 1, 0, a.h - 1)) * a.sstride;
        u32x4 t0[NV], t1[NV];
#pragma unroll
        for (int q = 0; q < NV; q++) {
            const int x = xs + 4 * q;
            if (x >= 0 && x + 3 < a.w) {
                t0[q] = *(g_u32x4 *)(p0 + 4 * static_cast<size_t>(x));
                t1[q] = *(g_u32x4 *)(p1 + 4 * static_cast<size_t>(x));
            } else {   // clamp-to-edge (effects.go:174-178)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int xc = clampi(x + e, 0, a.w - 1);
                    t0[q][e] = ld_px(p0, xc);
                    t1[q][e] = ld_px(p1, xc);
                }
            }
        }
        double acc0[HO][3], acc1[HO][3];
#pragma unroll
        for (int j = 0; j < HO; j++)
#pragma unroll
            for (int c = 0; c < 3; c++) acc0[j][c] = acc1[j][c] = 0.0;
#pragma unroll
        for (int i = 0; i < NPX; i++) {
            const uint32_t q0 = t0[i / 4][i % 4], q1 = t1[i / 4][i % 4];
            double f0[3], f1[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                f0[c] = u8_to_f64((q0 >> (8 * c)) & 0xffu);
                f1[c] = u8_to_f64((q1 >> (8 * c)) & 0xffu);
            }
#pragma unroll
            for (int j = 0; j < HO; j++) {
                const int k = i - j;
                if (k >= 0 && k < NT) {
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        acc0[j][c] = acc0[j][c] + f0[c] * a.wt[k];   // r += float64(pix) * wt, unfused (TU: -ffp-contract=off)
                        acc1[j][c] = acc1[j][c] + f1[c] * a.wt[k];
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < HO; j++) {
            const int c = j + R;
            const uint32_t s0 = t0[c / 4][c % 4], s1 = t1[c / 4][c % 4];
            s_tmp[(2 * rp) * TW + HO * g + j] = clampF_dev(acc0[j][0]) | (clampF_dev(acc0[j][1]) << 8) |
                                                (clampF_dev(acc0[j][2]) << 16) | (s0 & 0xff000000u);
            s_tmp[(2 * rp + 1) * TW + HO * g + j] = clampF_dev(acc1[j][0]) | (clampF_dev(acc1[j][1]) << 8) |
                                                    (clampF_dev(acc1[j][2]) << 16) | (s1 & 0xff000000u);
        }
    }
    __syncthreads();

    // ---- vertical pass (effects.go:195-217): item = 1 column x Q output rows ----
    {
        const int col = tid & 63, rg = tid >> 6;
        const int x = x0 + col;
        const uint32_t *colp = s_tmp + (rg * Q) * TW + col;
        double acc[Q][3];
#pragma unroll
        for (int j = 0; j < Q; j++) acc[j][0] = acc[j][1] = acc[j][2] = 0.0;
        uint32_t al[Q];
#pragma unroll
        for (int i = 0; i < Q + 2 * R; i++) {
            const uint32_t t = colp[i * TW];
            if (i >= R && i < R + Q) al[i - R] = t;
            double f[3];
#pragma unroll
            for (int c = 0; c < 3; c++) f[c] = u8_to_f64((t >> (8 * c)) & 0xffu);
#pragma unroll
            for (int j = 0; j < Q; j++) {
                const int k = i - j;
                if (k >= 0 && k < NT) {
#pragma unroll
                    for (int c = 0; c < 3; c++) acc[j][c] = acc[j][c] + f[c] * a.wt[k];
                }
            }
        }
        if (x < a.w) {
#pragma unroll
            for (int j = 0; j < Q; j++) {
                const int y = y0 + rg * Q + j;
                if (y < a.h)
                    *(g_u32w *)(dst + static_cast<size_t>(y) * a.dstride + 4 * static_cast<size_t>(x)) =
                        clampF_dev(acc[j][0]) | (clampF_dev(acc[j][1]) << 8) | (clampF_dev(acc[j][2]) << 16) | (al[j] & 0xff000000u);
            }
        }
    }
}

template <int R>
static int launch_exact(fnx_ctx *ctx, int n, ExactArgs &ea)
{
    constexpr int TH = ((64 - 2 * R) / 4) * 4;
    ea.tiles_x = (ea.w + 63) / 64;
    ea.tiles = ea.tiles_x * ((ea.h + TH - 1) / TH);
    note_route(ctx, FNX_PROF_MAIN, "blur_exact_kernel");
    hipLaunchKernelGGL((blur_exact_kernel<R>), dim3(8 * ((ea.tiles + 7) / 8), n), dim3(256), 0, ctx->stream, ea);
    FNX_HIP(hipGetLastError());
    return FNX_OK;
}

int launch_blur(fnx_ctx *ctx, int n, const uint8_t *src, const uint8_t *const *srcs, int sstride,
                int w, int h, const double *kernel, int radius, int flags, uint8_t *dst,
                uint8_t *const *dsts, int dstride)
{
    if (w <= 0 || h <= 0 || n <= 0) return FNX_OK;
    {   // radius <= 6, non-negative weights summing to 1: both passes on the matrix pipe (blur_mfma.hip), fast or exact
        const int st = launch_blur_mfma(ctx, n, src, srcs, sstride, w, h, kernel, radius, flags, dst, dsts, dstride);
        if (st != FNX_NOOP) return st;
    }
    if (flags & FNX_BLUR_EXACT) {
        if (radius < 1 || radius > FUSED_RMAX)
            return launch_generic<double>(ctx, n, src, srcs, sstride, w, h, kernel, radius, dst, dsts, dstride);
        // any caller-supplied table the guarded kernel's bound does not cover takes the fp64 kernel
        if (guard_kernel_ok(kernel, radius)) {
            FusedArgs ga{};
            ga.src = src; ga.srcs = srcs; ga.dst = dst; ga.dsts = dsts;
            ga.sstride = sstride; ga.dstride = dstride; ga.w = w; ga.h = h;
            for (int i = 0; i < 2 * radius + 1; i++) {
                ga.wt[i] = static_cast<float>(kernel[i]);
                if (radius <= SCORE_RMAX) ga.wd[i] = kernel[i];
            }
            if (radius > SCORE_RMAX) {
                void *dk = nullptr;
                FNX_TRY(upload_table(ctx, SLOT_TABLE1, kernel, sizeof(double) * (2 * radius + 1), &dk));
                ga.wdp = static_cast<const double *>(dk);
            }
            return launch_direct_radius<false, true>(ctx, radius, n, ga, direct_tall(ctx, radius, n, w, h));
        }
        if (radius > SCORE_RMAX)              // blur_exact_kernel is built for R <= 8
            return launch_generic<double>(ctx, n, src, srcs, sstride, w, h, kernel, radius, dst, dsts, dstride);
        ExactArgs ea{};
        ea.src = src; ea.srcs = srcs; ea.dst = dst; ea.dsts = dsts;
        ea.sstride = sstride; ea.dstride = dstride; ea.w = w; ea.h = h;
        for (int i = 0; i < 2 * radius + 1; i++) ea.wt[i] = kernel[i];
        switch (radius) {
        case 1: return launch_exact<1>(ctx, n, ea);
        case 2: return launch_exact<2>(ctx, n, ea);
        case 3: return launch_exact<3>(ctx, n, ea);
        case 4: return launch_exact<4>(ctx, n, ea);
        case 5: return launch_exact<5>(ctx, n, ea);
        case 6: return launch_exact<6>(ctx, n, ea);
        case 7: return launch_exact<7>(ctx, n, ea);
        case 8: return launch_exact<8>(ctx, n, ea);
        }
    }
    if (radius < 1 || radius > FUSED_RMAX) {
        // what the matrix kernels leave (images under 64 x 32, radii past 62, a weight >= 0.49).  An fp32 accumulator's rounding grows
        // with the tap count -- tools/fuzz_blur.py, seed 71: 0.4 % of the samples one LSB off at 127 taps, past the fast mode's
        // 0.1 % -- so beyond 53 taps the fast mode takes the fp64 passes as well (the reference's own arithmetic, 28 % slower)
        if (radius > 26) return launch_generic<double>(ctx, n, src, srcs, sstride, w, h, kernel, radius, dst, dsts, dstride);
        return launch_generic<float>(ctx, n, src, srcs, sstride, w, h, kernel, radius, dst, dsts, dstride);
    }
    FusedArgs fa{};
    fa.src = src; fa.srcs = srcs; fa.dst = dst; fa.dsts = dsts;
    fa.sstride = sstride; fa.dstride = dstride; fa.w = w; fa.h = h;
    for (int i = 0; i < 2 * radius + 1; i++) fa.wt[i] = static_cast<float>(kernel[i]);
    return launch_direct_radius<false>(ctx, radius, n, fa, direct_tall(ctx, radius, n, w, h));
}

}  // namespace fnx
