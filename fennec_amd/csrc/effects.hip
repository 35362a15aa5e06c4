// gaussianBlur3x3 / Sharpen / AdaptiveSharpen (effects.go:10-141) on gfx950: bit-exact, one launch per op.
//
//  * gaussianBlur3x3 is exact in integers: clampF(sum / 16.0) == (sum + 8) >> 4 for the binomial sums.
//  * Sharpen: val = orig + amount * (orig - blur) (effects.go:37) depends on (orig, d = orig - blur) only, and
//    clampF(fl(orig + fl(amount * d))) == clamp(orig + R[d]) with R[d] = floor(fl(amount * d) + 0.5): the sum's
//    fp64 rounding (<= 2^-44) can only matter when fl(amount * d) sits within that of a half-integer without
//    being one, which the host checks for all 511 values of d while it builds R (it then takes the fp64 kernel).
//    Exact ties are exact sums and round away from zero = up for every value that survives the clamp.
//  * AdaptiveSharpen (effects.go:49-112): val = orig + amount * e * (orig - blur), e = min(1, |Sobel| / 400) on
//    BT.601 luminance.  The integer milli-luminance I = 299 R + 587 G + 114 B makes the Sobel sums EXACT
//    integers; the rest runs in fp32 under a rounding guard (the technique of blur.hip's GUARD kernels): the
//    accumulator is within E of the real value, it is packed at acc and at acc + 2G (G > E + the fp64 chain's
//    own error), and a sample whose two bytes differ -- a rounding boundary within G -- is recomputed in fp64,
//    in the reference's operation order, from the staged tile.  Proven, not sampled; see fx_guard().
//
// Two forms of the same arithmetic.  fx_stream_kernel (round 3, what every image takes): a WAVE marches down a strip of
// 62 output columns with six rows of loads in flight, neighbours through a per-wave LDS row, the last three rows'
// horizontal sums in a register ring -- no tile, no barrier in the loop (see its own header below).  fx_march_kernel
// (round 2, kept for destinations past the 32-bit offsets and as the A/B): 64 x 24 outputs per 256-lane workgroup; the
// 66 x 26 source tile is staged once as two words per pixel with the channels spread into 16-bit fields
// (R | B << 16, G | A << 16) plus I; a lane then owns one column and walks down 6 output rows: the horizontal [1 2 1]
// sums, the Sobel column differences and row sums are computed once per tile row and shared by the three output rows
// they feed.
//
// fx_ref_kernel (the round-1 kernel: fp64, the reference's operation order) remains for amounts outside the
// guard's bound and for Sharpen amounts with a near-tie product.
#include <type_traits>
#include "common.hpp"
#include "devutil.hpp"

#include <cmath>
#include <cstdlib>

namespace fnx {

enum FxMode { FX_BLUR3 = 0, FX_SHARPEN = 1, FX_ADAPTIVE = 2 };

struct FxArgs {
    const uint8_t *src;
    uint8_t *dst;
    int sstride, dstride, w, h;
    double amount;
    // new kernels
    const int32_t *rtab;   // device: R[d + 255], d in [-255, 255] (+ one pad entry); Sharpen, AdaptiveSharpen saturated pixels
    float amt32, k32;      // AdaptiveSharpen: (float)amount, (float)(amount / 400000)
    float guard;           // G
    float flag_thr;        // AdaptiveSharpen, paired-row path: a sample with fract(acc) >= flag_thr has an integer within 2G above it
    int use_table;         // AdaptiveSharpen: surely saturated pixels (e == 1) take R instead of the guard
    int vec_ok;            // src base and stride 16-byte aligned
    int strips, segs, seg_rows;   // streaming form: per image strips x segs wave-sized items of seg_rows output rows
    int pairs;             // AdaptiveSharpen: interior tiles take the paired-row form (FNX_FX_PAIRS=0: the one-row form, A/B)
    // a batch of same-geometry tight images (fx_stream_kernel's blockIdx.y = image): device arrays of their pointers
    const uint8_t *const *srcs;
    uint8_t *const *dsts;
};

// ------------------------------------------------------------------------------------
// reference-order fp64 kernel (round 1): 64 x 8 outputs per workgroup
// ------------------------------------------------------------------------------------
constexpr int FXR_TW = 64, FXR_TH = 8;

// one interior pixel exactly as effects.go:70-87 / 28-42 computes it, from spread-field tile words:
// c_* = centre, n_*[9] = the 3 x 3 neighbourhood row-major (MODE ADAPTIVE reads all, SHARPEN the blur only)
template <int MODE>
__device__ __forceinline__ uint32_t fx_exact_px(const uint32_t (&nrb)[9], const uint32_t (&nga)[9], double amount)
{
    const uint32_t crb = nrb[4], cga = nga[4];
    const uint32_t c = crb | (cga << 8);
    const uint32_t srb = (nrb[0] + nrb[2] + nrb[6] + nrb[8]) + 2 * (nrb[1] + nrb[3] + nrb[5] + nrb[7]) + 4 * crb + 0x00080008u;
    const uint32_t sga = (nga[0] + nga[2] + nga[6] + nga[8]) + 2 * (nga[1] + nga[3] + nga[5] + nga[7]) + 4 * cga + 0x00080008u;
    const uint32_t blur[3] = {(srb >> 4) & 0xffu, (sga >> 4) & 0xffu, (srb >> 20) & 0xffu};
    if (MODE == FX_BLUR3) return blur[0] | (blur[1] << 8) | (blur[2] << 16) | (c & 0xff000000u);
    double amt = amount;
    if (MODE == FX_ADAPTIVE) {                                   // localEdgeStrength, effects.go:93-112
        double l[9];
#pragma unroll
        for (int i = 0; i < 9; i++) l[i] = lum601(nrb[i] | (nga[i] << 8));
        const double gx = -l[0] + l[2] - 2 * l[3] + 2 * l[5] - l[6] + l[8];
        const double gy = -l[0] - 2 * l[1] - l[2] + l[6] + 2 * l[7] + l[8];
        const double mag = sqrt(gx * gx + gy * gy);
        double normalized = mag / 400.0;
        if (normalized > 1) normalized = 1;
        amt = amount * normalized;                               // localAmount (effects.go:74)
    }
    uint32_t out = c & 0xff000000u;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const double orig = u8_to_f64((c >> (8 * ch)) & 0xffu);
        const double bl = u8_to_f64(blur[ch]);
        const double val = orig + amt * (orig - bl);             // effects.go:37,82
        out |= clampF_dev(val) << (8 * ch);
    }
    return out;
}

template <int MODE>
__global__ __launch_bounds__(256) void fx_ref_kernel(FxArgs a)
{
    constexpr int LW = FXR_TW + 2, LH = FXR_TH + 2;
    __shared__ uint32_t s_rb[LH * LW], s_ga[LH * LW];
    const int x0 = blockIdx.x * FXR_TW, y0 = blockIdx.y * FXR_TH;
    const int tid = threadIdx.x;
    for (int i = tid; i < LH * LW; i += 256) {
        const int ly = i / LW, lx = i - ly * LW;
        // clamped reads: out-of-image tile cells are only ever neighbours of border pixels, which
        // are copies of the source and never look at them
        const int x = clampi(x0 + lx - 1, 0, a.w - 1), y = clampi(y0 + ly - 1, 0, a.h - 1);
        const uint32_t p = ld_px(a.src + static_cast<size_t>(y) * a.sstride, x);
        s_rb[i] = p & 0x00ff00ffu;
        s_ga[i] = (p >> 8) & 0x00ff00ffu;
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
        uint32_t out = s_rb[ci] | (s_ga[ci] << 8);      // borders and alpha are copies of the source (effects.go:68,120)
        if (x >= 1 && y >= 1 && x < a.w - 1 && y < a.h - 1) {
            uint32_t nrb[9], nga[9];
#pragma unroll
            for (int j = 0; j < 3; j++)
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    nrb[3 * j + i] = s_rb[ci + (j - 1) * LW + i - 1];
                    nga[3 * j + i] = s_ga[ci + (j - 1) * LW + i - 1];
                }
            out = fx_exact_px<MODE>(nrb, nga, a.amount);
        }
        *(g_u32w *)(a.dst + static_cast<size_t>(y) * a.dstride + 4 * static_cast<size_t>(x)) = out;
    }
}

// ------------------------------------------------------------------------------------
// column-marching kernel
// ------------------------------------------------------------------------------------
constexpr int FX_TW = 64, FX_TH = 24, FX_RP = 6;       // tile, output rows per lane
constexpr int FX_LW = 72, FX_LH = FX_TH + 2;           // tile row pitch in words: image column x0 + i lives at word 4 + i,
                                                       // so that the 64-column body is 16-byte aligned; halos at 3 and 68
constexpr int FX_FIX_CAP = 512;

// 299 R + 587 G + 114 B: exact integer milli-luminance (three v_dot4_u32_u8; see ssim.hip)
__device__ __forceinline__ uint32_t lum_milli_u32(uint32_t p)
{
    uint32_t i = __builtin_amdgcn_udot4(p, 0x00004d00u, 0u, false);
    i = __builtin_amdgcn_udot4(p, 0x0000ff2cu, i, false);
    return __builtin_amdgcn_udot4(p, 0x0072ffffu, i, false);
}

// byte N of a word as a float (v_cvt_f32_ubyteN)
template <int N>
__device__ __forceinline__ float ubyte_f32(uint32_t w)
{
    float f;
    if (N == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(f) : "v"(w));
    else if (N == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(f) : "v"(w));
    else if (N == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(w));
    else asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(f) : "v"(w));
    return f;
}

// G | A << 16 of a pixel in one v_perm_b32 (the R | B << 16 word is one v_and_b32)
__device__ __forceinline__ uint32_t ga_fields(uint32_t p) { return __builtin_amdgcn_perm(0u, p, 0x0c030c01u); }

// AdaptiveSharpen is occupancy-bound (a workgroup is load tile -> barrier -> compute -> store, and what hides one
// phase is other workgroups): 6 per CU -- <= 80 VGPRs, <= 26 KB of LDS each (16-bit table and fix-up list)
template <int MODE>
__global__ __launch_bounds__(256, MODE == FX_ADAPTIVE ? 6 : 1) void fx_march_kernel(FxArgs a)
{
    constexpr int LW = FX_LW, LH = FX_LH;
    __shared__ __attribute__((aligned(16))) uint32_t s_rb[LH * LW], s_ga[LH * LW];
    __shared__ __attribute__((aligned(16))) uint32_t s_lum[MODE == FX_ADAPTIVE ? LH * LW : 4];
    __shared__ int16_t s_tab[MODE != FX_BLUR3 ? 512 : 2];                // |R[d]| <= 64 * 255 (build_rtab)
    __shared__ uint16_t s_fix[MODE == FX_ADAPTIVE ? FX_FIX_CAP : 2];
    __shared__ int s_nfix;
    const int x0 = blockIdx.x * FX_TW, y0 = blockIdx.y * FX_TH;
    const int tid = threadIdx.x;
    if (MODE == FX_ADAPTIVE && tid == 0) s_nfix = 0;
    if (MODE == FX_SHARPEN || (MODE == FX_ADAPTIVE && a.use_table)) {
        s_tab[tid] = static_cast<int16_t>(a.rtab[tid]);
        s_tab[256 + tid] = static_cast<int16_t>(a.rtab[256 + tid]);
    }
    auto put = [&](int cell, uint32_t p) {
        s_rb[cell] = p & 0x00ff00ffu;
        s_ga[cell] = ga_fields(p);
        if constexpr (MODE == FX_ADAPTIVE) s_lum[cell] = lum_milli_u32(p);
    };
    // ---- stage the (TH + 2) x (TW + 2) source tile: clamped reads -- out-of-image cells are only ever
    // neighbours of border pixels, which are copies of the source and never look at them
    if (a.vec_ok && x0 + FX_TW <= a.w) {
        // all of a lane's 16-byte loads are issued before the first is used (indices clamped, no branch around
        // them): one memory latency per tile instead of one per trip
        constexpr int NITEM = LH * (FX_TW / 4), TRIPS = (NITEM + 255) / 256;
        u32x4 v[TRIPS];
#pragma unroll
        for (int k = 0; k < TRIPS; k++) {
            const int i = min(tid + 256 * k, NITEM - 1);
            const int ly = i >> 4, c = i & 15;
            const int y = clampi(y0 + ly - 1, 0, a.h - 1);
            v[k] = *(g_u32x4 *)(a.src + static_cast<size_t>(y) * a.sstride + 4 * static_cast<size_t>(x0 + 4 * c));
        }
        uint32_t halo = 0;
        if (tid < 2 * LH) {                                      // halo columns x0 - 1 and x0 + 64
            const int ly = tid >> 1, side = tid & 1;
            const int y = clampi(y0 + ly - 1, 0, a.h - 1);
            const int x = clampi(side ? x0 + FX_TW : x0 - 1, 0, a.w - 1);
            halo = ld_px(a.src + static_cast<size_t>(y) * a.sstride, x);
        }
#pragma unroll
        for (int k = 0; k < TRIPS; k++) {
            const int i = tid + 256 * k;
            if (i < NITEM) {
                const int ly = i >> 4, c = i & 15;
                const int cell = ly * LW + 4 + 4 * c;
                *reinterpret_cast<u32x4 *>(&s_rb[cell]) = (u32x4){v[k][0] & 0x00ff00ffu, v[k][1] & 0x00ff00ffu, v[k][2] & 0x00ff00ffu, v[k][3] & 0x00ff00ffu};
                *reinterpret_cast<u32x4 *>(&s_ga[cell]) = (u32x4){ga_fields(v[k][0]), ga_fields(v[k][1]), ga_fields(v[k][2]), ga_fields(v[k][3])};
                if constexpr (MODE == FX_ADAPTIVE)
                    *reinterpret_cast<u32x4 *>(&s_lum[cell]) = (u32x4){lum_milli_u32(v[k][0]), lum_milli_u32(v[k][1]), lum_milli_u32(v[k][2]), lum_milli_u32(v[k][3])};
            }
        }
        if (tid < 2 * LH) put((tid >> 1) * LW + ((tid & 1) ? 4 + FX_TW : 3), halo);
    } else {
        for (int i = tid; i < LH * (FX_TW + 2); i += 256) {
            const int ly = i / (FX_TW + 2), lx = i - ly * (FX_TW + 2);
            const int x = clampi(x0 + lx - 1, 0, a.w - 1), y = clampi(y0 + ly - 1, 0, a.h - 1);
            put(ly * LW + 3 + lx, ld_px(a.src + static_cast<size_t>(y) * a.sstride, x));
        }
    }
    __syncthreads();

    const int cx = tid & 63, rg = tid >> 6;
    const int x = x0 + cx;
    const int base = (rg * FX_RP) * LW + 4 + cx;                 // tile cell of (x, first output row - 1)
    // per tile row: horizontal [1 2 1] of both field words; Sobel column difference and [1 2 1] row sum of I
    uint32_t hrb[FX_RP + 2], hga[FX_RP + 2];
    int32_t dxr[MODE == FX_ADAPTIVE ? FX_RP + 2 : 1], sxr[MODE == FX_ADAPTIVE ? FX_RP + 2 : 1];
    auto tile_row = [&](int r) {
        const int c = base + r * LW;
        hrb[r] = s_rb[c - 1] + s_rb[c + 1] + 2 * s_rb[c];
        hga[r] = s_ga[c - 1] + s_ga[c + 1] + 2 * s_ga[c];
        if constexpr (MODE == FX_ADAPTIVE) {
            const int32_t l = static_cast<int32_t>(s_lum[c - 1]), m = static_cast<int32_t>(s_lum[c]), rr = static_cast<int32_t>(s_lum[c + 1]);
            dxr[r] = rr - l;
            sxr[r] = l + rr + 2 * m;
        }
    };
    const float seed = 0.5f - a.guard, g2 = 2.0f * a.guard;
    const uint32_t doff0 = static_cast<uint32_t>(y0 + rg * FX_RP) * static_cast<uint32_t>(a.dstride) + 4u * static_cast<uint32_t>(x);
    // tiles that touch no image border (all but the frame) skip the per-pixel edge tests
    const bool interior = x0 >= 1 && y0 >= 1 && x0 + FX_TW < a.w && y0 + FX_TH < a.h;
    auto rows = [&](auto tag) {
        constexpr bool INTERIOR = decltype(tag)::value;
#pragma unroll
        for (int r = 0; r < FX_RP + 2; r++) tile_row(r);
#pragma unroll
        for (int j = 0; j < FX_RP; j++) {
            const int y = y0 + rg * FX_RP + j;
            const int ci = base + (j + 1) * LW;
            const uint32_t crb = s_rb[ci], cga = s_ga[ci];
            const uint32_t c = crb | (cga << 8);
            uint32_t out = c;                                    // borders and alpha are copies of the source (effects.go:68,120)
            bool flagged = false;
            if (INTERIOR || (x >= 1 && y >= 1 && x < a.w - 1 && y < a.h - 1)) {
                // [1 2 1] vertically over the horizontal sums; (sum + 8) >> 4 per field (effects.go:125-135)
                const uint32_t srb = hrb[j] + hrb[j + 2] + 2 * hrb[j + 1] + 0x00080008u;
                const uint32_t sga = hga[j] + hga[j + 2] + 2 * hga[j + 1] + 0x00080008u;
                if constexpr (MODE == FX_BLUR3) {
                    const uint32_t br = (srb >> 4) & 0xffu, bg = (sga >> 4) & 0xffu, bb = (srb >> 20) & 0xffu;
                    out = br | (bg << 8) | (bb << 16) | (c & 0xff000000u);
                } else {
                    // table form: clamp(orig + R[d]) on integers
                    auto tab = [&]() {
                        const int br = (srb >> 4) & 0xffu, bg = (sga >> 4) & 0xffu, bb = (srb >> 20) & 0xffu;
                        const int o_r = crb & 0xffu, o_g = cga & 0xffu, o_b = (crb >> 16) & 0xffu;
                        const int vr = clampi(o_r + s_tab[o_r - br + 255], 0, 255), vg = clampi(o_g + s_tab[o_g - bg + 255], 0, 255),
                                  vb = clampi(o_b + s_tab[o_b - bb + 255], 0, 255);
                        return static_cast<uint32_t>(vr) | (static_cast<uint32_t>(vg) << 8) | (static_cast<uint32_t>(vb) << 16) | (c & 0xff000000u);
                    };
                    if constexpr (MODE == FX_SHARPEN) {
                        out = tab();
                    } else {
                        // Sobel on I (exact integers, |g| <= 4 * 255000 < 2^24: the converts are exact)
                        const float gx = static_cast<float>(dxr[j] + dxr[j + 2] + 2 * dxr[j + 1]);
                        const float gy = static_cast<float>(sxr[j + 2] - sxr[j]);
                        const float m2 = fmaf(gy, gy, gx * gx);
                        const float t = fminf(a.amt32, __builtin_amdgcn_sqrtf(m2) * a.k32);   // amount * e, within 4e-7 relative
                        // bytes to floats directly (v_cvt_f32_ubyteN of the field words); orig - blur is exact in fp32
                        const uint32_t brb = srb >> 4, bga = sga >> 4;      // blurred R | B << 16, blurred G in byte 0
                        const float fr = static_cast<float>(crb & 0xffu), fg = static_cast<float>(cga & 0xffu),
                                    fb = static_cast<float>((crb >> 16) & 0xffu);
                        const float dr = fr - static_cast<float>(brb & 0xffu), dg = fg - static_cast<float>(bga & 0xffu),
                                    db = fb - static_cast<float>((brb >> 16) & 0xffu);
                        const float ar = fmaf(t, dr, fr + seed);
                        const float ag = fmaf(t, dg, fg + seed);
                        const float ab = fmaf(t, db, fb + seed);
                        const float hr = ar + g2, hg = ag + g2, hb = ab + g2;
                        fp32_round_toward_zero();
                        out = pk8(ab, 2, pk8(ag, 1, pk8(ar, 0, c)));
                        const uint32_t out2 = pk8(hb, 2, pk8(hg, 1, pk8(hr, 0, c)));
                        fp32_round_nearest();
                        flagged = out != out2;
                        if (a.use_table && m2 > 1.6000016e11f) {     // e == 1 for certain (400000^2 + 1e-5 relative): exact by table
                            out = tab();
                            flagged = false;
                        }
                        if (flagged) {
                            const int e = atomicAdd(&s_nfix, 1);
                            if (e < FX_FIX_CAP) s_fix[e] = static_cast<uint16_t>(((rg * FX_RP + j) << 8) | cx);
                        }
                    }
                }
            }
            // (32-bit offsets: the store takes the scalar base + one VGPR; launch_fx sends images of 2 GiB and more elsewhere)
            if ((INTERIOR || (x < a.w && y < a.h)) && !flagged)
                *(g_u32w *)(a.dst + (doff0 + static_cast<uint32_t>(j) * static_cast<uint32_t>(a.dstride))) = out;
        }
    };
    // AdaptiveSharpen, interior tiles: two rows at a time, so that the fp32 chain runs as v_pk_* on (row j, row j + 1)
    // pairs -- the same operations in the same order per sample as the one-row form above (same guard proof), with
    // one boundary test per sample instead of a second pack: acc = V~ + 0.5 - G is within E < G of V + 0.5 - G, so
    // V + 0.5 lies in (acc, acc + 2G) and floor(V + 0.5) == floor(acc) unless an integer does too, i.e. unless
    // fract(acc) >= 1 - 2G (flag_thr is that bound rounded down); v_fract_f32 is exact.  Negative acc and acc > 255
    // only ever flag too much: both forms saturate.
    auto rows2 = [&]() {
        const v2f seed2 = {seed, seed};
        const float amt = __builtin_canonicalizef(a.amt32), thr = a.flag_thr;
        const v2f k2 = {a.k32, a.k32};
        const bool use_table = a.use_table != 0;
        // the tile rows roll with the pairs (rows j + 2, j + 3 are read while pair j is computed) and nothing is
        // scheduled across a pair's end: with all ten rows read up front the kernel spilled six registers at its
        // 96-VGPR budget, and a reload inside this loop waits for vmcnt(0) -- i.e. for the previous pair's stores
        tile_row(0);
        tile_row(1);
#pragma unroll
        for (int j = 0; j < FX_RP; j += 2) {
            __builtin_amdgcn_sched_barrier(0);
            tile_row(j + 2);
            tile_row(j + 3);
            const int ci = base + (j + 1) * LW;
            const uint32_t crb0 = s_rb[ci], cga0 = s_ga[ci], crb1 = s_rb[ci + LW], cga1 = s_ga[ci + LW];
            const uint32_t srb0 = hrb[j] + hrb[j + 2] + 2 * hrb[j + 1] + 0x00080008u;
            const uint32_t sga0 = hga[j] + hga[j + 2] + 2 * hga[j + 1] + 0x00080008u;
            const uint32_t srb1 = hrb[j + 1] + hrb[j + 3] + 2 * hrb[j + 2] + 0x00080008u;
            const uint32_t sga1 = hga[j + 1] + hga[j + 3] + 2 * hga[j + 2] + 0x00080008u;
            const uint32_t brb0 = srb0 >> 4, bga0 = sga0 >> 4, brb1 = srb1 >> 4, bga1 = sga1 >> 4;
            const v2f gx = {static_cast<float>(dxr[j] + dxr[j + 2] + 2 * dxr[j + 1]), static_cast<float>(dxr[j + 1] + dxr[j + 3] + 2 * dxr[j + 2])};
            const v2f gy = {static_cast<float>(sxr[j + 2] - sxr[j]), static_cast<float>(sxr[j + 3] - sxr[j + 1])};
            const v2f m2 = __builtin_elementwise_fma(gy, gy, gx * gx);
            const v2f sq = {__builtin_amdgcn_sqrtf(m2.x), __builtin_amdgcn_sqrtf(m2.y)};
            const v2f tk = sq * k2;
            const v2f t = {fminf(amt, tk.x), fminf(amt, tk.y)};
            auto chan = [&](float o0, float o1, float b0, float b1) {
                const v2f f = {o0, o1}, bl = {b0, b1};
                return __builtin_elementwise_fma(t, f - bl, f + seed2);
            };
            // bytes to floats straight from the field words (opaque to the compiler, which otherwise subtracts in integers
            // behind byte masks: 7 instructions per sample where this is 2 + 1.5)
            const v2f ar = chan(ubyte_f32<0>(crb0), ubyte_f32<0>(crb1), ubyte_f32<0>(brb0), ubyte_f32<0>(brb1));
            const v2f ag = chan(ubyte_f32<0>(cga0), ubyte_f32<0>(cga1), ubyte_f32<0>(bga0), ubyte_f32<0>(bga1));
            const v2f ab = chan(ubyte_f32<2>(crb0), ubyte_f32<2>(crb1), ubyte_f32<2>(brb0), ubyte_f32<2>(brb1));
            const float f0 = fmaxf(fmaxf(__builtin_amdgcn_fractf(ar.x), __builtin_amdgcn_fractf(ag.x)), __builtin_amdgcn_fractf(ab.x));
            const float f1 = fmaxf(fmaxf(__builtin_amdgcn_fractf(ar.y), __builtin_amdgcn_fractf(ag.y)), __builtin_amdgcn_fractf(ab.y));
            fp32_round_toward_zero();
            uint32_t out0 = pk8(ab.x, 2, pk8(ag.x, 1, pk8(ar.x, 0, cga0 << 8)));
            uint32_t out1 = pk8(ab.y, 2, pk8(ag.y, 1, pk8(ar.y, 0, cga1 << 8)));
            fp32_round_nearest();
            bool flag0 = f0 >= thr, flag1 = f1 >= thr;
            if (use_table) {                                     // e == 1 for certain: exact by table (see the one-row form)
                auto tab = [&](uint32_t crb, uint32_t cga, uint32_t brb, uint32_t bga) {
                    const int br = brb & 0xffu, bg = bga & 0xffu, bb = (brb >> 16) & 0xffu;
                    const int o_r = crb & 0xffu, o_g = cga & 0xffu, o_b = (crb >> 16) & 0xffu;
                    const int vr = clampi(o_r + s_tab[o_r - br + 255], 0, 255), vg = clampi(o_g + s_tab[o_g - bg + 255], 0, 255),
                              vb = clampi(o_b + s_tab[o_b - bb + 255], 0, 255);
                    return static_cast<uint32_t>(vr) | (static_cast<uint32_t>(vg) << 8) | (static_cast<uint32_t>(vb) << 16) | ((cga << 8) & 0xff000000u);
                };
                if (m2.x > 1.6000016e11f) { out0 = tab(crb0, cga0, brb0, bga0); flag0 = false; }
                if (m2.y > 1.6000016e11f) { out1 = tab(crb1, cga1, brb1, bga1); flag1 = false; }
            }
            if (flag0 || flag1) {
                if (flag0) {
                    const int e = atomicAdd(&s_nfix, 1);
                    if (e < FX_FIX_CAP) s_fix[e] = static_cast<uint16_t>(((rg * FX_RP + j) << 8) | cx);
                }
                if (flag1) {
                    const int e = atomicAdd(&s_nfix, 1);
                    if (e < FX_FIX_CAP) s_fix[e] = static_cast<uint16_t>(((rg * FX_RP + j + 1) << 8) | cx);
                }
            }
            const uint32_t d0 = doff0 + static_cast<uint32_t>(j) * static_cast<uint32_t>(a.dstride);
            if (!flag0) *(g_u32w *)(a.dst + d0) = out0;
            if (!flag1) *(g_u32w *)(a.dst + (d0 + static_cast<uint32_t>(a.dstride))) = out1;
        }
    };
    if constexpr (MODE == FX_ADAPTIVE) {
        if (interior && a.pairs) rows2();
        else if (interior) rows(std::true_type{});
        else rows(std::false_type{});
    } else {
        if (interior) rows(std::true_type{});
        else rows(std::false_type{});
    }
    if constexpr (MODE == FX_ADAPTIVE) {
        // flagged pixels (a rounding boundary within G of the fp32 value): the reference's own fp64 arithmetic.
        // A list overflow recomputes every interior pixel of the tile.
        __syncthreads();
        const int nfix = s_nfix;
        const int total = nfix > FX_FIX_CAP ? FX_TW * FX_TH : nfix;
        for (int e = tid; e < total; e += 256) {
            const int row = nfix > FX_FIX_CAP ? e >> 6 : static_cast<int>(s_fix[e] >> 8);
            const int col = nfix > FX_FIX_CAP ? e & 63 : static_cast<int>(s_fix[e] & 0xffu);
            const int px = x0 + col, py = y0 + row;
            if (!(px >= 1 && py >= 1 && px < a.w - 1 && py < a.h - 1)) continue;
            const int ci = (row + 1) * LW + 4 + col;
            uint32_t nrb[9], nga[9];
#pragma unroll
            for (int jj = 0; jj < 3; jj++)
#pragma unroll
                for (int ii = 0; ii < 3; ii++) {
                    nrb[3 * jj + ii] = s_rb[ci + (jj - 1) * LW + ii - 1];
                    nga[3 * jj + ii] = s_ga[ci + (jj - 1) * LW + ii - 1];
                }
            *(g_u32w *)(a.dst + static_cast<size_t>(py) * a.dstride + 4 * static_cast<size_t>(px)) =
                fx_exact_px<FX_ADAPTIVE>(nrb, nga, a.amount);
        }
    }
}

// one flagged interior sample of AdaptiveSharpen, recomputed from the source image in the reference's own fp64 arithmetic
__device__ __forceinline__ void fx_fix_from_source(const FxArgs &a, int px, int py)
{
    uint32_t nrb[9], nga[9];
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const uint32_t p = ld_px(a.src + static_cast<size_t>(py + j - 1) * a.sstride, px + i - 1);
            nrb[3 * j + i] = p & 0x00ff00ffu;
            nga[3 * j + i] = (p >> 8) & 0x00ff00ffu;
        }
    *(g_u32w *)(a.dst + static_cast<size_t>(py) * a.dstride + 4 * static_cast<size_t>(px)) = fx_exact_px<FX_ADAPTIVE>(nrb, nga, a.amount);
}

// ------------------------------------------------------------------------------------
// streaming form (round 3): a WAVE marches down a strip of the image -- what every tight or strided image takes
// ------------------------------------------------------------------------------------
// The tile kernel above is phases -- load tile, barrier, compute, store -- and what hides one workgroup's phase is other
// workgroups: gaussianBlur3x3, with next to no arithmetic, takes 58 us per 8K image (4.5 TB/s against a 6.2-7 TB/s copy),
// AdaptiveSharpen 74-82 us at 70-80 % VALU issue, and neither moved with fewer instructions or more workgroups per CU.
// Here nothing waits for a tile: a wave owns FXS_COLS = 62 output columns (64 pixel columns, one per lane) and walks
// down a segment of rows with FXS_PF rows of loads in flight; per row a lane turns ITS pixel into the field words and the
// integer milli-luminance once, hands them to its two neighbours through a per-wave LDS row (no barrier: one wave, LDS
// is in order), and keeps the horizontal sums of the last three rows in a register ring (unrolled over the ring's
// phases, so every index is static).  Column halo 64 / 62, row halo (S + 2) / S; the arithmetic per output is the tile
// kernel's one-row form with the boundary test of its paired form (same guard, same proof).  Flagged samples go to a
// per-wave list and are recomputed at the end of the segment -- in fp64, in the reference's order, from the source.
constexpr int FXS_COLS = 62;
constexpr int FXS_PF = 6;                  // rows in flight per lane; the unroll is lcm(PF, 2 LDS slots).  (Eight registers, so that a row's
                                           // pixel stays where it was loaded while it is the next row's centre, and an eight-row unroll:
                                           // 74 us against 65 at 8K.)
constexpr int FXS_LW = 68;
constexpr int FXS_FIX = 64;

template <int MODE>
__global__ __launch_bounds__(256, 8) void fx_stream_kernel(FxArgs a)
{
    __shared__ uint32_t s_x[4][2][3][FXS_LW];                            // [wave][slot][R|B, G|A, I][lane + 1]
    __shared__ int16_t s_tab[MODE != FX_BLUR3 ? 512 : 2];
    __shared__ uint32_t s_fix[4][MODE == FX_ADAPTIVE ? FXS_FIX : 1];
    __shared__ int s_nfix[4];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    if (MODE == FX_SHARPEN || (MODE == FX_ADAPTIVE && a.use_table)) {
        s_tab[tid] = static_cast<int16_t>(a.rtab[tid]);
        s_tab[256 + tid] = static_cast<int16_t>(a.rtab[256 + tid]);
    }
    if (tid < 4) s_nfix[tid] = 0;
    if (a.srcs) {                                                        // a batch: this workgroup's image
        a.src = a.srcs[blockIdx.y];
        a.dst = a.dsts[blockIdx.y];
    }
    __syncthreads();                                                     // the only barrier
    const int item = blockIdx.x * 4 + wave;
    if (item >= a.strips * a.segs) return;                               // wave-uniform
    const int seg = item / a.strips, strip = item - seg * a.strips;
    const int x = strip * FXS_COLS - 1 + lane;
    const uint32_t xoff = 4u * static_cast<uint32_t>(clampi(x, 0, a.w - 1));   // clamped reads: see the tile kernel
    const uint32_t xo = 4u * static_cast<uint32_t>(x);                   // (only lanes with 0 <= x < w store)
    const int y0 = seg * a.seg_rows;
    const int nout = min(a.seg_rows, a.h - y0), nrows = nout + 2;        // rows y0 - 1 .. y0 + nout
    const bool mine = lane >= 1 && lane <= FXS_COLS && x < a.w;          // lanes 0 and 63 are taps only
    const bool xborder = x <= 0 || x >= a.w - 1;
    auto load_row = [&](int r) {                                         // r is wave-uniform: a scalar row base + one VGPR
        const int yy = clampi(y0 - 1 + min(r, nrows - 1), 0, a.h - 1);
        return *(g_u32 *)(a.src + (static_cast<uint32_t>(yy) * static_cast<uint32_t>(a.sstride) + xoff));
    };
    uint32_t q[FXS_PF];
#pragma unroll
    for (int k = 0; k < FXS_PF; k++) q[k] = load_row(k);
    // Vertical [1 2 1] over the horizontal sums h of rows T, M, B (B = the newest) as two PAIR sums, (h[T] + h[M]) + (h[M] + h[B]):
    // the older pair is last row's newer one, so a row costs two adds per word where the three-entry ring took a shift, an add
    // and an add3; the +8 of the rounding rides in as +2 per h.  Same for the Sobel column differences; the row sums need the
    // entry two rows back only.  The centre pixel (row M) is last row's source word as it was loaded: its bytes are the
    // channels' floats (v_cvt_f32_ubyteN), its alpha the byte the packs leave alone, and the whole word the copy a border gets.
    uint32_t hl_rb = 0u, hl_ga = 0u, pl_rb = 0u, pl_ga = 0u, px_m = 0u;   // h[M], h[T] + h[M], the source pixel of row M
    int32_t dl = 0, pdl = 0, sl1 = 0, sl2 = 0;                              // dx[M], dx[T] + dx[M], sx[M], sx[T]
    const float seed = 0.5f - a.guard;
    const float amt = __builtin_canonicalizef(a.amt32), k32 = a.k32, thr = a.flag_thr;
    const bool use_table = a.use_table != 0;
    // per-lane forms of the amount and of the saturation bound: the image's border columns are copies of the source
    const float amt_lane = xborder ? 0.0f : amt, sat_lane = xborder ? __builtin_inff() : 1.6000016e11f;
    uint32_t(*sw)[3][FXS_LW] = s_x[wave];

    for (int r0 = 0; r0 < nrows; r0 += 6) {
#pragma unroll
        for (int p = 0; p < 6; p++) {
            const int r = r0 + p;
            if (r < nrows) {                                             // wave-uniform
                const int slot = p & 1;
                const uint32_t px = q[p % FXS_PF];
                q[p % FXS_PF] = load_row(r + FXS_PF);                    // no branch around the load
                const uint32_t rb = px & 0x00ff00ffu, ga = ga_fields(px);
                sw[slot][0][lane + 1] = rb;
                sw[slot][1][lane + 1] = ga;
                uint32_t lum = 0;
                if constexpr (MODE == FX_ADAPTIVE) {
                    lum = lum_milli_u32(px);                    // (two v_dot2_u32_u16 on the spread fields measured SLOWER: 69 against 65 us at 8K)
                    sw[slot][2][lane + 1] = lum;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t h_rb = sw[slot][0][lane] + sw[slot][0][lane + 2] + (2 * rb + 0x00020002u);
                const uint32_t h_ga = sw[slot][1][lane] + sw[slot][1][lane + 2] + (2 * ga + 0x00020002u);
                const uint32_t p_rb = hl_rb + h_rb, p_ga = hl_ga + h_ga;
                const uint32_t srb = pl_rb + p_rb, sga = pl_ga + p_ga;   // 16 x the blur + 8 per field (<= 4088)
                int32_t d_new = 0, pd = 0, s_new = 0;
                if constexpr (MODE == FX_ADAPTIVE) {
                    const int32_t l = static_cast<int32_t>(sw[slot][2][lane]), rr = static_cast<int32_t>(sw[slot][2][lane + 2]);
                    d_new = rr - l;
                    pd = dl + d_new;
                    s_new = l + rr + 2 * static_cast<int32_t>(lum);
                }
                if (r >= 2) {
                    // output row yo = the row before the newest: T (yo - 1), M (yo), B (yo + 1)
                    const int yo = y0 + r - 2;
                    const uint32_t c = px_m;
                    uint32_t out = c;                                    // borders and alpha are copies of the source (effects.go:68,120)
                    bool flagged = false;
                    if (yo >= 1 && yo < a.h - 1) {                       // wave-uniform
                        const uint32_t brb = srb >> 4, bga = sga >> 4;   // blurred R | B << 16 (bytes 0, 2), blurred G in byte 0
                        auto tab = [&]() {
                            const int br = brb & 0xffu, bg = bga & 0xffu, bb = (brb >> 16) & 0xffu;
                            const int o_r = c & 0xffu, o_g = (c >> 8) & 0xffu, o_b = (c >> 16) & 0xffu;
                            const int vr = clampi(o_r + s_tab[o_r - br + 255], 0, 255), vg = clampi(o_g + s_tab[o_g - bg + 255], 0, 255),
                                      vb = clampi(o_b + s_tab[o_b - bb + 255], 0, 255);
                            return static_cast<uint32_t>(vr) | (static_cast<uint32_t>(vg) << 8) | (static_cast<uint32_t>(vb) << 16) | (c & 0xff000000u);
                        };
                        uint32_t v;
                        if constexpr (MODE == FX_BLUR3) {
                            v = __builtin_amdgcn_perm(bga, brb, 0x0c020400u) | (c & 0xff000000u);   // R, G, B bytes of the sums
                        } else if constexpr (MODE == FX_SHARPEN) {
                            v = tab();
                        } else {
                            const float gx = static_cast<float>(pdl + pd);   // dx[T] + 2 dx[M] + dx[B]
                            const float gy = static_cast<float>(s_new - sl2);
                            const float m2 = fmaf(gy, gy, gx * gx);
                            float t;                                     // min(amount, amount * |Sobel| / 400000); 0 on the image's border columns
                            asm("v_min_f32 %0, %1, %2" : "=v"(t) : "v"(__builtin_amdgcn_sqrtf(m2) * k32), "v"(amt_lane));
                            const float fr = ubyte_f32<0>(c), fg = ubyte_f32<1>(c), fb = ubyte_f32<2>(c);
                            const float ar = fmaf(t, fr - ubyte_f32<0>(brb), fr + seed);
                            const float ag = fmaf(t, fg - ubyte_f32<0>(bga), fg + seed);
                            const float ab = fmaf(t, fb - ubyte_f32<2>(brb), fb + seed);
                            const float fmx = fmaxf(fmaxf(__builtin_amdgcn_fractf(ar), __builtin_amdgcn_fractf(ag)), __builtin_amdgcn_fractf(ab));
                            fp32_round_toward_zero();
                            v = pk8(ab, 2, pk8(ag, 1, pk8(ar, 0, c)));       // (byte 3 stays: the source's alpha)
                            fp32_round_nearest();
                            flagged = fmx >= thr;
                            if (use_table && m2 > sat_lane) {            // e == 1 for certain: exact by table
                                v = tab();
                                flagged = false;
                            }
                        }
                        // (AdaptiveSharpen: a border column runs with amount 0 -- acc = orig + 0.5 - G packs to orig and never flags)
                        out = (MODE != FX_ADAPTIVE && xborder) ? c : v;
                        flagged = flagged && mine;
                        if constexpr (MODE == FX_ADAPTIVE) {
                            if (flagged) {
                                const int e = atomicAdd(&s_nfix[wave], 1);
                                if (e < FXS_FIX) s_fix[wave][e] = (static_cast<uint32_t>(yo) << 16) | static_cast<uint32_t>(x);
                            }
                        }
                    }
                    if (mine && !flagged) *(g_u32w *)(a.dst + (static_cast<uint32_t>(yo) * static_cast<uint32_t>(a.dstride) + xo)) = out;
                }
                hl_rb = h_rb; hl_ga = h_ga; pl_rb = p_rb; pl_ga = p_ga; px_m = px;
                if constexpr (MODE == FX_ADAPTIVE) { dl = d_new; pdl = pd; sl2 = sl1; sl1 = s_new; }
            }
        }
    }
    if constexpr (MODE == FX_ADAPTIVE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // a list that overflowed (an amount / image that flags nearly everything): every interior sample of the segment
        const int nfix = s_nfix[wave];
        const bool over = nfix > FXS_FIX;
        const int total = over ? nout * FXS_COLS : nfix;
        for (int e = lane; e < total; e += 64) {
            int fx, fy;
            if (over) {
                fy = y0 + e / FXS_COLS;
                fx = strip * FXS_COLS + (e - (e / FXS_COLS) * FXS_COLS);
            } else {
                const uint32_t ent = s_fix[wave][e];
                fx = static_cast<int>(ent & 0xffffu);
                fy = static_cast<int>(ent >> 16);
            }
            if (fx >= 1 && fy >= 1 && fx < a.w - 1 && fy < a.h - 1) fx_fix_from_source(a, fx, fy);
        }
    }
}

// R[d + 255] = floor(fl(amount * d) + 0.5), d in [-255, 255].  false: some product is within 1e-6 of a
// half-integer without being one (the sum's fp64 rounding could then decide), or the table would not fit.
// *ties: some product IS a half-integer (only the guard's statistics care).
static bool build_rtab(double amount, int32_t (&tab)[512], bool *ties)
{
    *ties = false;
    if (!(std::fabs(amount) <= 64.0)) return false;
    for (int d = -255; d <= 255; d++) {
        const double p = amount * static_cast<double>(d);        // the reference's amount * (orig - blur)
        const double fl = std::floor(p), f = p - fl;
        if (f == 0.5) *ties = true;
        else if (std::fabs(f - 0.5) < 1e-6) return false;
        tab[d + 255] = static_cast<int32_t>(fl) + (f >= 0.5 ? 1 : 0);
    }
    tab[511] = 0;
    return true;
}

// AdaptiveSharpen's rounding guard.  With I = 1000 * luminance exact, the real value of the reference's
// expression is V = orig + A e d, e = min(1, sqrt(gx^2 + gy^2) / 400000), |d| <= 255, A = amount in (0, 8].
// fp32 chain: gx, gy exact; m2 = fma(gy, gy, gx gx): <= 2^-23 relative; v_sqrt_f32 (1 ulp) of it: <= 1.5 * 2^-23;
// times k32 = fl32(A / 400000), one multiply: <= 2.5 * 2^-23 < 3e-7; min with fl32(A): same bound.  So
// t = A e (1 + dt), |dt| <= 4e-7 (margin included), |t d - A e d| <= A * 255 * 4e-7.  orig + seed rounds by
// <= 2^-17 (< 256); the final fma rounds by half an ulp of a value below 256 (1 + A) + 1.  The reference's fp64
// chain is within 1e-9 of V (luminances to 3 ulp, IEEE sqrt and divide).  The second pack's addend rounds by
// one more half ulp (eta).  G = E + eta + 1e-5 > E + eta + 1e-9.
static bool fx_guard(double amount, float *guard)
{
    if (!(amount > 0.0 && amount <= 8.0)) return false;
    const double top = 256.0 * (1.0 + amount) + 1.0;
    const double half_ulp = std::ldexp(1.0, static_cast<int>(std::ceil(std::log2(top))) - 24);
    const double E = amount * 255.0 * 4e-7 + std::ldexp(1.0, -17) + half_ulp;
    const double G = E + half_ulp + 1e-5;
    float g = static_cast<float>(G);
    if (static_cast<double>(g) < G) g = std::nextafterf(g, 1.0f);
    *guard = g;
    return true;
}

// ------------------------------------------------------------------------------------
// SubImage inputs (sstride != 4w): the reference's FLAT copies
// ------------------------------------------------------------------------------------
// gaussianBlur3x3 and AdaptiveSharpen start from copy(dst.Pix, img.Pix) (effects.go:120,68): Go's copy() moves the
// first min(len(dst.Pix), len(img.Pix)) = 4wh FLAT bytes of the source slice into the tight destination, which are the
// image's rows only when Stride == 4w.  So on a SubImage the reference's 3x3 blur has the flat bytes on its border and
// in EVERY alpha byte (the interior writes R, G, B only, effects.go:135), AdaptiveSharpen has them on its border
// (its interior writes all four bytes, :83-85), and Sharpen -- whose own output is written pixel by pixel (:28-42) --
// sees them through the border of its blurred operand: val = orig + amount * (orig - flat).  The tile kernels above
// compute every strided read of the reference; this pass, launched behind them for sstride != 4w only, rewrites what
// the flat copy decides.  Pixel (x, y) of the tight destination is flat bytes [4(wy + x), +4) of the source slice.
template <int MODE>
__global__ __launch_bounds__(256) void fx_flat_kernel(FxArgs a)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.w || y >= a.h) return;
    const bool border = x == 0 || y == 0 || x == a.w - 1 || y == a.h - 1;
    if (MODE != FX_BLUR3 && !border) return;
    const uint32_t flat = ld_px(a.src + (static_cast<size_t>(y) * a.w) * 4, x);
    g_u32w *dp = (g_u32w *)(a.dst + static_cast<size_t>(y) * a.dstride + 4 * static_cast<size_t>(x));
    if (MODE == FX_BLUR3) {
        *dp = border ? flat : ((*dp & 0x00ffffffu) | (flat & 0xff000000u));
    } else if (MODE == FX_ADAPTIVE) {
        *dp = flat;
    } else {
        const uint32_t c = ld_px(a.src + static_cast<size_t>(y) * a.sstride, x);
        uint32_t out = c & 0xff000000u;                               // effects.go:40
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const double orig = u8_to_f64((c >> (8 * ch)) & 0xffu);
            const double bl = u8_to_f64((flat >> (8 * ch)) & 0xffu);
            const double val = orig + a.amount * (orig - bl);         // effects.go:37
            out |= clampF_dev(val) << (8 * ch);
        }
        *dp = out;
    }
}

// nimg > 1: a batch of same-geometry TIGHT images in one launch of the streaming kernel (d_srcs / d_dsts: device arrays of their
// pointers); FNX_NOOP -- nothing launched -- where that kernel does not apply (the caller then goes image by image)
template <int MODE>
static int launch_fx(fnx_ctx *ctx, const uint8_t *src, int sstride, int w, int h, double amount,
                     uint8_t *dst, int dstride, int nimg = 1, const uint8_t *const *d_srcs = nullptr, uint8_t *const *d_dsts = nullptr)
{
    if (w <= 0 || h <= 0) return FNX_OK;
    const unsigned nz = static_cast<unsigned>(nimg > 1 ? nimg : 1);
    FxArgs a{};
    if (nz > 1) { a.srcs = d_srcs; a.dsts = d_dsts; }
    a.src = src; a.dst = dst; a.sstride = sstride; a.dstride = dstride; a.w = w; a.h = h; a.amount = amount;
    a.vec_ok = aligned16(src, sstride) ? 1 : 0;
    bool march = true;
    const char *force_ref = form_value(ctx, FORM_FX_REF);                // A/B and tests: "1" takes the fp64 reference-order kernel
    if (MODE != FX_BLUR3) {
        int32_t tab[512];
        bool ties = false;
        const bool tab_ok = build_rtab(amount, tab, &ties);
        if (MODE == FX_SHARPEN) {
            march = tab_ok;
        } else {
            march = fx_guard(amount, &a.guard);
            {
                // 1 - 2G, rounded down, with a margin far above the rounding of `seed` and of this subtraction
                const double thr = 1.0 - 2.0 * static_cast<double>(a.guard) - 1e-6;
                float t32 = static_cast<float>(thr);
                if (static_cast<double>(t32) > thr) t32 = std::nextafterf(t32, 0.0f);
                a.flag_thr = t32;
                const char *pe = form_value(ctx, FORM_FX_PAIRS);          // tile kernel, A/B and tests: "0" takes the one-row form
                a.pairs = pe ? atoi(pe) : 1;
            }
            a.amt32 = static_cast<float>(amount);
            a.k32 = static_cast<float>(amount / 400000.0);
            a.use_table = (tab_ok && ties) ? 1 : 0;     // tie-prone amounts (1.5, 2.5, ...): saturated pixels skip the guard
        }
        if (march && (MODE == FX_SHARPEN || a.use_table)) {
            void *d = nullptr;
            FNX_TRY(upload_table(ctx, SLOT_TABLE0, tab, sizeof(tab), &d));
            a.rtab = static_cast<const int32_t *>(d);
        }
    }
    if (force_ref && force_ref[0] == '1') march = false;
    // the tile kernels address the destination with 32-bit offsets
    if (static_cast<long long>(h) * dstride >= (1ll << 31)) march = false;
    const char *stream_env = form_value(ctx, FORM_FX_STREAM);            // A/B and tests: "0" takes the tile kernel
    const int stream_on = stream_env ? atoi(stream_env) : 1;
    const bool streams = march && stream_on && w < 65536 && h < 65536 && static_cast<long long>(h) * sstride < (1ll << 31);
    if (nz > 1 && (!streams || sstride != w * 4)) return FNX_NOOP;       // (a SubImage's flat-copy pass is per image)
    FNX_TRY(prof_begin(ctx, FNX_PROF_FX));
    if (streams) {
        // one wave per (strip, segment); segments sized so that the launch is one round of resident waves
        static const int per_cu = [] {
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fx_stream_kernel<MODE>, 256, 0) != hipSuccess || nb < 1) nb = 4;
            return nb;
        }();
        static const int rounds = [] { const char *e = dev_env("FNX_FX_ROUNDS"); return e ? atoi(e) : 1; }();
        a.strips = (w + FXS_COLS - 1) / FXS_COLS;
        const long capacity = static_cast<long>(ctx->num_cus) * per_cu * 4 * rounds;
        int segs = static_cast<int>(capacity / (static_cast<long>(a.strips) * nz));
        segs = segs < 1 ? 1 : segs;
        a.seg_rows = (h + segs - 1) / segs;
        if (a.seg_rows < 16) a.seg_rows = h < 16 ? h : 16;
        a.segs = (h + a.seg_rows - 1) / a.seg_rows;
        const int items = a.strips * a.segs;
        hipLaunchKernelGGL((fx_stream_kernel<MODE>), dim3((items + 3) / 4, nz), dim3(256), 0, ctx->stream, a);
    } else if (march) {
        dim3 grid((w + FX_TW - 1) / FX_TW, (h + FX_TH - 1) / FX_TH);
        // FNX_FX_LDS_PAD=<bytes> of unused dynamic LDS: an A/B knob for workgroups per CU (8192 -> 4 instead of 5)
        static const unsigned pad = [] { const char *e = dev_env("FNX_FX_LDS_PAD"); return e ? static_cast<unsigned>(atoi(e)) : 0u; }();
        hipLaunchKernelGGL((fx_march_kernel<MODE>), grid, dim3(256), pad, ctx->stream, a);
    } else {
        dim3 grid((w + FXR_TW - 1) / FXR_TW, (h + FXR_TH - 1) / FXR_TH);
        hipLaunchKernelGGL((fx_ref_kernel<MODE>), grid, dim3(256), 0, ctx->stream, a);
    }
    if (sstride != w * 4) {                                      // a SubImage: the reference's flat copies (see fx_flat_kernel)
        dim3 grid((w + 63) / 64, (h + 3) / 4);
        hipLaunchKernelGGL((fx_flat_kernel<MODE>), grid, dim3(256), 0, ctx->stream, a);
    }
    FNX_HIP(hipGetLastError());
    FNX_TRY(prof_end(ctx));
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

// n same-geometry tight device images in one launch (d_srcs / d_dsts: device pointer arrays; src0 / dst0: the first pair, for
// the alignment the kernel form depends on); FNX_NOOP where the streaming kernel does not apply
int launch_sharpen_batch(fnx_ctx *ctx, bool adaptive, int n, const uint8_t *src0, const uint8_t *const *d_srcs, int sstride, int w, int h,
                         double amount, uint8_t *dst0, uint8_t *const *d_dsts, int dstride)
{
    return adaptive ? launch_fx<FX_ADAPTIVE>(ctx, src0, sstride, w, h, amount, dst0, dstride, n, d_srcs, d_dsts)
                    : launch_fx<FX_SHARPEN>(ctx, src0, sstride, w, h, amount, dst0, dstride, n, d_srcs, d_dsts);
}

}  // namespace fnx
