// GaussianBlur (effects.go:146-220), radius <= 6 (sigma <= 2), with both separable passes on the gfx950 i8 matrix pipe --
// and, with SCORE, SSIMFast's boxDownsample sums (ssim.go:244-309) of the source and of the blurred image in the same pass.
//
// Why the matrix pipe for a stencil.  The direct form (blur.hip: blur_direct_kernel) spends 83 VALU instructions per pixel
// -- 39 packed fp32 FMAs, 14 byte->float converts, 6 packs, 11 for the box sums -- and is VALU-issue bound at 0.36 of the
// HBM peak (profiles/r03_onepass_sq_counters.txt).  Here the 13-tap sums are INTEGER dot products on
// v_mfma_i32_16x16x64_i8 / 16x16x32_i8: the bytes go in as they lie in memory (no converts), the weights are 24-bit fixed
// point split into three signed base-256 digits (three matrix instructions, two v_lshl_add_u32 per sample to add the
// digits' sums), and the rounding is "+2^23, take the top byte".  The integer sum S = sum wq[k] p[k] is EXACT; with
// sum wq = 2^24 it never leaves [0, 255 * 2^24], so there is no clamp and no overflow.
//
//  * H pass, one set = 16 rows x 4 output px: A = source bytes (M = row, K = the 64 bytes of the 16-px window), B = the
//    weights as a Toeplitz matrix over the RGBA-interleaved bytes (K x N = 16 output bytes: 4 px x RGBA; 13 of 64 K
//    entries are non-zero per column).  C comes out with 4 CONSECUTIVE ROWS of one byte column per lane: packed, that is
//    one dword of the TRANSPOSED uint8 intermediate (effects.go:186-188) T[byte column][row].
//  * V pass, one set = 16 output rows x 16 byte columns: A = T (M = byte column, K = 32 staged rows, 8 contiguous bytes per
//    lane), B = the weights (K x N = 16 output rows).  C is row-major again: lane (row, chunk) ends up with 4 px = 16 bytes.
//  * alpha: the H sets' alpha columns carry the centre pixel's alpha through (weight digit 1, byte 0 of the sum), the V
//    sets fetch it from the intermediate's alpha columns (effects.go:215: alpha comes from the ORIGINAL).
//  * SCORE: lane (row, chunk) layouts ARE matrix A operands, so the box sums are two more matrix instructions per 16 x 16
//    px block (B = 0/1 indicator of up to 5 box columns x RGB over the 64 bytes) and 4 + 4 LDS atomics on what is left.
//  * GUARD (FNX_BLUR_EXACT): S / 2^24 - exact sum lies in [-255 N, 255 P], N / P the sums of the negative / positive
//    differences wq[k] / 2^24 - w[k] (computed per call).  A sample whose fraction lies within G >= max of the two of the
//    rounding boundary is recomputed in fp64 in the reference's tap order (effects.go:169-217) and patched in place -- one
//    sample at a time by the whole wave, weights through the scalar cache (mf_exact_u).  Everything else is proven equal
//    to the reference's clampF, not sampled.
//
// Memory shape.  A workgroup (4 waves) owns a 64-px column strip of SEG rows and marches down it 16 rows per step.
// Loads and stores are workgroup-wide through LDS stages -- 304 / 256 contiguous bytes per row and wave instruction: with
// each wave fetching its own 16-px strip (64 bytes per row) the same kernel ran at 3 TB/s, the tiled-copy floor of that
// shape.  Each wave filters its own 16 px: H set s (16 staged rows) into a two-slot ring of T (32 rows, 3.3 KB per wave),
// then V set s-1; one barrier per step; stores trail by two more steps.  The H halo is 22 rows per SEG, not 12 per 104.
#include <hip/hip_ext.h>

#include "common.hpp"
#include "devutil.hpp"

#include <algorithm>
#include <cmath>
#include <type_traits>
#include <vector>

namespace fnx {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int MF_RMAX = 6;           // 4 output px + 2 R <= 16 px = the 64-byte K of one H instruction
constexpr int MF_P = 48;             // bytes per byte column of the ring: 32 rows + 16 (= 16 mod 32: conflict-free 4-row dword writes)
constexpr int MF_WT = 64 * MF_P + 256;   // ring bytes per wave: 64 byte columns + the 64-byte skew of each 16-column group
constexpr int MF_SP = 352;           // pitch of a staged source row (19 chunks of 16 bytes): conflict-free as the A operand (tools/lds_conflicts.py)
constexpr int MF_OP = 272;           // pitch of an output row in its stage
constexpr int MF_SEG_SCORE = 272;    // most rows per workgroup with SCORE (row tables and box tables in LDS)
constexpr int MF_SEG_SCORE_WIDE = 544;   // ... of blur_mfma_wide_kernel<2, ., SCORE>: two workgroups per CU either way (its ring and stages are 43 KB)

struct MfmaArgs {
    const uint8_t *src;
    const uint8_t *const *srcs;
    uint8_t *dst;
    uint8_t *const *dsts;
    int sstride, dstride, w, h;
    int tiles_x, tiles, seg;          // per image; seg = output rows per workgroup (multiple of 16)
    int radius;
    const uint32_t *tab;              // BH[3][64] x 16 B | BV[3][64] x 8 B | the caller's 13 fp64 weights, centred (GUARD)
    int seed_h, seed_v, thr;          // rounding seeds (+ G in units of 2^-24 with GUARD), 2 G
    // SCORE
    const int32_t *bx, *by;           // box column / row of each source column / row (-1: none)
    unsigned long long *slabs;        // [image][tile][2][slabn] packed 4 x u16 channel sums (blur.hip: box_from_slabs_kernel)
    int nbx, nby;
    int dbg;                          // development switches (FNX_MFMA_DBG)
};

// (hi * 256 + mid) * 256 + lo as two v_lshl_add_u32 (left alone the compiler builds two shifts and a v_add3)
__device__ __forceinline__ int mf_comb3(int hi, int mid, int lo)
{
    int t = hi * 256 + mid;
    asm volatile("" : "+v"(t));
    return t * 256 + lo;
}

// One flagged sample again, in the reference's own arithmetic (effects.go:169-217): acc = acc + float64(p) * w, taps ascending,
// clampF.  Always 13 taps: the table holds the caller's weights centred in the radius-6 frame with zeros around them, and
// adding p * 0.0 = +0.0 to the non-negative running sum leaves it as it is.  H: the staged source bytes (p ^ 0x80) at
// base[4 t]; V: the ring's bytes at base[(ring0 + t) & 31].
// UNIFORM: the whole wave recomputes ONE flagged sample -- the 13 weights (padded to 16 with zeros) arrive through the scalar
// cache (two s_load, one wait), the bytes are LDS broadcasts from base[((ring0 + t) & mask) * step].  No call, no per-lane
// weight loads: the out-of-line per-lane version this replaces cost 20 VGPRs and ~1.5 us of a wave per sample (the other three
// waves of the workgroup standing at the barrier meanwhile); resize_mfma.hip's fix-ups work the same way.
// a wave-uniform pointer the compiler may hold in VGPRs, moved to SGPRs for the s_load's of the fix-ups
__device__ __forceinline__ const double *mf_scalar_ptr(const double *p)
{
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v)), hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
    return reinterpret_cast<const double *>((static_cast<unsigned long long>(hi) << 32) | lo);
}
typedef int mf_s16i __attribute__((ext_vector_type(16)));
__device__ __forceinline__ uint32_t mf_exact_u(const double *wd, const uint8_t *base, int step, int ring0, int mask)
{
    mf_s16i w0, w1;
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(w0), "=&s"(w1) : "s"(mf_scalar_ptr(wd)) : "memory");
    uint32_t v[13];
#pragma unroll
    for (int t = 0; t < 13; t++) v[t] = base[((ring0 + t) & mask) * step];
    double acc = 0;
#pragma unroll
    for (int t = 0; t < 13; t++) {
        const int lo = t < 8 ? w0[2 * (t & 7)] : w1[2 * (t & 7)], hi = t < 8 ? w0[2 * (t & 7) + 1] : w1[2 * (t & 7) + 1];
        acc = acc + u8_to_f64(v[t] ^ 0x80u) * __hiloint2double(hi, lo);
    }
    return clampF_dev(acc);
}

// SCORE: a lane's four rows (4 g .. 4 g + 3) of one box column and channel into the box table.  (Measured and dropped, r5: adding
// the rows of one box row up in the lane first -- one or two atomics instead of four for ~10 more vector instructions: the fast form
// unchanged at 21.5 us per image, the exact form 2 % slower; CHANGELOG r5.)
__device__ __forceinline__ void mf_box_add4(uint32_t *tbl, const u32x4 ro, uint32_t coln, const v4i cb)
{
#pragma unroll
    for (int k = 0; k < 4; k++)
        __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(tbl) + ro[k] + coln), static_cast<uint32_t>(cb[k]),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <bool SCORE, bool GUARD>
__global__ __launch_bounds__(256, GUARD ? 3 : 1) void blur_mfma_kernel(MfmaArgs a)
{
    constexpr int P = MF_P, WT = MF_WT, SP = MF_SP, OP = MF_OP;
    __shared__ __attribute__((aligned(16))) uint8_t s_t[4 * WT];
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[2 * 16 * SP];
    __shared__ __attribute__((aligned(16))) uint8_t s_out[2 * 16 * OP];
    // SCORE: byte offset of a staged row's / an output row's box row in the tables (spare row: outside the segment or the image)
    __shared__ __attribute__((aligned(16))) uint32_t s_rowh[SCORE ? MF_SEG_SCORE + 32 : 4];
    __shared__ __attribute__((aligned(16))) uint32_t s_rowv[SCORE ? MF_SEG_SCORE + 16 : 4];
    __shared__ uint32_t s_colbox[SCORE ? 64 : 1];         // box column (relative to the tile's first) of each tile column, 255: none
    extern __shared__ __attribute__((aligned(16))) uint32_t s_box[];   // SCORE: [source | blurred][(nby+1)(nbx+1)][R, G, B, -]

    const int tile = xcd_tile(blockIdx.x, a.tiles);
    if (tile < 0) return;
    const int z = blockIdx.y;
    const uint8_t *src = a.srcs ? a.srcs[z] : a.src;
    uint8_t *dst = a.dsts ? a.dsts[z] : a.dst;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int x0 = tx * 64, y0 = ty * a.seg;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, g = lane >> 4;
    const int NJ = min(a.seg, ((a.h - y0 + 15) >> 4) << 4) >> 4;   // V sets of this segment (the image's last one may be short)
    const int NI = NJ + 1;                                          // H sets: staged rows y0 - 6 .. y0 + 16 NJ + 9
    const bool xedge = x0 - 6 < 0 || x0 + 70 > a.w;                 // the strip's source window leaves the image: clamped px loads, masked stores

    const v4i *tbh = reinterpret_cast<const v4i *>(a.tab);
    const long *tbv = reinterpret_cast<const long *>(a.tab + 3 * 64 * 4);
    const v4i bh2 = tbh[lane], bh1 = tbh[64 + lane], bh0 = tbh[128 + lane];
    const long bv2 = tbv[lane], bv1 = tbv[64 + lane], bv0 = tbv[128 + lane];
    const bool alane = (r & 3) == 3;                                // H sets: this lane's output byte column is an alpha column
    const int seed_hl = alane ? (1 << 23) + 128 : a.seed_h;         // alpha lanes: (a - 128) + 128 in byte 0, no guard offset
    const v4i sh = {seed_hl, seed_hl, seed_hl, seed_hl}, sv = {a.seed_v, a.seed_v, a.seed_v, a.seed_v};
    const v4i zero = {0, 0, 0, 0};
    const uint32_t sel01 = alane ? 0x0c0c0400u : 0x0c0c0703u, sel23 = alane ? 0x04000c0cu : 0x07030c0cu;

    uint8_t *tw = s_t + wave * WT;
    // stage: chunk ids 0..303 = 16 rows x 19 chunks of 16 bytes (px x0 - 6 .. x0 + 69); thread tid takes id tid and, tid < 48, id 256 + tid
    const int id1 = min(256 + tid, 16 * 19 - 1);                    // (lanes 48.. fetch the last chunk again and drop it: see hload)
    const int srow0 = tid / 19, sch0 = tid - 19 * srow0, srow1 = id1 / 19, sch1 = id1 - 19 * srow1;
    const bool two = tid < 48;
    const int st_w0 = srow0 * SP + 16 * sch0, st_w1 = srow1 * SP + 16 * sch1;
    const int st_r = r * SP + 64 * wave + 16 * g;                   // A operand of H set qq: + 16 qq
    uint8_t *t_w = tw + r * P + 4 * g;                              // + (16 qq) P + 64 qq + 16 slot
    const int m4 = r >> 2, mi = r & 3;
    // A operand of V column set q (+ 4 q P): lane chunk g holds staged rows 8 g .. 8 g + 7 of the set's 32; an odd set's first
    // 16 staged rows sit in ring rows 16..31, its last 16 in rows 0..15 -- a second lane address, the same weights
    const uint8_t *t_r = tw + (16 * m4 + mi) * P + 64 * m4 + 8 * g;
    const uint8_t *t_ro = tw + (16 * m4 + mi) * P + 64 * m4 + ((8 * g + 16) & 31);
    const uint8_t *t_ae = tw + (16 * g + 3) * P + 64 * g + r + 6;   // centre row's alpha, even / odd V sets: + 4 q P
    const uint8_t *t_ao = tw + (16 * g + 3) * P + 64 * g + ((r + 22) & 31);
    const int o_w = r * OP + 64 * wave + 16 * g;
    const int orow = tid >> 4, och = tid & 15;
    const int o_r = orow * OP + 16 * och;
    const int xo = x0 + 4 * och;

    // ---- SCORE set-up: row / column -> table offsets, this wave's indicator matrix ----
    v4i bbox = {0, 0, 0, 0}, sbox = {0, 0, 0, 0};
    uint32_t coln = 0;                                              // byte offset of this lane's (box column, channel) in a table row
    const int slabn = SCORE ? (a.nbx + 1) * (a.nby + 1) : 0;
    uint32_t *tbl_s = s_box, *tbl_b = s_box + 4 * slabn;
    // (called from the march once the first two row sets' loads are in flight: its table look-ups then wait beside them)
    auto score_setup = [&]() {
        if (!(a.dbg & 4)) for (int e = tid; e < 8 * slabn; e += 256) s_box[e] = 0;
        const int rowbytes = 16 * (a.nbx + 1);
        const int b0y = a.by[y0];
        for (int u = tid; u < 16 * NI; u += 256) {                  // staged row u = tile row u - 6
            const int t = u - 6;
            const int v = (t >= 0 && t < a.seg && y0 + t < a.h) ? a.by[y0 + t] : -1;
            s_rowh[u] = rowbytes * ((v >= 0 && b0y >= 0) ? v - b0y : a.nby);
        }
        for (int t = tid; t < 16 * NJ; t += 256) {
            const int v = (t < a.seg && y0 + t < a.h) ? a.by[y0 + t] : -1;
            s_rowv[t] = rowbytes * ((v >= 0 && b0y >= 0) ? v - b0y : a.nby);
        }
        if (tid < 64) {
            const int b0x = a.bx[x0], v = x0 + tid < a.w ? a.bx[x0 + tid] : -1;
            s_colbox[tid] = (v >= 0 && b0x >= 0) ? v - b0x : 255u;
        }
        __syncthreads();
        // the wave's 16 px: first box column present -> slot 0; lane n = 3 slot + channel (n = 15: nothing)
        uint32_t first = 255u;
        for (int i = 0; i < 16; i++) first = min(first, s_colbox[16 * wave + i]);
        const int slot = r / 3, ch = r - 3 * slot;
        int cnt = 0;
        for (int i = 0; i < 16; i++) cnt += (r < 15 && s_colbox[16 * wave + i] == first + slot) ? 1 : 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {                               // K = byte 16 g + 4 e + ch' of the row's 64 bytes
            const bool in = r < 15 && first != 255u && s_colbox[16 * wave + 4 * g + e] == first + slot;
            bbox[e] = in ? (1 << (8 * ch)) : 0;
        }
        const int seed = 128 * cnt;                                 // the A operands are (p - 128): the sums come out as sums of p
        sbox = (v4i){seed, seed, seed, seed};
        const uint32_t bc = (r < 15 && first != 255u && cnt > 0) ? first + slot : static_cast<uint32_t>(a.nbx);
        coln = 16u * bc + 4u * (r < 15 ? ch : 3);
    };

    // ---- exact recomputation of flagged samples (GUARD), the reference's own arithmetic (effects.go:169-217) ----
    const double *wd = reinterpret_cast<const double *>(a.tab + 3 * 64 * 4 + 3 * 64 * 2);   // 13 fp64 weights, centred

    auto stage_write = [&](const u32x4 (&d)[2], int buf) {
        uint8_t *sb = s_stage + buf * 16 * SP;
        *reinterpret_cast<u32x4 *>(sb + st_w0) = d[0] ^ 0x80808080u;
        if (two) *reinterpret_cast<u32x4 *>(sb + st_w1) = d[1] ^ 0x80808080u;
    };
    // H set s: 16 staged rows from stage `buf` into ring slot `slot`
    auto hset = [&](int s, int buf, int slot) {
        const uint8_t *sbuf = s_stage + buf * 16 * SP;
        const uint8_t *sb = sbuf + st_r;
        if constexpr (!GUARD && !SCORE) {
            // plain fast blur: two halves of two sets each -- 24 accumulator registers live instead of 48, 113 VGPRs in all:
            // FOUR workgroups per CU (4K: 15.8 -> 14.75 us per image, same box).  The one-pass form lost 3 % with it (129
            // registers, still three workgroups, less room for the scheduler), the exact form 1 % (143): both keep the
            // whole-group form.
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {
                v4i c2[2], c1[2], c0[2];
#pragma unroll
                for (int q2 = 0; q2 < 2; q2++) {
                    const v4i A = *reinterpret_cast<const v4i *>(sb + 16 * (2 * hf + q2));
                    c2[q2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bh2, zero, 0, 0, 0);
                    c1[q2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bh1, zero, 0, 0, 0);
                    c0[q2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bh0, sh, 0, 0, 0);
                }
#pragma unroll
                for (int q2 = 0; q2 < 2; q2++) {
                    const int qq = 2 * hf + q2;
                    v4i u;
#pragma unroll
                    for (int k = 0; k < 4; k++) u[k] = mf_comb3(c2[q2][k], c1[q2][k], c0[q2][k]);
                    const uint32_t t01 = __builtin_amdgcn_perm((uint32_t)u[1], (uint32_t)u[0], sel01);
                    const uint32_t t23 = __builtin_amdgcn_perm((uint32_t)u[3], (uint32_t)u[2], sel23);
                    *reinterpret_cast<uint32_t *>(t_w + (16 * qq) * P + 64 * qq + 16 * slot) = t01 | t23;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
        v4i c2[4], c1[4], c0[4];
#pragma unroll
        for (int qq = 0; qq < 4; qq++) {
            const v4i A = *reinterpret_cast<const v4i *>(sb + 16 * qq);
            c2[qq] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bh2, zero, 0, 0, 0);
            c1[qq] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bh1, zero, 0, 0, 0);
            c0[qq] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bh0, sh, 0, 0, 0);
        }
        if constexpr (SCORE) {   // source side of the box sums: the strip's own 16 px of these 16 rows (24 bytes into the window)
            const u32x2 lo = *reinterpret_cast<const u32x2 *>(sb + 24), hi = *reinterpret_cast<const u32x2 *>(sb + 32);
            const v4i A = {(int)lo.x, (int)lo.y, (int)hi.x, (int)hi.y};
            const v4i cb = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bbox, sbox, 0, 0, 0);
            const u32x4 ro = *reinterpret_cast<const u32x4 *>(s_rowh + 16 * s + 4 * g);
            if (!(a.dbg & 1)) mf_box_add4(tbl_s, ro, coln, cb);
        }
        v4i u[4];
#pragma unroll
        for (int qq = 0; qq < 4; qq++)
#pragma unroll
            for (int k = 0; k < 4; k++) u[qq][k] = mf_comb3(c2[qq][k], c1[qq][k], c0[qq][k]);
#pragma unroll
        for (int qq = 0; qq < 4; qq++) {
            const uint32_t t01 = __builtin_amdgcn_perm((uint32_t)u[qq][1], (uint32_t)u[qq][0], sel01);
            const uint32_t t23 = __builtin_amdgcn_perm((uint32_t)u[qq][3], (uint32_t)u[qq][2], sel23);
            *reinterpret_cast<uint32_t *>(t_w + (16 * qq) * P + 64 * qq + 16 * slot) = t01 | t23;
        }
        if constexpr (GUARD) {
            // one test for the lane's 16 samples: the smallest fraction field (alpha lanes: their seed puts 2^23 + alpha there).
            // Flagged samples are recomputed AFTER the provisional dword went to the ring and patched there byte by byte: the
            // accumulators are dead by then, so the out-of-line recomputation does not add its registers to theirs.
            uint32_t m[4];
#pragma unroll
            for (int qq = 0; qq < 4; qq++) {
                const uint32_t f0 = static_cast<uint32_t>(u[qq][0]) & 0x00ffffffu, f1 = static_cast<uint32_t>(u[qq][1]) & 0x00ffffffu;
                const uint32_t f2 = static_cast<uint32_t>(u[qq][2]) & 0x00ffffffu, f3 = static_cast<uint32_t>(u[qq][3]) & 0x00ffffffu;
                m[qq] = min(min(min(f0, f1), f2), f3);
            }
            const uint32_t mm = min(min(min(m[0], m[1]), m[2]), m[3]);
            if (__builtin_amdgcn_ballot_w64(mm < static_cast<uint32_t>(a.thr))) {   // ~1 set group in 10
                uint32_t fl = 0;                                                     // bit 4 qq + k
#pragma unroll
                for (int qq = 0; qq < 4; qq++)
#pragma unroll
                    for (int k = 0; k < 4; k++) fl |= ((static_cast<uint32_t>(u[qq][k]) & 0x00ffffffu) < static_cast<uint32_t>(a.thr) ? 1u : 0u) << (4 * qq + k);
                unsigned long long todo = __builtin_amdgcn_ballot_w64(fl != 0);
                while (todo) {                                                       // wave-uniform: one sample at a time, all lanes on it
                    const int L = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    uint32_t flL = __builtin_amdgcn_readlane(fl, L);
                    const int rL = L & 15, gL = L >> 4;
                    while (flL) {
                        const int b = __builtin_ctz(flL), qq = b >> 2, k = b & 3;
                        flL &= flL - 1;
                        const uint32_t e = mf_exact_u(wd, sbuf + (4 * gL + k) * SP + 4 * (16 * wave + 4 * qq + (rL >> 2)) + (rL & 3), 4, 0, 0xffff) ^ 0x80u;
                        if (lane == L) *(t_w + (16 * qq) * P + 64 * qq + 16 * slot + k) = static_cast<uint8_t>(e);
                    }
                }
            }
        }
    };
    // V set j: 16 output rows into out stage `buf`; odd sets find their first 16 staged rows in ring rows 16..31
    auto vset = [&](int j, int buf, auto oddt) {
        constexpr bool ODD = decltype(oddt)::value;
        long A[4];
        uint32_t al[4];
#pragma unroll
        for (int q = 0; q < 4; q++) A[q] = *reinterpret_cast<const long *>((ODD ? t_ro : t_r) + (4 * q) * P);
#pragma unroll
        for (int q = 0; q < 4; q++) al[q] = *((ODD ? t_ao : t_ae) + (4 * q) * P);
        int u[4][3];
        if constexpr (!GUARD && !SCORE) {   // plain fast blur: two halves of two column sets (see the H sets)
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {
                v4i c2[2], c1[2], c0[2];
#pragma unroll
                for (int q2 = 0; q2 < 2; q2++) {
                    c2[q2] = __builtin_amdgcn_mfma_i32_16x16x32_i8(A[2 * hf + q2], bv2, zero, 0, 0, 0);
                    c1[q2] = __builtin_amdgcn_mfma_i32_16x16x32_i8(A[2 * hf + q2], bv1, zero, 0, 0, 0);
                    c0[q2] = __builtin_amdgcn_mfma_i32_16x16x32_i8(A[2 * hf + q2], bv0, sv, 0, 0, 0);
                }
#pragma unroll
                for (int q2 = 0; q2 < 2; q2++)
#pragma unroll
                    for (int i = 0; i < 3; i++) u[2 * hf + q2][i] = mf_comb3(c2[q2][i], c1[q2][i], c0[q2][i]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            v4i c2[4], c1[4], c0[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                c2[q] = __builtin_amdgcn_mfma_i32_16x16x32_i8(A[q], bv2, zero, 0, 0, 0);
                c1[q] = __builtin_amdgcn_mfma_i32_16x16x32_i8(A[q], bv1, zero, 0, 0, 0);
                c0[q] = __builtin_amdgcn_mfma_i32_16x16x32_i8(A[q], bv0, sv, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int i = 0; i < 3; i++) u[q][i] = mf_comb3(c2[q][i], c1[q][i], c0[q][i]);
        }
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t t01 = __builtin_amdgcn_perm((uint32_t)u[q][1], (uint32_t)u[q][0], 0x0c0c0703u);
            const uint32_t t23 = __builtin_amdgcn_perm(al[q], (uint32_t)u[q][2], 0x04030c0cu);
            o[q] = t01 | t23;
        }
        uint8_t *op = s_out + buf * 16 * OP + o_w;
        *reinterpret_cast<u32x4 *>(op) = o;
        if constexpr (GUARD) {   // as in the H sets: provisional pixels to the out stage first, flagged bytes patched there
            uint32_t m[4];
#pragma unroll
            for (int q = 0; q < 4; q++)
                m[q] = min(min(static_cast<uint32_t>(u[q][0]) & 0x00ffffffu, static_cast<uint32_t>(u[q][1]) & 0x00ffffffu), static_cast<uint32_t>(u[q][2]) & 0x00ffffffu);
            const uint32_t mm = min(min(min(m[0], m[1]), m[2]), m[3]);
            if (__builtin_amdgcn_ballot_w64(mm < static_cast<uint32_t>(a.thr))) {
                uint32_t fl = 0;                                                     // bit 4 q + i
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int i = 0; i < 3; i++) fl |= ((static_cast<uint32_t>(u[q][i]) & 0x00ffffffu) < static_cast<uint32_t>(a.thr) ? 1u : 0u) << (4 * q + i);
                unsigned long long todo = __builtin_amdgcn_ballot_w64(fl != 0);
                while (todo) {
                    const int L = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    uint32_t flL = __builtin_amdgcn_readlane(fl, L);
                    const int rL = L & 15, gL = L >> 4;
                    while (flL) {
                        const int b = __builtin_ctz(flL), q = b >> 2, i = b & 3;
                        flL &= flL - 1;
                        const uint32_t e = mf_exact_u(wd, tw + (16 * gL + 4 * q + i) * P + 64 * gL, 1, (ODD ? 16 : 0) + rL, 31);
                        if (lane == L) op[4 * q + i] = static_cast<uint8_t>(e);
                    }
                }
                if constexpr (SCORE) o = *reinterpret_cast<const u32x4 *>(op);       // the box sums are those of the exact image
            }
        }
        if constexpr (SCORE) {   // blurred side: lane (row r, chunk g) holds 4 px of row r -- an A operand as it is
            const u32x4 os = o ^ 0x80808080u;
            const v4i A = {(int)os[0], (int)os[1], (int)os[2], (int)os[3]};
            const v4i cb = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bbox, sbox, 0, 0, 0);
            const u32x4 ro = *reinterpret_cast<const u32x4 *>(s_rowv + 16 * j + 4 * g);
            if (!(a.dbg & 1)) mf_box_add4(tbl_b, ro, coln, cb);
        }
    };

    // the march itself, in two forms: strips whose source window stays inside the image's columns (16-byte loads, rows
    // clamped where a set leaves the image) and the first / last strips (clamped px loads, masked stores)
    auto march = [&](auto xedget) {
        constexpr bool XEDGE = decltype(xedget)::value;
        // Interior strips: every lane issues both loads of every set, rows clamped per lane, no branch anywhere -- a load under
        // a branch (the few lanes of the second chunk, a set that touches the image's edge, the last sets of the march)
        // makes the compiler wait for ALL outstanding loads at the joins, and the march then runs one set ahead, not two.
        auto hload = [&](int i, u32x4 (&d)[2]) {
            const int ys = y0 - 6 + 16 * i;
            if constexpr (!XEDGE) {
                const uint8_t *sb = src + 4 * static_cast<ptrdiff_t>(x0 - 6);
                const int ya = clampi(ys + srow0, 0, a.h - 1), yb = clampi(ys + srow1, 0, a.h - 1);   // clamp-to-edge (effects.go:174-178, 200-204)
                d[0] = *(g_u32x4 *)(sb + static_cast<ptrdiff_t>(ya) * a.sstride + 16 * sch0);
                d[1] = *(g_u32x4 *)(sb + static_cast<ptrdiff_t>(yb) * a.sstride + 16 * sch1);
            } else {
    #pragma unroll
                for (int k = 0; k < 2; k++) {
                    if (k == 1 && !two) break;
                    const int y = clampi(ys + (k ? srow1 : srow0), 0, a.h - 1);
                    const uint8_t *rowp = src + static_cast<size_t>(y) * a.sstride;
                    const int xc = x0 - 6 + 4 * (k ? sch1 : sch0);
    #pragma unroll
                    for (int e = 0; e < 4; e++) d[k][e] = ld_px(rowp, clampi(xc + e, 0, a.w - 1));
                }
            }
        };
        auto out_store = [&](int j, int buf) {                          // V set j's 16 rows, from the out stage
            const u32x4 o = *reinterpret_cast<const u32x4 *>(s_out + buf * 16 * OP + o_r);
            const int y = y0 + 16 * j + orow;
            if (y < a.h) {
                uint8_t *dp = dst + static_cast<size_t>(y) * a.dstride + 4 * static_cast<size_t>(xo);
                if (!XEDGE || xo + 3 < a.w) *(g_u32x4w *)(dp) = o;
                else {
    #pragma unroll
                    for (int e = 0; e < 4; e++) if (xo + e < a.w) *(g_u32w *)(dp + 4 * e) = o[e];
                }
            }
        };
        // step s: [stage write of H set s | store of V set s-3]  barrier  [H set s -> ring slot s&1 | V set s-1 -> out stage s&1]
        u32x4 ra[2], rb[2];
        hload(0, ra);
        __builtin_amdgcn_sched_barrier(0);                              // ra's loads first: the loop's vmcnt waits count on the order
        hload(min(1, NI - 1), rb);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SCORE) score_setup();
        auto step = [&](int s, u32x4 (&d)[2], auto oddv) {
            constexpr int par = decltype(oddv)::value ? 0 : 1;          // s & 1 (V set s-1 is odd when s is even)
            if (s < NI) stage_write(d, par);
            if (XEDGE) { if (s + 2 < NI) hload(s + 2, d); } else hload(min(s + 2, NI - 1), d);
            if (s >= 3 && s - 3 < NJ) out_store(s - 3, par);           // written in step s-2 (same parity)
            __syncthreads();
            if (s < NI) hset(s, par, par);
            if (s >= 1 && s - 1 < NJ) vset(s - 1, par, oddv);
        };
    #pragma unroll 1
        for (int s = 0; s < NJ + 3; s += 2) {
            step(s, ra, std::true_type{});
            step(s + 1, rb, std::false_type{});
        }

    };
    if (xedge) march(std::true_type{}); else march(std::false_type{});

    if (SCORE && !(a.dbg & 2)) {   // the tile's slab: [0, slabn) source, [slabn, 2 slabn) blurred, 4 x u16 per entry (a box is <= 256 px)
        __syncthreads();
        unsigned long long *slab = a.slabs + (static_cast<size_t>(z) * a.tiles + tile) * 2 * slabn;
        for (int e = tid; e < 2 * slabn; e += 256) {
            const u32x4 v = *reinterpret_cast<const u32x4 *>(s_box + 4 * e);
            slab[e] = static_cast<unsigned long long>(v.x | (v.y << 16)) | (static_cast<unsigned long long>(v.z) << 32);
        }
    }
}

// ------------------------------------------------------------------------------------
// radii 7 .. 24 (sigma <= 8): the same march with a wider frame
// ------------------------------------------------------------------------------------
// H window 4 + 2 RF px = NKH chunks of 64 bytes: NKH chained matrix instructions per digit and set (RF = 14, 22, 24 for
// NKH = 2, 3, 4).  V: v_mfma_i32_16x16x64_i8 over a 64-row window (16 + 2 RF <= 64), the ring holds four 16-row slots, a V
// set runs three steps behind its first H set.  Same stages, same barrier per step, same exactness argument (the guard
// distance grows with the tap count: G ~ 3e-4 at 45 taps).  Plain blur only (the one-pass SSIMFast form stops at radius 6).
// r5: radii 25 .. 62 (sigma <= 20.67; the generic passes took 410 .. 820 us per 4K image from sigma = 8.4 on).  NKH = 5 .. 8 H
// chunks, and the V window as NKV = 2 / 3 chained 64-row instructions over a ring of NS = 4 NKV slots: 16 + 2 RF <= 64 NKV rows,
// a V set runs NS - 1 steps behind its first H set, its store two more.  NKV = 1 is the kernel as it was.
template <int NKH> struct MfWide {
    static constexpr int NKV = NKH <= 4 ? 1 : (NKH <= 7 ? 2 : 3);         // 64-row chunks of the V window
    static constexpr int RFH = 8 * NKH - 2, RFV = 32 * NKV - 8;
    static constexpr int RF = RFH < RFV ? RFH : RFV;                      // frame radius: taps sit centred in it (24 at NKH = 4)
    static constexpr int NC = 16 + RF / 2;                                // 16-byte chunks of a staged row: px x0 - RF .. x0 + 63 + RF
    static constexpr int NL = (16 * NC + 255) / 256;                      // chunk loads per lane and H set
    static constexpr int SP = 16 * NC + 48;                               // staged row pitch, conflict-free as the A operand (tools/lds_conflicts.py)
    static constexpr int NT = 2 * RF + 1;
    static constexpr int NS = 4 * NKV;                                    // ring slots of 16 rows
    static constexpr int RING = 16 * NS;
    static constexpr int P = RING + 16;                                   // ring: RING rows + 16 per byte column (= 16 mod 32)
    static constexpr int WT = 64 * P + 256;
};

// per-lane, out of line: what the widest tables keep (NKH >= 3: 37+ taps; inlined uniform chains of that length did not pay)
template <int NT>
__device__ __noinline__ uint32_t mf_exact_h_n(const uint8_t *p, const double *wd)
{
    double acc = 0;
#pragma unroll 15
    for (int t = 0; t < NT; t++) acc = acc + u8_to_f64(p[4 * t] ^ 0x80u) * wd[t];
    return clampF_dev(acc);
}
template <int NT, int RING = 64>
__device__ __noinline__ uint32_t mf_exact_v_n(const uint8_t *p, int ring0, const double *wd)
{
    double acc = 0;
#pragma unroll 15
    for (int t = 0; t < NT; t++) acc = acc + u8_to_f64(p[static_cast<unsigned>(ring0 + t) % static_cast<unsigned>(RING)] ^ 0x80u) * wd[t];
    return clampF_dev(acc);
}

// mf_exact_u for NT taps (the table is padded with zeros to a multiple of 8 doubles): eight weights per s_load
template <int NT>
__device__ __forceinline__ uint32_t mf_exact_u_n(const double *wd, const uint8_t *base, int step, int ring0, int mask)
{
    double acc = 0;
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += 8) {
        mf_s16i w;
        asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(w) : "s"(mf_scalar_ptr(wd)), "n"(8 * t0) : "memory");
        uint32_t v[8];
#pragma unroll
        for (int t = 0; t < 8; t++) v[t] = base[((ring0 + t0 + min(t, NT - 1 - t0)) & mask) * step];
#pragma unroll
        for (int t = 0; t < 8; t++)
            if (t0 + t < NT) acc = acc + u8_to_f64(v[t] ^ 0x80u) * __hiloint2double(w[2 * t + 1], w[2 * t]);
    }
    return clampF_dev(acc);
}

// r5, SCORE (NKH = 2: radii 7 .. 14): SSIMFast's box sums of the source and of the blurred image in the same pass, exactly as
// blur_mfma_kernel<SCORE> takes them -- two more matrix instructions per 16 x 16 px block and 4 + 4 LDS atomics; the strip's own
// 16 px of a staged row start 4 RF bytes into the window (8-byte aligned: two 8-byte reads make the A operand).
template <int NKH, bool GUARD, bool SCORE = false>
__global__ __launch_bounds__(256, MfWide<NKH>::NKV == 3 ? 1 : 2) void blur_mfma_wide_kernel(MfmaArgs a)   // (NKV = 3: 89 KB of LDS)
{
    using C = MfWide<NKH>;
    constexpr int RF = C::RF, NC = C::NC, SP = C::SP, P = C::P, WT = C::WT, OP = MF_OP, NT = C::NT;
    constexpr int NKV = C::NKV, NS = C::NS, RING = C::RING, NL = C::NL;
    static_assert(!SCORE || (4 * RF) % 8 == 0, "the source-side A operand is read as two 8-byte halves");
    static_assert(!SCORE || NKV == 1, "the one-pass form is NKH = 2's");
    __shared__ __attribute__((aligned(16))) uint8_t s_t[4 * WT];
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[2 * 16 * SP];
    __shared__ __attribute__((aligned(16))) uint8_t s_out[2 * 16 * OP];
    __shared__ __attribute__((aligned(16))) uint32_t s_rowh[SCORE ? MF_SEG_SCORE_WIDE + 48 : 4];   // (see blur_mfma_kernel)
    __shared__ __attribute__((aligned(16))) uint32_t s_rowv[SCORE ? MF_SEG_SCORE_WIDE + 16 : 4];
    __shared__ uint32_t s_colbox[SCORE ? 64 : 1];
    extern __shared__ __attribute__((aligned(16))) uint32_t s_box[];

    const int tile = xcd_tile(blockIdx.x, a.tiles);
    if (tile < 0) return;
    const int z = blockIdx.y;
    const uint8_t *src = a.srcs ? a.srcs[z] : a.src;
    uint8_t *dst = a.dsts ? a.dsts[z] : a.dst;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int x0 = tx * 64, y0 = ty * a.seg;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, g = lane >> 4;
    const int NJ = min(a.seg, ((a.h - y0 + 15) >> 4) << 4) >> 4;
    const int NI = NJ + NS - 1;                                     // a V set reads the NS H sets from its own on
    const bool xedge = x0 - RF < 0 || x0 + 64 + RF > a.w;

    const v4i *tbh = reinterpret_cast<const v4i *>(a.tab);
    v4i bh[NKH][3];
#pragma unroll
    for (int kk = 0; kk < NKH; kk++)
#pragma unroll
        for (int l = 0; l < 3; l++) bh[kk][l] = tbh[(3 * kk + l) * 64 + lane];   // digits hi, mid, lo
    const v4i *tbv = tbh + 3 * NKH * 64;
    v4i bv[NKV][3];
#pragma unroll
    for (int c = 0; c < NKV; c++)
#pragma unroll
        for (int l = 0; l < 3; l++) bv[c][l] = tbv[(3 * c + l) * 64 + lane];                // digits hi, mid, lo
    const double *wd = reinterpret_cast<const double *>(tbv + 3 * NKV * 64);
    const bool alane = (r & 3) == 3;
    const int seed_hl = alane ? (1 << 23) + 128 : a.seed_h;
    const v4i sh = {seed_hl, seed_hl, seed_hl, seed_hl}, sv = {a.seed_v, a.seed_v, a.seed_v, a.seed_v};
    const v4i zero = {0, 0, 0, 0};
    const uint32_t sel01 = alane ? 0x0c0c0400u : 0x0c0c0703u, sel23 = alane ? 0x04000c0cu : 0x07030c0cu;

    uint8_t *tw = s_t + wave * WT;
    // chunk ids tid, 256 + tid, ...: (lanes past the last chunk fetch it again and drop it: see hload)
    int srow[NL], sch[NL], st_w[NL];
    bool have[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) {
        const int id = min(256 * k + tid, 16 * NC - 1);
        srow[k] = id / NC; sch[k] = id - NC * srow[k];
        st_w[k] = srow[k] * SP + 16 * sch[k];
        have[k] = 256 * k + tid < 16 * NC;
    }
    const int st_r = r * SP + 64 * wave + 16 * g;                   // A operand of H set qq, K chunk kk: + 16 qq + 64 kk
    uint8_t *t_w = tw + r * P + 4 * g;                              // + (16 qq) P + 64 qq + 16 slot
    const int m4 = r >> 2, mi = r & 3;
    const uint8_t *t_r = tw + (16 * m4 + mi) * P + 64 * m4;         // + 4 q P + 16 ((j + 4 c + g) mod NS)
    const uint8_t *t_a = tw + (16 * g + 3) * P + 64 * g;            // + 4 q P + ((16 j + r + RF) mod RING)
    const int o_w = r * OP + 64 * wave + 16 * g;
    const int orow = tid >> 4, och = tid & 15;
    const int o_r = orow * OP + 16 * och;
    const int xo = x0 + 4 * och;

    // ---- SCORE set-up (blur_mfma_kernel's, with the wide frame's row offset) ----
    v4i bbox = {0, 0, 0, 0}, sbox = {0, 0, 0, 0};
    uint32_t coln = 0;
    const int slabn = SCORE ? (a.nbx + 1) * (a.nby + 1) : 0;
    uint32_t *tbl_s = s_box, *tbl_b = s_box + 4 * slabn;
    auto score_setup = [&]() {
        for (int e = tid; e < 8 * slabn; e += 256) s_box[e] = 0;
        const int rowbytes = 16 * (a.nbx + 1);
        const int b0y = a.by[y0];
        for (int u = tid; u < 16 * NI; u += 256) {                  // staged row u = tile row u - RF
            const int t = u - RF;
            const int v = (t >= 0 && t < a.seg && y0 + t < a.h) ? a.by[y0 + t] : -1;
            s_rowh[u] = rowbytes * ((v >= 0 && b0y >= 0) ? v - b0y : a.nby);
        }
        for (int t = tid; t < 16 * NJ; t += 256) {
            const int v = (t < a.seg && y0 + t < a.h) ? a.by[y0 + t] : -1;
            s_rowv[t] = rowbytes * ((v >= 0 && b0y >= 0) ? v - b0y : a.nby);
        }
        if (tid < 64) {
            const int b0x = a.bx[x0], v = x0 + tid < a.w ? a.bx[x0 + tid] : -1;
            s_colbox[tid] = (v >= 0 && b0x >= 0) ? v - b0x : 255u;
        }
        __syncthreads();
        uint32_t first = 255u;
        for (int i = 0; i < 16; i++) first = min(first, s_colbox[16 * wave + i]);
        const int slot = r / 3, ch = r - 3 * slot;
        int cnt = 0;
        for (int i = 0; i < 16; i++) cnt += (r < 15 && s_colbox[16 * wave + i] == first + slot) ? 1 : 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const bool in = r < 15 && first != 255u && s_colbox[16 * wave + 4 * g + e] == first + slot;
            bbox[e] = in ? (1 << (8 * ch)) : 0;
        }
        const int seed = 128 * cnt;
        sbox = (v4i){seed, seed, seed, seed};
        const uint32_t bc = (r < 15 && first != 255u && cnt > 0) ? first + slot : static_cast<uint32_t>(a.nbx);
        coln = 16u * bc + 4u * (r < 15 ? ch : 3);
    };

    auto stage_write = [&](const u32x4 (&d)[NL], int buf) {
        uint8_t *sb = s_stage + buf * 16 * SP;
        *reinterpret_cast<u32x4 *>(sb + st_w[0]) = d[0] ^ 0x80808080u;
#pragma unroll
        for (int k = 1; k < NL; k++)
            if (have[k]) *reinterpret_cast<u32x4 *>(sb + st_w[k]) = d[k] ^ 0x80808080u;
    };
    auto hset = [&](int s, int buf, int slot) {
        const uint8_t *sbuf = s_stage + buf * 16 * SP;
        const uint8_t *sb = sbuf + st_r;
        v4i c2[4], c1[4], c0[4];
#pragma unroll
        for (int qq = 0; qq < 4; qq++) { c2[qq] = zero; c1[qq] = zero; c0[qq] = sh; }
        if constexpr (SCORE) {   // source side of the box sums: the strip's own 16 px of these 16 staged rows
            const u32x2 lo = *reinterpret_cast<const u32x2 *>(sb + 4 * RF), hi = *reinterpret_cast<const u32x2 *>(sb + 4 * RF + 8);
            const v4i A = {(int)lo.x, (int)lo.y, (int)hi.x, (int)hi.y};
            const v4i cb = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bbox, sbox, 0, 0, 0);
            const u32x4 ro = *reinterpret_cast<const u32x4 *>(s_rowh + 16 * s + 4 * g);
#pragma unroll
            for (int k = 0; k < 4; k++)
                __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(tbl_s) + ro[k] + coln), static_cast<uint32_t>(cb[k]),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#pragma unroll
        for (int kk = 0; kk < NKH; kk++)
#pragma unroll
            for (int qq = 0; qq < 4; qq++) {
                const v4i A = *reinterpret_cast<const v4i *>(sb + 16 * qq + 64 * kk);
                c2[qq] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bh[kk][0], c2[qq], 0, 0, 0);
                c1[qq] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bh[kk][1], c1[qq], 0, 0, 0);
                c0[qq] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bh[kk][2], c0[qq], 0, 0, 0);
            }
        v4i u[4];
#pragma unroll
        for (int qq = 0; qq < 4; qq++)
#pragma unroll
            for (int k = 0; k < 4; k++) u[qq][k] = mf_comb3(c2[qq][k], c1[qq][k], c0[qq][k]);
#pragma unroll
        for (int qq = 0; qq < 4; qq++) {
            const uint32_t t01 = __builtin_amdgcn_perm((uint32_t)u[qq][1], (uint32_t)u[qq][0], sel01);
            const uint32_t t23 = __builtin_amdgcn_perm((uint32_t)u[qq][3], (uint32_t)u[qq][2], sel23);
            *reinterpret_cast<uint32_t *>(t_w + (16 * qq) * P + 64 * qq + 16 * slot) = t01 | t23;
        }
        if constexpr (GUARD) {
            uint32_t m[4];
#pragma unroll
            for (int qq = 0; qq < 4; qq++) {
                const uint32_t f0 = static_cast<uint32_t>(u[qq][0]) & 0x00ffffffu, f1 = static_cast<uint32_t>(u[qq][1]) & 0x00ffffffu;
                const uint32_t f2 = static_cast<uint32_t>(u[qq][2]) & 0x00ffffffu, f3 = static_cast<uint32_t>(u[qq][3]) & 0x00ffffffu;
                m[qq] = min(min(min(f0, f1), f2), f3);
            }
            const uint32_t mm = min(min(min(m[0], m[1]), m[2]), m[3]);
            if (__builtin_amdgcn_ballot_w64(mm < static_cast<uint32_t>(a.thr))) {
                uint32_t fl = 0;
#pragma unroll
                for (int qq = 0; qq < 4; qq++)
#pragma unroll
                    for (int k = 0; k < 4; k++) fl |= ((static_cast<uint32_t>(u[qq][k]) & 0x00ffffffu) < static_cast<uint32_t>(a.thr) ? 1u : 0u) << (4 * qq + k);
                if constexpr (NKH > 2) {
                    while (fl) {
                        const int b = __builtin_ctz(fl), qq = b >> 2, k = b & 3;
                        fl &= fl - 1;
                        const uint32_t e = mf_exact_h_n<NT>(sbuf + (4 * g + k) * SP + 4 * (16 * wave + 4 * qq + (r >> 2)) + (r & 3), wd) ^ 0x80u;
                        *(t_w + (16 * qq) * P + 64 * qq + 16 * slot + k) = static_cast<uint8_t>(e);
                    }
                }
                unsigned long long todo = NKH > 2 ? 0ull : __builtin_amdgcn_ballot_w64(fl != 0);
                while (todo) {                                       // wave-uniform (blur_mfma_kernel's fix-ups)
                    const int L = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    uint32_t flL = __builtin_amdgcn_readlane(fl, L);
                    const int rL = L & 15, gL = L >> 4;
                    while (flL) {
                        const int b = __builtin_ctz(flL), qq = b >> 2, k = b & 3;
                        flL &= flL - 1;
                        const uint32_t e = mf_exact_u_n<NT>(wd, sbuf + (4 * gL + k) * SP + 4 * (16 * wave + 4 * qq + (rL >> 2)) + (rL & 3), 4, 0, 0xffff) ^ 0x80u;
                        if (lane == L) *(t_w + (16 * qq) * P + 64 * qq + 16 * slot + k) = static_cast<uint8_t>(e);
                    }
                }
            }
        }
    };
    auto vset = [&](int j, int buf) {
        uint32_t al[4];
        const int jm = static_cast<int>(static_cast<unsigned>(j) % static_cast<unsigned>(NS));   // (wave-uniform)
        const int ra_ = static_cast<int>(static_cast<unsigned>(16 * jm + r + RF) % static_cast<unsigned>(RING));
#pragma unroll
        for (int q = 0; q < 4; q++) al[q] = *(t_a + (4 * q) * P + ra_);
        v4i c2[4], c1[4], c0[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { c2[q] = zero; c1[q] = zero; c0[q] = sv; }
#pragma unroll
        for (int c = 0; c < NKV; c++) {
            const int ro = 16 * static_cast<int>(static_cast<unsigned>(jm + 4 * c + g) % static_cast<unsigned>(NS));
            v4i A[4];
#pragma unroll
            for (int q = 0; q < 4; q++) A[q] = *reinterpret_cast<const v4i *>(t_r + (4 * q) * P + ro);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                c2[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[q], bv[c][0], c2[q], 0, 0, 0);
                c1[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[q], bv[c][1], c1[q], 0, 0, 0);
                c0[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[q], bv[c][2], c0[q], 0, 0, 0);
            }
        }
        int u[4][3];
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int i = 0; i < 3; i++) u[q][i] = mf_comb3(c2[q][i], c1[q][i], c0[q][i]);
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t t01 = __builtin_amdgcn_perm((uint32_t)u[q][1], (uint32_t)u[q][0], 0x0c0c0703u);
            const uint32_t t23 = __builtin_amdgcn_perm(al[q], (uint32_t)u[q][2], 0x04030c0cu);
            o[q] = t01 | t23;
        }
        uint8_t *op = s_out + buf * 16 * OP + o_w;
        *reinterpret_cast<u32x4 *>(op) = o;
        if constexpr (GUARD) {
            uint32_t m[4];
#pragma unroll
            for (int q = 0; q < 4; q++)
                m[q] = min(min(static_cast<uint32_t>(u[q][0]) & 0x00ffffffu, static_cast<uint32_t>(u[q][1]) & 0x00ffffffu), static_cast<uint32_t>(u[q][2]) & 0x00ffffffu);
            const uint32_t mm = min(min(min(m[0], m[1]), m[2]), m[3]);
            if (__builtin_amdgcn_ballot_w64(mm < static_cast<uint32_t>(a.thr))) {
                uint32_t fl = 0;
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int i = 0; i < 3; i++) fl |= ((static_cast<uint32_t>(u[q][i]) & 0x00ffffffu) < static_cast<uint32_t>(a.thr) ? 1u : 0u) << (4 * q + i);
                if constexpr (NKH > 2) {
                    while (fl) {
                        const int b = __builtin_ctz(fl), q = b >> 2, i = b & 3;
                        fl &= fl - 1;
                        op[4 * q + i] = static_cast<uint8_t>(mf_exact_v_n<NT, RING>(tw + (16 * g + 4 * q + i) * P + 64 * g, 16 * jm + r, wd));
                    }
                }
                unsigned long long todo = NKH > 2 ? 0ull : __builtin_amdgcn_ballot_w64(fl != 0);
                while (todo) {
                    const int L = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    uint32_t flL = __builtin_amdgcn_readlane(fl, L);
                    const int rL = L & 15, gL = L >> 4;
                    while (flL) {
                        const int b = __builtin_ctz(flL), q = b >> 2, i = b & 3;
                        flL &= flL - 1;
                        const uint32_t e = mf_exact_u_n<NT>(wd, tw + (16 * gL + 4 * q + i) * P + 64 * gL, 1, 16 * j + rL, 63);   // (NKH = 2: a 64-row ring)
                        if (lane == L) op[4 * q + i] = static_cast<uint8_t>(e);
                    }
                }
                if constexpr (SCORE) o = *reinterpret_cast<const u32x4 *>(op);       // the box sums are those of the exact image
            }
        }
        if constexpr (SCORE) {   // blurred side: lane (row r, chunk g) holds 4 px of row r -- an A operand as it is
            const u32x4 os = o ^ 0x80808080u;
            const v4i A = {(int)os[0], (int)os[1], (int)os[2], (int)os[3]};
            const v4i cb = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bbox, sbox, 0, 0, 0);
            const u32x4 ro = *reinterpret_cast<const u32x4 *>(s_rowv + 16 * j + 4 * g);
#pragma unroll
            for (int k = 0; k < 4; k++)
                __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(tbl_b) + ro[k] + coln), static_cast<uint32_t>(cb[k]),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };

    auto march = [&](auto xedget) {
        constexpr bool XEDGE = decltype(xedget)::value;
        // (interior strips: both loads of every set from every lane, rows clamped per lane, no branch -- blur_mfma_kernel's hload)
        auto hload = [&](int i, u32x4 (&d)[NL]) {
            const int ys = y0 - RF + 16 * i;
            if constexpr (!XEDGE) {
                const uint8_t *sb = src + 4 * static_cast<ptrdiff_t>(x0 - RF);
#pragma unroll
                for (int k = 0; k < NL; k++) {
                    const int yk = clampi(ys + srow[k], 0, a.h - 1);
                    d[k] = *(g_u32x4 *)(sb + static_cast<ptrdiff_t>(yk) * a.sstride + 16 * sch[k]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < NL; k++) {
                    if (k > 0 && !have[k]) break;
                    const int y = clampi(ys + srow[k], 0, a.h - 1);
                    const uint8_t *rowp = src + static_cast<size_t>(y) * a.sstride;
                    const int xc = x0 - RF + 4 * sch[k];
#pragma unroll
                    for (int e = 0; e < 4; e++) d[k][e] = ld_px(rowp, clampi(xc + e, 0, a.w - 1));
                }
            }
        };
        auto out_store = [&](int j, int buf) {
            const u32x4 o = *reinterpret_cast<const u32x4 *>(s_out + buf * 16 * OP + o_r);
            const int y = y0 + 16 * j + orow;
            if (y < a.h) {
                uint8_t *dp = dst + static_cast<size_t>(y) * a.dstride + 4 * static_cast<size_t>(xo);
                if (!XEDGE || xo + 3 < a.w) *(g_u32x4w *)(dp) = o;
                else {
#pragma unroll
                    for (int e = 0; e < 4; e++) if (xo + e < a.w) *(g_u32w *)(dp + 4 * e) = o[e];
                }
            }
        };
        // step s: [stage write of H set s | store of V set s-NS-1]  barrier  [H set s -> ring slot s mod NS | V set s-NS+1 -> out stage s&1]
        u32x4 ra[NL], rb[NL];
        hload(0, ra);
        __builtin_amdgcn_sched_barrier(0);
        hload(1, rb);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SCORE) score_setup();
        auto step = [&](int s, u32x4 (&d)[NL], auto part) {
            constexpr int par = decltype(part)::value;
            if (s < NI) stage_write(d, par);
            if (XEDGE) { if (s + 2 < NI) hload(s + 2, d); } else hload(min(s + 2, NI - 1), d);
            if (s >= NS + 1 && s - (NS + 1) < NJ) out_store(s - (NS + 1), par);
            __syncthreads();
            if (s < NI) hset(s, par, static_cast<int>(static_cast<unsigned>(s) % static_cast<unsigned>(NS)));
            if (s >= NS - 1 && s - (NS - 1) < NJ) vset(s - (NS - 1), par);
        };
#pragma unroll 1
        for (int s = 0; s < NJ + NS + 1; s += 2) {
            step(s, ra, std::integral_constant<int, 0>{});
            step(s + 1, rb, std::integral_constant<int, 1>{});
        }
    };
    if (xedge) march(std::true_type{}); else march(std::false_type{});

    if constexpr (SCORE) {   // the tile's slab (blur_mfma_kernel's)
        __syncthreads();
        unsigned long long *slab = a.slabs + (static_cast<size_t>(z) * a.tiles + tile) * 2 * slabn;
        for (int e = tid; e < 2 * slabn; e += 256) {
            const u32x4 v = *reinterpret_cast<const u32x4 *>(s_box + 4 * e);
            slab[e] = static_cast<unsigned long long>(v.x | (v.y << 16)) | (static_cast<unsigned long long>(v.z) << 32);
        }
    }
}

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------
// Fixed-point form of a blur kernel: wq[k] = round(w[k] 2^24), the centre takes what is left of 2^24.
// Returns false when the table is outside what the three-digit form or the error bound covers.
constexpr int MF_RWIDE = 62;         // blur_mfma_wide_kernel: 16 + 2 R <= the 64 NKV rows of its V window (NKH = 8: RF = 62)
struct MfmaWeights {
    long long wq[2 * MF_RWIDE + 1];
    double err255;                // 255 * max(sum of the positive, sum of the negative wq[k] - w[k] 2^24): bound of |S - exact sum * 2^24|
};
static bool mfma_quantise(const double *kernel, int radius, MfmaWeights *q, int rmax = MF_RMAX)
{
    if (radius < 1 || radius > rmax) return false;
    const int nt = 2 * radius + 1;
    double sum = 0;
    for (int i = 0; i < nt; i++) {
        if (!(kernel[i] >= 0) || !(kernel[i] < 0.49)) return false;   // three signed digits reach 127 * 65793 = 0.498 * 2^24
        sum += kernel[i];
    }
    if (!(std::fabs(sum - 1.0) <= 1e-6)) return false;
    long long tot = 0;
    for (int i = 0; i < nt; i++) {
        q->wq[i] = std::llround(kernel[i] * 16777216.0);
        tot += q->wq[i];
    }
    q->wq[radius] += 16777216 - tot;
    if (q->wq[radius] < 0 || q->wq[radius] > 8355711) return false;
    // S - (exact sum) 2^24 = sum_k (wq[k] - w[k] 2^24) p[k] with 0 <= p[k] <= 255: between -255 (sum of the negative
    // differences) and +255 (sum of the positive ones) -- half of 255 sum |.| when the differences cancel, as they nearly do
    long double ep = 0, en = 0;
    for (int i = 0; i < nt; i++) {
        const long double d = static_cast<long double>(q->wq[i]) - static_cast<long double>(kernel[i]) * 16777216.0L;
        if (d > 0) ep += d; else en -= d;
    }
    q->err255 = static_cast<double>(255.0L * std::max(ep, en));
    return true;
}
static void mfma_digits(long long v, int d[3])
{
    for (int i = 0; i < 3; i++) {
        long long lo = ((v % 256) + 256) % 256;
        if (lo >= 128) lo -= 256;
        d[i] = static_cast<int>(lo);
        v = (v - lo) / 256;
    }
}

// device table: BH[3][64] x 16 bytes (digits hi, mid, lo) | BV[3][64] x 8 | 13 fp64 weights
constexpr size_t MF_TAB_WORDS = 3 * 64 * 4 + 3 * 64 * 2;
constexpr size_t MF_TAB_ALL = MF_TAB_WORDS + 2 * 16;   // + the fp64 weights: 13, padded to 16 with zeros (mf_exact_u's two s_load_dwordx16)
static void mfma_build_table(const MfmaWeights &q, int radius, uint32_t *tab)
{
    std::fill(tab, tab + MF_TAB_WORDS, 0u);
    int8_t *bh = reinterpret_cast<int8_t *>(tab);
    int8_t *bv = reinterpret_cast<int8_t *>(tab + 3 * 64 * 4);
    const int nt = 2 * radius + 1, off = MF_RMAX - radius;
    for (int lane = 0; lane < 64; lane++) {
        const int n = lane & 15, kc = lane >> 4;
        for (int b = 0; b < 16; b++) {   // H: K index 16 kc + b = byte of the 64-byte window; output byte n = 4 px + channel
            const int px = 4 * kc + b / 4, ch = b % 4, c = n % 4, pj = n / 4;
            const int t = px - pj - off;
            int d[3] = {0, 0, 0};
            if (ch == c && c < 3 && t >= 0 && t < nt) mfma_digits(q.wq[t], d);
            if (ch == c && c == 3 && px == pj + MF_RMAX) d[0] = 1;   // alpha column: the centre pixel's alpha as it is (effects.go:189)
            for (int l = 0; l < 3; l++) bh[((2 - l) * 64 + lane) * 16 + b] = static_cast<int8_t>(d[l]);
        }
        for (int b = 0; b < 8; b++) {    // V: K index 8 kc + b = staged row of the set's 32; output row n
            const int te = 8 * kc + b - n - off;
            int de[3] = {0, 0, 0};
            if (te >= 0 && te < nt) mfma_digits(q.wq[te], de);
            for (int l = 0; l < 3; l++) bv[((2 - l) * 64 + lane) * 8 + b] = static_cast<int8_t>(de[l]);
        }
    }
}

// FNX_BLUR_EXACT takes this kernel too (FNX_BLUR_MFMA_EXACT=0: blur.hip's guarded fp32 kernel, kept for A/B runs):
// 4K plain 18.7 us against 23.8, one-pass 26.4 against 29.2 per image
bool blur_mfma_exact_enabled()
{
    static const bool on = [] { const char *e = dev_env("FNX_BLUR_MFMA_EXACT"); return !(e && e[0] == '0'); }();
    return on;
}

bool blur_mfma_covers(const double *kernel, int radius, int w, int h)
{
    static const bool off = [] { const char *e = dev_env("FNX_BLUR_MFMA"); return e && e[0] == '0'; }();
    if (off || w < 64 || h < 32) return false;
    MfmaWeights q;
    return mfma_quantise(kernel, radius, &q);
}
static bool blur_mfma_wide_covers(const double *kernel, int radius, int w, int h)
{
    static const bool off = [] { const char *e = dev_env("FNX_BLUR_MFMA"); return e && e[0] == '0'; }();
    if (off || w < 64 || h < 32 || radius <= MF_RMAX) return false;
    MfmaWeights q;
    return mfma_quantise(kernel, radius, &q, MF_RWIDE);
}

// does launch_blur_mfma take this call?  (blur.hip's one-pass form asks: its fast-mode images must be the two-call route's)
bool blur_mfma_takes(const double *kernel, int radius, int w, int h, bool exact)
{
    if (exact && !blur_mfma_exact_enabled()) return false;
    return blur_mfma_covers(kernel, radius, w, h) || blur_mfma_wide_covers(kernel, radius, w, h);
}

// Rows per workgroup.  A workgroup of NJ 16-row sets runs NJ + 4 steps (the pipeline's fill and drain); the grid runs in
// rounds of occ workgroups per CU.  The segment count that minimises rounds x steps: long segments for batches (the halo
// is 22 staged rows per segment), one full round of short ones for a single image (4K alone: 17 segments of 128 rows,
// 18 us against 23 with "at least eight workgroups per CU").
int blur_mfma_segment(const fnx_ctx *ctx, int n, int w, int h, int cap, int occ)
{
    const long tiles_x = (w + 63) / 64, slots = static_cast<long>(occ) * ctx->num_cus;
    const int sets = (h + 15) / 16;
    int best_seg = 16, first = 1;
    double best = 0;
    for (int nseg = std::max(1, (h + cap - 1) / cap); nseg <= std::max(1, sets / 2); nseg++) {
        const int nj = (sets + nseg - 1) / nseg, seg = 16 * nj;
        if (seg > cap) continue;
        const long segs = (h + seg - 1) / seg, wgs = tiles_x * segs * n;
        const double cost = static_cast<double>((wgs + slots - 1) / slots) * (nj + 4);
        if (first || cost < best * 0.98) { best = cost; best_seg = seg; first = 0; }   // ties go to the longer segment
    }
    return best_seg;
}

template <bool SCORE, bool GUARD>
static int launch_mfma_cfg(fnx_ctx *ctx, int n, MfmaArgs &ma, size_t lds)
{
    ma.tiles_x = (ma.w + 63) / 64;
    ma.tiles = ma.tiles_x * ((ma.h + ma.seg - 1) / ma.seg);
    dim3 grid(8 * ((ma.tiles + 7) / 8), n);
    // the launch's events ride on its own packet (common.hpp: LaunchEvents); SCORE launches always carry a stop event:
    // the step's tail on the ctx's second stream waits for it (blur.hip: launch_blur_scored)
    LaunchEvents ev;
    FNX_TRY(prof_bind(ctx, FNX_PROF_MAIN, &ev));
    if constexpr (SCORE) {
        if (!ev.stop) ev.stop = ctx->ev_blur[ctx->parity];
        ctx->blur_done = ev.stop;
    }
    note_route(ctx, FNX_PROF_MAIN, SCORE ? (GUARD ? "blur_mfma_kernel<SCORE, GUARD>" : "blur_mfma_kernel<SCORE>")
                                         : (GUARD ? "blur_mfma_kernel<GUARD>" : "blur_mfma_kernel"));
    hipExtLaunchKernelGGL((blur_mfma_kernel<SCORE, GUARD>), grid, dim3(256), lds, ctx->stream, ev.start, ev.stop, 0, ma);
    FNX_HIP(hipGetLastError());
    return FNX_OK;
}

// weights, seeds and the guard distance of one call; the matrix table goes to its own table slot
static int mfma_prepare(fnx_ctx *ctx, const double *kernel, int radius, bool exact, MfmaArgs *ma)
{
    MfmaWeights q;
    if (!mfma_quantise(kernel, radius, &q)) return FNX_NOOP;
    uint32_t tab[MF_TAB_ALL] = {};
    mfma_build_table(q, radius, tab);
    memcpy(reinterpret_cast<double *>(tab + MF_TAB_WORDS) + (MF_RMAX - radius), kernel, sizeof(double) * (2 * radius + 1));   // centred, zeros around
    void *dt = nullptr;
    FNX_TRY(upload_table(ctx, SLOT_TABLE_MFMA, tab, sizeof(tab), &dt));
    ma->tab = static_cast<const uint32_t *>(dt);
#ifdef FNX_DEVELOP                     // a development build only (make DEVELOP=1): bits 1 / 2 / 4 skip the box sums -- wrong scores
    { static const int dbg = [] { const char *e = dev_env("FNX_MFMA_DBG"); return e ? atoi(e) : 0; }(); ma->dbg = dbg; }
#else
    ma->dbg = 0;
#endif
    ma->radius = radius;
    // G (units of 2^-24): the fixed-point error bound, the reference's own fp64 chain error (13 roundings below 256:
    // < 4e-13 = 7e-6 units) and one unit for the bound's own arithmetic
    const long long gq = exact ? static_cast<long long>(std::ceil(q.err255)) + 2 : 0;
    if (gq > (1 << 20)) return FNX_NOOP;
    ma->seed_h = static_cast<int>((1u << 23) + static_cast<uint32_t>(gq));                 // staged bytes come out as (value ^ 0x80)
    ma->seed_v = static_cast<int>((1u << 23) + (1u << 31) + static_cast<uint32_t>(gq));    // plain bytes
    ma->thr = static_cast<int>(2 * gq);
    return FNX_OK;
}

// blur_mfma_wide_kernel's table: BH[NKH][3][64] x 16 bytes | BV[3][64] x 16 bytes | 2 RF + 1 fp64 weights, centred
template <int NKH, bool SCORE = false>
static int launch_mfma_wide(fnx_ctx *ctx, int n, MfmaArgs &ma, const MfmaWeights &q, const double *kernel, int radius, bool exact, size_t lds = 0)
{
    using C = MfWide<NKH>;
    constexpr int RF = C::RF;
    constexpr int NKV = C::NKV;
    constexpr size_t words = (3 * NKH + 3 * NKV) * 64 * 4 + 2 * ((C::NT + 7) / 8 * 8);   // (weights padded to whole s_load_dwordx16's)
    std::vector<uint32_t> tab(words, 0u);
    int8_t *bh = reinterpret_cast<int8_t *>(tab.data());
    int8_t *bv = bh + 3 * NKH * 64 * 16;
    const int nt = 2 * radius + 1, off = RF - radius;
    for (int lane = 0; lane < 64; lane++) {
        const int nn = lane & 15, kc = lane >> 4;
        for (int kk = 0; kk < NKH; kk++)
            for (int b = 0; b < 16; b++) {   // H: K index 64 kk + 16 kc + b = byte of the window
                const int px = 16 * kk + 4 * kc + b / 4, ch = b % 4, c = nn % 4, pj = nn / 4;
                const int t = px - pj - off;
                int d[3] = {0, 0, 0};
                if (ch == c && c < 3 && t >= 0 && t < nt) mfma_digits(q.wq[t], d);
                if (ch == c && c == 3 && px == pj + RF) d[0] = 1;
                for (int l = 0; l < 3; l++) bh[((3 * kk + (2 - l)) * 64 + lane) * 16 + b] = static_cast<int8_t>(d[l]);
            }
        for (int c = 0; c < NKV; c++)
            for (int b = 0; b < 16; b++) {   // V: K index 64 c + 16 kc + b = staged row of the set's 64 NKV; output row nn
                const int t = 64 * c + 16 * kc + b - nn - off;
                int d[3] = {0, 0, 0};
                if (t >= 0 && t < nt) mfma_digits(q.wq[t], d);
                for (int l = 0; l < 3; l++) bv[((3 * c + (2 - l)) * 64 + lane) * 16 + b] = static_cast<int8_t>(d[l]);
            }
    }
    memcpy(reinterpret_cast<double *>(tab.data() + (3 * NKH + 3 * NKV) * 64 * 4) + off, kernel, sizeof(double) * nt);
    void *dt = nullptr;
    FNX_TRY(upload_table(ctx, SLOT_TABLE_MFMA, tab.data(), sizeof(uint32_t) * words, &dt));
    ma.tab = static_cast<const uint32_t *>(dt);
    ma.radius = radius;
    ma.tiles_x = (ma.w + 63) / 64;
    ma.tiles = ma.tiles_x * ((ma.h + ma.seg - 1) / ma.seg);
    dim3 grid(8 * ((ma.tiles + 7) / 8), n);
    LaunchEvents ev;
    FNX_TRY(prof_bind(ctx, FNX_PROF_MAIN, &ev));
    if constexpr (SCORE) {               // (launch_mfma_cfg: the step's tail on the ctx's second stream waits for this event)
        if (!ev.stop) ev.stop = ctx->ev_blur[ctx->parity];
        ctx->blur_done = ev.stop;
        note_route(ctx, FNX_PROF_MAIN, exact ? "blur_mfma_wide_kernel<SCORE, GUARD>" : "blur_mfma_wide_kernel<SCORE>");
    } else {
        note_route(ctx, FNX_PROF_MAIN, exact ? "blur_mfma_wide_kernel<GUARD>" : "blur_mfma_wide_kernel");
    }
    if (exact) hipExtLaunchKernelGGL((blur_mfma_wide_kernel<NKH, true, SCORE>), grid, dim3(256), lds, ctx->stream, ev.start, ev.stop, 0, ma);
    else hipExtLaunchKernelGGL((blur_mfma_wide_kernel<NKH, false, SCORE>), grid, dim3(256), lds, ctx->stream, ev.start, ev.stop, 0, ma);
    FNX_HIP(hipGetLastError());
    return FNX_OK;
}

// GaussianBlur of n images; FNX_NOOP (nothing launched): the table or the shape is not this kernel's
int launch_blur_mfma(fnx_ctx *ctx, int n, const uint8_t *src, const uint8_t *const *srcs, int sstride, int w, int h,
                     const double *kernel, int radius, int flags, uint8_t *dst, uint8_t *const *dsts, int dstride)
{
    const bool exact = flags & FNX_BLUR_EXACT;
    if (exact && !blur_mfma_exact_enabled()) return FNX_NOOP;
    if (n > 65535) return FNX_NOOP;
    if (blur_mfma_wide_covers(kernel, radius, w, h)) {     // radius 7 .. 24
        MfmaWeights q;
        mfma_quantise(kernel, radius, &q, MF_RWIDE);
        const long long gq = exact ? static_cast<long long>(std::ceil(q.err255)) + 2 : 0;
        if (gq > (1 << 20)) return FNX_NOOP;
        MfmaArgs ma{};
        ma.src = src; ma.srcs = srcs; ma.dst = dst; ma.dsts = dsts;
        ma.sstride = sstride; ma.dstride = dstride; ma.w = w; ma.h = h;
        ma.seed_h = static_cast<int>((1u << 23) + static_cast<uint32_t>(gq));
        ma.seed_v = static_cast<int>((1u << 23) + (1u << 31) + static_cast<uint32_t>(gq));
        ma.thr = static_cast<int>(2 * gq);
        ma.seg = blur_mfma_segment(ctx, n, w, h, 544, 3);
        if (radius <= MfWide<2>::RF) return launch_mfma_wide<2>(ctx, n, ma, q, kernel, radius, exact);
        if (radius <= MfWide<3>::RF) return launch_mfma_wide<3>(ctx, n, ma, q, kernel, radius, exact);
        if (radius > MfWide<4>::RF) {        // r5: 25 .. 62, the V window in two or three chained instructions
            if (radius <= MfWide<5>::RF) return launch_mfma_wide<5>(ctx, n, ma, q, kernel, radius, exact);
            if (radius <= MfWide<6>::RF) return launch_mfma_wide<6>(ctx, n, ma, q, kernel, radius, exact);
            if (radius <= MfWide<7>::RF) return launch_mfma_wide<7>(ctx, n, ma, q, kernel, radius, exact);
            return launch_mfma_wide<8>(ctx, n, ma, q, kernel, radius, exact);
        }
        return launch_mfma_wide<4>(ctx, n, ma, q, kernel, radius, exact);
    }
    if (!blur_mfma_covers(kernel, radius, w, h)) return FNX_NOOP;
    MfmaArgs ma{};
    const int st = mfma_prepare(ctx, kernel, radius, exact, &ma);
    if (st != FNX_OK) return st;
    ma.src = src; ma.srcs = srcs; ma.dst = dst; ma.dsts = dsts;
    ma.sstride = sstride; ma.dstride = dstride; ma.w = w; ma.h = h;
    static const int cap = [] { const char *e = dev_env("FNX_MFMA_SEG"); return e ? atoi(e) : 544; }();   // development: rows per workgroup
    ma.seg = blur_mfma_segment(ctx, n, w, h, cap, exact ? 3 : 4);
    return exact ? launch_mfma_cfg<false, true>(ctx, n, ma, 0) : launch_mfma_cfg<false, false>(ctx, n, ma, 0);
}

// the one-pass form: blur + the tile slabs of both box-sum sides (blur.hip's launch_blur_scored owns the geometry)
int launch_blur_mfma_scored(fnx_ctx *ctx, int n, const uint8_t *const *srcs, int sstride, int w, int h, const double *kernel,
                            int radius, int flags, uint8_t *const *dsts, int dstride, const int32_t *bx, const int32_t *by,
                            unsigned long long *slabs, int nbx, int nby, int seg)
{
    const bool exact = flags & FNX_BLUR_EXACT;
    MfmaArgs ma{};
    const int st = mfma_prepare(ctx, kernel, radius, exact, &ma);
    if (st != FNX_OK) return st;
    ma.srcs = srcs; ma.dsts = dsts;
    ma.sstride = sstride; ma.dstride = dstride; ma.w = w; ma.h = h;
    ma.seg = seg;
    ma.bx = bx; ma.by = by; ma.slabs = slabs; ma.nbx = nbx; ma.nby = nby;
    const size_t lds = sizeof(uint32_t) * 8 * static_cast<size_t>(nbx + 1) * (nby + 1);
    return exact ? launch_mfma_cfg<true, true>(ctx, n, ma, lds) : launch_mfma_cfg<true, false>(ctx, n, ma, lds);
}

// radii 7 .. 24 (r5): the same through blur_mfma_wide_kernel<2 / 3 / 4, ., SCORE>
bool blur_mfma_wide_scored_covers(const double *kernel, int radius, int w, int h, bool exact)
{
    if (radius > MfWide<4>::RF || (exact && !blur_mfma_exact_enabled())) return false;      // (r5: 7 .. 14, then 15 .. 24)
    return blur_mfma_wide_covers(kernel, radius, w, h);
}
int launch_blur_mfma_wide_scored(fnx_ctx *ctx, int n, const uint8_t *const *srcs, int sstride, int w, int h, const double *kernel,
                                 int radius, int flags, uint8_t *const *dsts, int dstride, const int32_t *bx, const int32_t *by,
                                 unsigned long long *slabs, int nbx, int nby, int seg)
{
    const bool exact = flags & FNX_BLUR_EXACT;
    if (!blur_mfma_wide_scored_covers(kernel, radius, w, h, exact) || seg > MF_SEG_SCORE_WIDE) return FNX_NOOP;
    MfmaWeights q;
    mfma_quantise(kernel, radius, &q, MF_RWIDE);
    const long long gq = exact ? static_cast<long long>(std::ceil(q.err255)) + 2 : 0;
    if (gq > (1 << 20)) return FNX_NOOP;
    MfmaArgs ma{};
    ma.srcs = srcs; ma.dsts = dsts;
    ma.sstride = sstride; ma.dstride = dstride; ma.w = w; ma.h = h;
    ma.seed_h = static_cast<int>((1u << 23) + static_cast<uint32_t>(gq));
    ma.seed_v = static_cast<int>((1u << 23) + (1u << 31) + static_cast<uint32_t>(gq));
    ma.thr = static_cast<int>(2 * gq);
    ma.seg = seg;
    ma.bx = bx; ma.by = by; ma.slabs = slabs; ma.nbx = nbx; ma.nby = nby;
    const size_t lds = sizeof(uint32_t) * 8 * static_cast<size_t>(nbx + 1) * (nby + 1);
    if (radius <= MfWide<2>::RF) return launch_mfma_wide<2, true>(ctx, n, ma, q, kernel, radius, exact, lds);
    if (radius <= MfWide<3>::RF) return launch_mfma_wide<3, true>(ctx, n, ma, q, kernel, radius, exact, lds);
    return launch_mfma_wide<4, true>(ctx, n, ma, q, kernel, radius, exact, lds);
}

}  // namespace fnx

extern "C" int fnx_blur_fixed_point(const double *kernel, int radius, long long *wq, double *err255)
{
    if (!kernel || !wq || !err255) {
        fnx::set_error("invalid argument: fnx_blur_fixed_point");
        return FNX_ERR_INVALID;
    }
    fnx::MfmaWeights q;
    if (!fnx::mfma_quantise(kernel, radius, &q, fnx::MF_RWIDE)) return FNX_NOOP;       // (r5: the wide kernels' radii 7 .. 62 too)
    for (int i = 0; i < 2 * radius + 1; i++) wq[i] = q.wq[i];
    *err255 = q.err255;
    return FNX_OK;
}
