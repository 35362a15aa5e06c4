// lanczosResize (resize.go:37-161) with both passes on the gfx950 i8 matrix pipe: opaque images, scale ratios up to ~2.2.
// Same idea as blur_mfma.hip, with matrices that change from output group to output group and PLANAR channels.
//
// What the reference computes for an output whose window is opaque (every A = 255; resize.go:93-113 / 137-156):
//     clampF(fl(r * inv)),  r = sum_k fl(R_k * aw_k),  aw_k = fl(255 w_k),  inv = fl(1 / sum_k aw_k)
// whose value is within 1e-12 of X = sum_k R_k W_k, W_k = aw_k inv (a real number).  Here W_k becomes the integer
// Wq_k ~ W_k 2^S (S = 23 where every |W_k| < 0.996, else 22), the roundings chosen so that sum_k Wq_k = 2^S exactly (largest
// remainders), split into three signed base-256 digits; the bytes go in as R - 128 (R ^ 0x80 read as int8), so
//     u = sum_k Wq_k (R_k - 128) + (128 + 64) 2^S + 2^(S-1) + G  =  (X' + 64) 2^S + 2^(S-1) + G
// is an EXACT uint32 (X' + 64 in 0 .. 511: the host checks the negative lobes) with
// -255 sum_k (W_k - Wq_k 2^-S)^+ <= X' - X <= 255 sum_k (Wq_k 2^-S - W_k)^+ (0 <= R_k <= 255; the host computes both per
// output; G >= the largest + 2 units).  The output byte is sat_u8((u >> S) - 64) -- floor(X + 1/2) clamped, clampF --
// unless u's S fraction bits lie below 2 G: such a sample (one in ~10 000 on photographs) is recomputed
// in fp64 in the reference's own order and patched.  Proven, not sampled -- the rounding-guard argument of resize.hip with
// an integer sum in place of fp32 FMAs.
//
// A workgroup owns 64 output columns and a run of 16-row output groups; it marches down the SOURCE 16 rows ("slot") at a
// time.  Staging splits the RGBA chunks into three byte planes (eight v_perm per chunk; the alphas are only checked): with
// interleaved channels an output byte meets its own channel in a quarter of the window, a 4-px group needed a 128-byte
// window and its own 6 KB matrix, and the 96 registers of matrices per wave left two workgroups per CU
// (experiments/mfma/resize_mfma_interleaved.hip).  Planar, one 3 KB matrix serves a wave's 16 output px in all channels.
//  * H set of a slot, per plane: A = staged rows (M = row, K = the 64 px of the wave's window), B = the banded weights
//    (registers, for the whole march), C = 4 consecutive rows of one output px per lane = one dword of the transposed
//    uint8 intermediate T[plane][px][row], a four-slot ring per wave.
//  * V set of an output group, run as soon as its last slot is in the ring, per plane: A = T (M = px, K = the 64 rows of
//    slots vb .. vb + 3), B = the group's weights (LDS), C = 4 px of one output row per lane; the planes are interleaved
//    again (eight v_perm) on the way to the output stage.
// Loads and stores are workgroup-wide through LDS stages, as in the blur.
//
// What it does NOT handle it hands back: a workgroup that meets a pixel with A != 255, or a set dense with flagged
// samples (a linear ramp at an integer ratio puts every output on an exact tie), stops and marks the tiles of
// resize_fused_kernel (resize.hip) that cover its region; resize_fused_sparse_kernel runs afterwards on marked tiles only
// and overwrites them in the reference's arithmetic.  Both are bit-exact, so the overlap is harmless.
//
// Where it stands (profiles/r04_time_resize_*.txt): 4K -> 1080p on photo-like content 33 us against resize_fused_kernel's
// 38; 1080p -> 4K 37.5 against 39.4, and slower than it on that shape when the input is the downscaled ramp.  The matrix
// instructions are a tenth of the time: what is left per sample -- two v_lshl_add to join the digits, 1.75 to clamp and
// pack, 2 for the guard -- is more than the 3.5 packed FMAs of a 7-tap upscale and not far below the 6.5 of a 13-tap
// downscale.  So resize_fused takes this route for DOWNSCALES (both ratios >= 1.25) and leaves the rest where it was;
// FNX_RESIZE_MFMA=2 takes it wherever the tables allow (tests), =0 never.
#include <hip/hip_ext.h>

#include "common.hpp"
#include "devutil.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <type_traits>
#include <vector>

namespace fnx {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned short v2us __attribute__((ext_vector_type(2)));

// (fixed point of the weights: 2^23 where every |W| stays below 0.996 -- three signed digits reach 0.498 * 2^24 --, else 2^22;
// per table, RzMfTable::S)
constexpr int RM_P = 80;                 // ring: 64 rows + 16 per column (blur_mfma_wide_kernel's)
constexpr int RM_WT = 48 * RM_P + 192;   // per wave: 3 planes x 16 columns, each plane's group skewed by 64 bytes
constexpr int RM_OP = 272;
constexpr int RM_NCMAX = 48;             // 16-byte source chunks of a staged row (3 loads per lane)
constexpr int RM_MAXJ = 8;               // output groups per workgroup (their matrices live in LDS)
constexpr int RM_MAXTAPS = 16;
#ifndef RM_DEPTH
#define RM_DEPTH 2
#endif
constexpr int RM_DENSE = 16;             // flagged lanes in one set from which the workgroup gives up

struct RmH { int hb, mat; };
struct RmEx { int n, s0; double inv; double aw[16]; };   // taps, first source index, 1 / sum aw, aw = 255 w (zeros behind the taps)
static_assert(sizeof(RmEx) == 144, "scalar loads below");
typedef int s4i __attribute__((ext_vector_type(4)));
typedef int s16i __attribute__((ext_vector_type(16)));
struct RmV { int vb, need, mat, pad; };

struct RzMfArgs {
    const uint8_t *src;
    uint8_t *dst;
    int sstride, dstride, srcW, srcH, dstW, dstH;
    int tiles_x, tiles, segj, nvg;
    int NC, SP;                          // chunks per staged row; pitch of a staged PLANE row (bytes = px)
    const v4i *hmat;
    const RmH *hmeta;                    // per H group (16 outputs): window px offset in the staged row, matrix index
    const int32_t *sbase;                // per strip: source px of the staged row's first chunk (a multiple of 16)
    const v4i *vmat;
    const RmV *vmeta;                    // per V group (16 output rows): first slot, last slot, matrix index
    int seed_h, thr_h, seed_v, thr_v;
    int sh_h, sh_v;                      // S - 16 of each pass
    const RmEx *exh, *exv;               // per output: the reference's own operands for the fp64 fix-ups
    uint32_t *todo;
    unsigned *gave_up;                   // + 1 per workgroup that hands its region back
    // a batch of same-geometry images (blockIdx.y = image): device arrays of their pointers, and the images' spacing in `todo`
    const uint8_t *const *srcs;
    uint8_t *const *dsts;
    uint32_t todo_stride;
    uint32_t gen;
    int old_tw, old_th, old_gx;
};

__device__ __forceinline__ int rm_comb3(int hi, int mid, int lo)
{
    int t = hi * 256 + mid;
    asm volatile("" : "+v"(t));
    return t * 256 + lo;
}

// sat_u8((x >> S) - 64) of two sums u = (X + 64) 2^S + ... (unsigned: X + 64 is 0 .. 511): their high halves side by side, a
// logical shift of both, the offset off, a saturating pack
__device__ __forceinline__ uint32_t rm_bytes2(int u1, int u0, int sh)
{
    const uint32_t hi = __builtin_amdgcn_perm(static_cast<uint32_t>(u1), static_cast<uint32_t>(u0), 0x07060302u);
    v2us h;
    __builtin_memcpy(&h, &hi, 4);
    h = h >> static_cast<unsigned short>(sh);
    v2s hsg;
    __builtin_memcpy(&hsg, &h, 4);
    hsg = hsg - static_cast<short>(64);
    uint32_t hs, o;
    __builtin_memcpy(&hs, &hsg, 4);
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(o) : "v"(hs));
    return o;
}
__device__ __forceinline__ uint32_t rm_bytes4(const v4i &u, int sh)
{
    return __builtin_amdgcn_perm(rm_bytes2(u[3], u[2], sh), rm_bytes2(u[1], u[0], sh), 0x05040100u);
}
__device__ __forceinline__ uint32_t rm_minfrac(const v4i &u, uint32_t frac)
{
    const uint32_t f0 = static_cast<uint32_t>(u[0]) & frac, f1 = static_cast<uint32_t>(u[1]) & frac;
    const uint32_t f2 = static_cast<uint32_t>(u[2]) & frac, f3 = static_cast<uint32_t>(u[3]) & frac;
    return min(min(min(f0, f1), f2), f3);
}

// One flagged output channel in the reference's own arithmetic, opaque window (resize.go:95-112 / 139-155):
// aw = 255 w; r = r + R aw, taps ascending; clampF(r * inv).  UNIFORM: the whole wave runs the chain on one sample -- its
// record comes through the scalar cache (three s_load, one wait: a per-lane version chased two dependent global loads,
// ~2 us, while the workgroup's other waves stood at the barrier), its bytes (R ^ 0x80) are LDS broadcasts from
// base[(s0 + off0 + t) & mask].  Always 16 taps: the record pads aw with +0.0 and adding R * 0.0 changes nothing.
__device__ __forceinline__ uint32_t rm_exact_u(const RmEx *ex, const uint8_t *base, int off0, int mask)
{
    s4i hd;
    s16i w0, w1;
    asm volatile("s_load_dwordx4 %0, %3, 0x0\n\ts_load_dwordx16 %1, %3, 0x10\n\ts_load_dwordx16 %2, %3, 0x50\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(hd), "=&s"(w0), "=&s"(w1) : "s"(ex) : "memory");
    const int i0 = hd[1] + off0;
    uint32_t v[16];
#pragma unroll
    for (int t = 0; t < 16; t++) v[t] = base[(i0 + t) & mask];
    double r = 0;
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const int lo = t < 8 ? w0[2 * (t & 7)] : w1[2 * (t & 7)], hi = t < 8 ? w0[2 * (t & 7) + 1] : w1[2 * (t & 7) + 1];
        const double aw = __hiloint2double(hi, lo);
        r = r + u8_to_f64(v[t] ^ 0x80u) * aw;
    }
    const double inv = __hiloint2double(hd[3], hd[2]);
    return clampF_dev(r * inv);
}

// Dynamic LDS: per-wave rings | output stage | the workgroup's V matrices | two source stages of three planes
template <int NL>
__global__ __launch_bounds__(256, 3) void resize_mfma_kernel(RzMfArgs a)
{
    constexpr int P = RM_P, WT = RM_WT, OP = RM_OP;
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    __shared__ RmV s_vm[RM_MAXJ];
    __shared__ int s_bad[2];
    uint8_t *s_t = s_dyn;
    uint8_t *s_out = s_dyn + 4 * WT;
    v4i *s_vmat = reinterpret_cast<v4i *>(s_out + 2 * 16 * OP);
    uint8_t *s_stage = reinterpret_cast<uint8_t *>(s_vmat + a.segj * 192);

    if (a.srcs) {                                                  // a batch: this workgroup's image
        a.src = a.srcs[blockIdx.y];
        a.dst = a.dsts[blockIdx.y];
        a.todo += static_cast<size_t>(blockIdx.y) * a.todo_stride;
    }
    const int tile = xcd_tile(blockIdx.x, a.tiles);
    if (tile < 0) return;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int x0 = 64 * tx;
    const int j0 = ty * a.segj, J = min(a.segj, a.nvg - j0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    if (tid < J) s_vm[tid] = a.vmeta[j0 + tid];                     // (the H record and matrix below are fetched meanwhile)
    if (tid < 2) s_bad[tid] = 0;
    const int SP = a.SP, NC = a.NC, STG = 48 * SP;
    const int spx0 = a.sbase[tx];

    const RmH hm = a.hmeta[4 * tx + wave];
    v4i bh[3];
    {
        const v4i *mp = a.hmat + static_cast<size_t>(hm.mat) * 192 + lane;
#pragma unroll
        for (int l = 0; l < 3; l++) bh[l] = mp[l * 64];
    }
    const int st_r = r * SP + hm.hb + 16 * g;                       // A operand of plane c: + 16 c SP
    const v4i sh = {a.seed_h, a.seed_h, a.seed_h, a.seed_h}, sv = {a.seed_v, a.seed_v, a.seed_v, a.seed_v};
    const v4i zero = {0, 0, 0, 0};
    const uint32_t frac_h = (0x10000u << a.sh_h) - 1u, frac_v = (0x10000u << a.sh_v) - 1u;

    uint8_t *tw = s_t + wave * WT;
    int srow[NL], sch[NL], st_w[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) {
        // every lane loads (lanes past the staged rows' 16 NC chunks fetch the last chunk again and park it in the spare
        // dword behind row 0 of each plane): loads under a branch make the compiler wait for ALL outstanding loads at every join
        const int id = min(tid + 256 * k, 16 * NC - 1);
        srow[k] = id / NC;
        sch[k] = id - NC * srow[k];
        st_w[k] = tid + 256 * k < 16 * NC ? srow[k] * SP + 4 * sch[k] : 4 * NC;
    }
    uint8_t *t_w = tw + r * P + 4 * g;                              // plane c: + 16 c P + 64 c; + 16 slot
    const uint8_t *t_r = tw + r * P;                                // plane c: + 16 c P + 64 c; + 16 ((rel + g) & 3)
    const int o_w = r * OP + 64 * wave + 16 * g;
    const int orow = tid >> 4, och = tid & 15;
    const int o_r = orow * OP + 16 * och;
    const int xo = x0 + 4 * och;

    __syncthreads();
    const int S0 = s_vm[0].vb, SL = s_vm[J - 1].need;               // source slots S0 .. SL
    // the workgroup's V matrices: every fetch is issued before the first is waited for (a workgroup lives for a dozen
    // iterations: one round trip per matrix, one after the other, was a third of its life), parked in LDS inside march()
    constexpr int NVM = (RM_MAXJ * 192 + 255) / 256;
    v4i vmt[NVM];
#pragma unroll
    for (int k = 0; k < NVM; k++) {
        const int e = min(tid + 256 * k, J * 192 - 1), j = e / 192;
        vmt[k] = a.vmat[static_cast<size_t>(s_vm[j].mat) * 192 + (e - 192 * j)];
    }
    int it = 0;                                                     // barriers passed

    // H set of one slot: per plane, 16 rows x this wave's 16 output px
    auto hset = [&](int par, int slot) {
        const uint8_t *sbuf = s_stage + par * STG;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const v4i A = *reinterpret_cast<const v4i *>(sbuf + 16 * c * SP + st_r);
            const v4i c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bh[0], zero, 0, 0, 0);
            const v4i c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bh[1], zero, 0, 0, 0);
            const v4i c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bh[2], sh, 0, 0, 0);
            v4i u;
#pragma unroll
            for (int k = 0; k < 4; k++) u[k] = rm_comb3(c2[k], c1[k], c0[k]);
            uint8_t *tp = t_w + 16 * c * P + 64 * c + 16 * slot;
            *reinterpret_cast<uint32_t *>(tp) = rm_bytes4(u, a.sh_h) ^ 0x80808080u;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(rm_minfrac(u, frac_h) < static_cast<uint32_t>(a.thr_h));
            if (bal) {
                if (__builtin_popcountll(bal) > RM_DENSE) {
                    s_bad[(it + 1) & 1] = 1;
                } else {
                    uint32_t fl = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) fl |= ((static_cast<uint32_t>(u[k]) & frac_h) < static_cast<uint32_t>(a.thr_h) ? 1u : 0u) << k;
                    unsigned long long todo = __builtin_amdgcn_ballot_w64(fl != 0);
                    while (todo) {
                        const int L = __builtin_ctzll(todo);
                        todo &= todo - 1;
                        uint32_t flL = __builtin_amdgcn_readlane(fl, L);
                        const int rL = L & 15, gL = L >> 4;
                        while (flL) {
                            const int k = __builtin_ctz(flL);
                            flL &= flL - 1;
                            const uint32_t e = rm_exact_u(a.exh + (x0 + 16 * wave + rL), sbuf + (16 * c + 4 * gL + k) * SP, -spx0, 0xffff) ^ 0x80u;
                            if (lane == L) tp[k] = static_cast<uint8_t>(e);
                        }
                    }
                }
            }
        }
    };
    // V set of one output group: per plane, 16 output rows x this wave's 16 px; the planes meet again in the output stage
    auto vset = [&](int jl, int ob) {
        const int ro = 16 * ((s_vm[jl].vb - S0 + g) & 3);
        uint8_t *op = s_out + ob * 16 * OP + o_w;
        const v4i b2 = s_vmat[jl * 192 + lane], b1 = s_vmat[jl * 192 + 64 + lane], b0 = s_vmat[jl * 192 + 128 + lane];
        uint32_t pl[3], fl = 0;
        bool dense = false;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const v4i A = *reinterpret_cast<const v4i *>(t_r + 16 * c * P + 64 * c + ro);
            const v4i c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, b2, zero, 0, 0, 0);
            const v4i c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, b1, zero, 0, 0, 0);
            const v4i c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, b0, sv, 0, 0, 0);
            v4i u;
#pragma unroll
            for (int i = 0; i < 4; i++) u[i] = rm_comb3(c2[i], c1[i], c0[i]);
            pl[c] = rm_bytes4(u, a.sh_v);
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(rm_minfrac(u, frac_v) < static_cast<uint32_t>(a.thr_v));
            if (bal) {
                if (__builtin_popcountll(bal) > RM_DENSE) { s_bad[(it + 1) & 1] = 1; dense = true; }
#pragma unroll
                for (int i = 0; i < 4; i++) fl |= ((static_cast<uint32_t>(u[i]) & frac_v) < static_cast<uint32_t>(a.thr_v) ? 1u : 0u) << (4 * c + i);
            }
        }
        // [R0 R1 R2 R3] [G0 ..] [B0 ..] -> four RGBA px, A = 255 (resize.go:112: clampF(a), the host checked it)
        const uint32_t t0 = __builtin_amdgcn_perm(pl[1], pl[0], 0x05010400u), t1 = __builtin_amdgcn_perm(pl[1], pl[0], 0x07030602u);
        const uint32_t u0 = __builtin_amdgcn_perm(pl[2], pl[2], 0x0d010d00u), u1 = __builtin_amdgcn_perm(pl[2], pl[2], 0x0d030d02u);
        u32x4 o;
        o[0] = __builtin_amdgcn_perm(u0, t0, 0x05040100u);
        o[1] = __builtin_amdgcn_perm(u0, t0, 0x07060302u);
        o[2] = __builtin_amdgcn_perm(u1, t1, 0x05040100u);
        o[3] = __builtin_amdgcn_perm(u1, t1, 0x07060302u);
        *reinterpret_cast<u32x4 *>(op) = o;
        unsigned long long todo = dense ? 0ull : __builtin_amdgcn_ballot_w64(fl != 0);
        while (todo) {
            const int L = __builtin_ctzll(todo);
            todo &= todo - 1;
            uint32_t flL = __builtin_amdgcn_readlane(fl, L);
            const int rL = L & 15, gL = L >> 4;
            while (flL) {
                const int b = __builtin_ctz(flL), c = b >> 2, i = b & 3;
                flL &= flL - 1;
                const uint32_t e = rm_exact_u(a.exv + (16 * (j0 + jl) + rL), tw + (16 * c + 4 * gL + i) * P + 64 * c, -16 * S0, 63);
                if (lane == L) op[4 * i + c] = static_cast<uint8_t>(e);
            }
        }
    };
    auto out_store = [&](int jl, int ob) {
        const u32x4 o = *reinterpret_cast<const u32x4 *>(s_out + ob * 16 * OP + o_r);
        const int y = 16 * (j0 + jl) + orow;
        if (y < a.dstH) {
            uint8_t *dp = a.dst + static_cast<size_t>(y) * a.dstride + 4 * static_cast<size_t>(xo);
            if (xo + 3 < a.dstW) *(g_u32x4w *)(dp) = o;
            else {
#pragma unroll
                for (int e = 0; e < 4; e++) if (xo + e < a.dstW) *(g_u32w *)(dp + 4 * e) = o[e];
            }
        }
    };

    auto march = [&]() -> bool {
        // (the host places every strip's staged columns inside the row: no edge form)
        auto hload = [&](int sabs, u32x4 (&d)[NL]) {
            const int ys = 16 * sabs;
#pragma unroll
            for (int k = 0; k < NL; k++) {
                const int y = min(ys + srow[k], a.srcH - 1);
                d[k] = *(g_u32x4 *)(a.src + static_cast<size_t>(y) * a.sstride + 4 * static_cast<ptrdiff_t>(spx0 + 4 * sch[k]));
            }
        };
        // four RGBA px -> one dword of each plane (R ^ 0x80 ...), the alphas checked on the way
        auto stage_write = [&](const u32x4 (&d)[NL], int par) {
            uint8_t *sb = s_stage + par * STG;
            uint32_t am = 0xffffffffu;
#pragma unroll
            for (int k = 0; k < NL; k++) {
                const uint32_t t0 = __builtin_amdgcn_perm(d[k][1], d[k][0], 0x05010400u), t1 = __builtin_amdgcn_perm(d[k][3], d[k][2], 0x05010400u);
                const uint32_t u0 = __builtin_amdgcn_perm(d[k][1], d[k][0], 0x07030602u), u1 = __builtin_amdgcn_perm(d[k][3], d[k][2], 0x07030602u);
                const uint32_t R4 = __builtin_amdgcn_perm(t1, t0, 0x05040100u), G4 = __builtin_amdgcn_perm(t1, t0, 0x07060302u);
                const uint32_t B4 = __builtin_amdgcn_perm(u1, u0, 0x05040100u), A4 = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
                am &= A4;
                *reinterpret_cast<uint32_t *>(sb + st_w[k]) = R4 ^ 0x80808080u;
                *reinterpret_cast<uint32_t *>(sb + 16 * SP + st_w[k]) = G4 ^ 0x80808080u;
                *reinterpret_cast<uint32_t *>(sb + 32 * SP + st_w[k]) = B4 ^ 0x80808080u;
            }
            if (am != 0xffffffffu) s_bad[it & 1] = 1;               // a pixel that is not opaque: read after the coming barrier
        };
        int s = S0, jl = 0, pend = -1, pob = 0, ob = 0;
        // after barrier `it`: s_bad[it & 1] holds what was raised before it; what the sets raise now goes to the other cell
        auto post = [&]() -> bool {
            if (s_bad[it & 1]) return false;
            if (pend >= 0) { out_store(pend, pob); pend = -1; }
            return true;
        };
        auto vready = [&]() { return jl < J && __builtin_amdgcn_readfirstlane(s_vm[min(jl, J - 1)].need) < s; };
        auto vrun = [&]() { vset(jl, ob); pend = jl; pob = ob; ob ^= 1; jl++; };
        // Loads run RM_DEPTH slots ahead of the slot being staged, in RM_DEPTH register sets used round-robin (the loop
        // body is unrolled once per set: the sets must be named statically): an iteration is a few hundred cycles of work
        // and a load takes thousands, so with two sets every iteration waited out most of a memory round trip
        u32x4 rs[RM_DEPTH][NL];
#pragma unroll
        for (int k = 0; k < RM_DEPTH; k++) {
            hload(min(S0 + k, SL), rs[k]);
            __builtin_amdgcn_sched_barrier(0);                      // in slot order: the loop's vmcnt waits count on it
        }
#pragma unroll
        for (int k = 0; k < NVM; k++)
            if (tid + 256 * k < J * 192) s_vmat[tid + 256 * k] = vmt[k];
        // one source slot: 0 = go on, 1 = that was the last, -1 = give up
        auto body = [&](u32x4 (&d)[NL]) -> int {
            stage_write(d, (s - S0) & 1);
            hload(min(s + RM_DEPTH, SL), d);
            __syncthreads();
            if (!post()) return -1;
            hset((s - S0) & 1, (s - S0) & 3);
            s++;
            if (vready()) vrun();
            it++;
            while (vready()) {
                __syncthreads();
                if (!post()) return -1;
                vrun();
                it++;
            }
            return s > SL ? 1 : 0;
        };
        int rc = 0;
#pragma unroll 1
        for (;;) {
            if ((rc = body(rs[0]))) break;
            if ((rc = body(rs[1]))) break;
#if RM_DEPTH > 2
            if ((rc = body(rs[2]))) break;
#endif
#if RM_DEPTH > 3
            if ((rc = body(rs[3]))) break;
#endif
#if RM_DEPTH > 4
            if ((rc = body(rs[4]))) break;
            if ((rc = body(rs[5]))) break;
#endif
        }
        if (rc < 0) return false;
        __syncthreads();
        return post();
    };
    const bool done = march();
    if (!done) {
        // hand the region back: resize_fused_kernel's tiles (old_tw x old_th output px) over columns x0 .. x0 + 63, this run of rows
        const int y_first = 16 * j0, y_last = min(16 * (j0 + J), a.dstH) - 1;
        const int by0 = y_first / a.old_th, by1 = y_last / a.old_th, bx = x0 / a.old_tw;
        for (int b = by0 + tid; b <= by1; b += 256) a.todo[b * a.old_gx + bx] = a.gen;
        if (tid == 0) atomicAdd(a.gave_up, 1u);
    }
}

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------
static void rm_digits(long long v, int d[3])
{
    for (int i = 0; i < 3; i++) {
        long long lo = ((v % 256) + 256) % 256;
        if (lo >= 128) lo -= 256;
        d[i] = static_cast<int>(lo);
        v = (v - lo) / 256;
    }
}

// 0: never, 1: downscales (the default), 2: wherever the tables allow
static int resize_mfma_mode(const fnx_ctx *ctx)
{
    const char *e = form_value(ctx, FORM_RESIZE_MFMA);
    return e && e[0] >= '0' && e[0] <= '2' ? e[0] - '0' : 1;
}

void resize_mfma_free(RzMfTable *t)
{
    if (t->blob) (void)hipFree(t->blob);
    *t = RzMfTable();
}

// The matrix form of one tap table; false (nothing allocated): outside what the kernel covers.
// `inv`: 1 / a per output as the guard form computed it (resize.hip: build_guard, a = sum of 255 w in tap order).
bool resize_mfma_build(const fnx_ctx *ctx, const TapTable &t, int srcN, bool vertical, const double *inv, RzMfTable *out)
{
    *out = RzMfTable();
    const int mode = resize_mfma_mode(ctx);
    const int nout = t.nout;
    if (mode == 0 || nout < 16 || srcN < 16) return false;
    if (mode == 1 && 4 * srcN < 5 * nout) return false;              // (the header's last paragraph)
    // fixed-point weights per output
    std::vector<std::vector<long long>> wq(nout);
    std::vector<int> first(nout), cnt(nout);
    long double emax = 0, wmax = 0;
    std::vector<std::vector<long double>> Wr(nout);
    for (int d = 0; d < nout; d++) {
        const int t0 = t.off[d], n = t.off[d + 1] - t0;
        if (n < 1 || n > RM_MAXTAPS) return false;
        for (int k = 1; k < n; k++)
            if (t.idx[t0 + k] != t.idx[t0] + k) return false;
        first[d] = t.idx[t0];
        cnt[d] = n;
        double a = 0;
        for (int k = 0; k < n; k++) a += 255.0 * t.wt[t0 + k];
        if (!(a >= 254.5) || !(a < 255.5)) return false;              // clampF(a) must be 255 (resize.go:112): the kernel writes that
        Wr[d].resize(n);
        long double neg = 0;
        for (int k = 0; k < n; k++) {
            Wr[d][k] = static_cast<long double>(255.0 * t.wt[t0 + k]) * static_cast<long double>(inv[d]);
            wmax = std::max(wmax, fabsl(Wr[d][k]));
            if (Wr[d][k] < 0) neg -= Wr[d][k];
        }
        if (255.0L * neg > 63.0L) return false;                       // the sums carry an offset of 64: X >= -64
    }
    const int S = wmax * 8388608.0L <= 8355000.0L ? 23 : 22;
    const long double scale = ldexpl(1.0L, S);
    const long long one = 1LL << S;
    for (int d = 0; d < nout; d++) {
        const int n = cnt[d];
        std::vector<long double> W(n);
        std::vector<std::pair<long double, int>> fr(n);
        long long tot = 0;
        wq[d].resize(n);
        for (int k = 0; k < n; k++) {
            W[k] = Wr[d][k] * scale;
            const long double f = floorl(W[k]);
            wq[d][k] = static_cast<long long>(f);
            fr[k] = {W[k] - f, k};
            tot += wq[d][k];
        }
        const long long rem = one - tot;
        if (rem < 0 || rem > n) return false;
        std::sort(fr.begin(), fr.end(), [](const auto &x, const auto &y) { return x.first > y.first; });
        for (long long i = 0; i < rem; i++) wq[d][fr[i].second] += 1;
        // X' - X = sum_k (Wq_k - W_k 2^S) R_k 2^-S with 0 <= R_k <= 255: between -255 (sum of the negative differences)
        // and +255 (sum of the positive ones)
        long double ep = 0, en = 0;
        for (int k = 0; k < n; k++) {
            if (wq[d][k] > 8355711 || wq[d][k] < -8355711) return false;
            const long double dlt = static_cast<long double>(wq[d][k]) - W[k];
            if (dlt > 0) ep += dlt; else en -= dlt;
        }
        emax = std::max(emax, 255.0L * std::max(ep, en));
    }
    // G: the fixed-point bound, the reference's own fp64 chain (< 1e-11 = 4e-5 units) and two units for this arithmetic
    const long long gq = static_cast<long long>(ceill(emax)) + 2;
    if (gq > (1 << 14)) return false;
    out->S = S;
    out->seed = static_cast<int>((192u << S) + (1u << (S - 1)) + static_cast<uint32_t>(gq));     // 128 (the bytes' offset) + 64 (the sums')
    out->thr = static_cast<int>(2 * gq);

    std::vector<RmEx> exv(nout);
    for (int d = 0; d < nout; d++) {
        RmEx &e = exv[d];
        e.n = cnt[d]; e.s0 = first[d]; e.inv = inv[d];
        for (int k = 0; k < 16; k++) e.aw[k] = k < cnt[d] ? 255.0 * t.wt[t.off[d] + k] : 0.0;
    }
    std::map<std::string, int> seen;
    std::vector<int8_t> mats;
    std::vector<int32_t> meta, sbase;
    auto intern = [&](const std::vector<int8_t> &m) {
        const std::string key(reinterpret_cast<const char *>(m.data()), m.size());
        auto itf = seen.find(key);
        if (itf != seen.end()) return itf->second;
        const int id = static_cast<int>(seen.size());
        seen.emplace(key, id);
        mats.insert(mats.end(), m.begin(), m.end());
        return id;
    };
    if (!vertical) {
        // groups of 16 outputs = one wave's columns; the window starts on a 16-px boundary and must end within 64 px
        const int ntx = (nout + 63) / 64, ng = 4 * ntx;
        std::vector<int> gfirst(ng, -1), wb(ng, -1), wend(ng, 0);
        for (int hg = 0; 16 * hg < nout; hg++) {
            int s_first = 1 << 30, s_end = 0;
            for (int d = 16 * hg; d < std::min(16 * hg + 16, nout); d++) {
                s_first = std::min(s_first, first[d]);
                s_end = std::max(s_end, first[d] + cnt[d]);
            }
            gfirst[hg] = s_first;
            wend[hg] = s_end;
        }
        // a strip stages 4 NC px from sbase on, never past the end of the row (the last strips start earlier instead); a
        // group's window starts a multiple of 16 px after sbase
        int NC = 0;
        std::vector<int> lo(ntx, 1 << 30), hi(ntx, 0);
        for (int tx = 0; tx < ntx; tx++) {
            for (int hg = 4 * tx; hg < 4 * tx + 4; hg++)
                if (gfirst[hg] >= 0) { lo[tx] = std::min(lo[tx], gfirst[hg]); hi[tx] = std::max(hi[tx], wend[hg]); }
            NC = std::max(NC, (hi[tx] - lo[tx] + 3) / 4);
        }
        if (NC > RM_NCMAX || 4 * NC > srcN) return false;
        sbase.assign(ntx, 0);
        for (int tx = 0; tx < ntx; tx++) {
            sbase[tx] = std::min(lo[tx], srcN - 4 * NC);
            for (int hg = 4 * tx; hg < 4 * tx + 4; hg++) {
                if (gfirst[hg] < 0) continue;
                wb[hg] = sbase[tx] + ((gfirst[hg] - sbase[tx]) & ~15);
                if (wend[hg] - wb[hg] > 64) return false;
            }
        }
        std::vector<int8_t> m(3 * 64 * 16);
        meta.assign(2 * static_cast<size_t>(ng), 0);
        const int zero_id = intern(std::vector<int8_t>(m.size(), 0));
        for (int hg = 0; hg < ng; hg++) {
            if (wb[hg] < 0) { meta[2 * hg] = 0; meta[2 * hg + 1] = zero_id; continue; }
            std::fill(m.begin(), m.end(), 0);
            for (int lane = 0; lane < 64; lane++) {
                const int nn = lane & 15, kc = lane >> 4, d = 16 * hg + nn;
                if (d >= nout) continue;
                for (int b = 0; b < 16; b++) {
                    const int tp = wb[hg] + 16 * kc + b - first[d];
                    if (tp < 0 || tp >= cnt[d]) continue;
                    int dg[3];
                    rm_digits(wq[d][tp], dg);
                    for (int l = 0; l < 3; l++) m[((2 - l) * 64 + lane) * 16 + b] = static_cast<int8_t>(dg[l]);
                }
            }
            meta[2 * hg] = wb[hg] - sbase[hg / 4];
            meta[2 * hg + 1] = intern(m);
        }
        out->KH = 1; out->NC = NC; out->ngroups = ng;
    } else {
        const int ng = (nout + 15) / 16;
        std::vector<int8_t> m(3 * 64 * 16);
        meta.assign(4 * static_cast<size_t>(ng), 0);
        int pvb = 0, pneed = 0;
        for (int vg = 0; vg < ng; vg++) {
            int s_first = 1 << 30, s_end = 0;
            for (int d = 16 * vg; d < std::min(16 * vg + 16, nout); d++) {
                s_first = std::min(s_first, first[d]);
                s_end = std::max(s_end, first[d] + cnt[d]);
            }
            const int vb = s_first >> 4, need = (s_end - 1) >> 4;
            if (need - vb > 3 || vb < pvb || need < pneed) return false;
            pvb = vb; pneed = need;
            std::fill(m.begin(), m.end(), 0);
            for (int lane = 0; lane < 64; lane++) {
                const int nn = lane & 15, kc = lane >> 4, d = 16 * vg + nn;
                if (d >= nout) continue;
                for (int b = 0; b < 16; b++) {
                    const int tp = 16 * vb + 16 * kc + b - first[d];
                    if (tp < 0 || tp >= cnt[d]) continue;
                    int dg[3];
                    rm_digits(wq[d][tp], dg);
                    for (int l = 0; l < 3; l++) m[((2 - l) * 64 + lane) * 16 + b] = static_cast<int8_t>(dg[l]);
                }
            }
            meta[4 * vg] = vb; meta[4 * vg + 1] = need; meta[4 * vg + 2] = intern(m);
        }
        out->ngroups = ng;
    }
    auto al16 = [](size_t n) { return (n + 15) & ~size_t(15); };
    const size_t b_m = al16(mats.size()), b_meta = al16(sizeof(int32_t) * meta.size()), b_sb = al16(sizeof(int32_t) * sbase.size());
    const size_t b_ex = sizeof(RmEx) * exv.size();
    std::vector<unsigned char> host(b_m + b_meta + b_sb + b_ex + 16, 0);
    std::memcpy(host.data() + b_m + b_meta + b_sb, exv.data(), b_ex);
    std::memcpy(host.data(), mats.data(), mats.size());
    std::memcpy(host.data() + b_m, meta.data(), sizeof(int32_t) * meta.size());
    if (!sbase.empty()) std::memcpy(host.data() + b_m + b_meta, sbase.data(), sizeof(int32_t) * sbase.size());
    if (hipMalloc(&out->blob, host.size()) != hipSuccess) { *out = RzMfTable(); return false; }
    if (hipMemcpy(out->blob, host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(out->blob);
        *out = RzMfTable();
        return false;
    }
    const unsigned char *base = static_cast<const unsigned char *>(out->blob);
    out->mats = base;
    out->meta = reinterpret_cast<const int32_t *>(base + b_m);
    out->sbase = reinterpret_cast<const int32_t *>(base + b_m + b_meta);
    out->ex = base + b_m + b_meta + b_sb;
    out->nmat = static_cast<int>(seen.size());
    out->ok = true;
    return true;
}

int resize_mfma_launch(fnx_ctx *ctx, const RzMfTable &h, const RzMfTable &v,
                       const uint8_t *src, int sstride, int srcW, int srcH, uint8_t *dst, int dstride, int dstW, int dstH,
                       uint32_t *todo, unsigned *gave_up, uint32_t gen, int old_tw, int old_th, int old_gx, int *workgroups,
                       int nimg, const uint8_t *const *d_srcs, uint8_t *const *d_dsts, uint32_t todo_stride)
{
    RzMfArgs a{};
    a.srcs = nimg > 1 ? d_srcs : nullptr; a.dsts = nimg > 1 ? d_dsts : nullptr; a.todo_stride = todo_stride;
    a.src = src; a.dst = dst; a.sstride = sstride; a.dstride = dstride;
    a.srcW = srcW; a.srcH = srcH; a.dstW = dstW; a.dstH = dstH;
    a.tiles_x = (dstW + 63) / 64;
    a.nvg = v.ngroups;
    a.NC = h.NC;
    a.SP = ((4 * h.NC + 4 + 31) / 64) * 64 + 32;                  // = 32 mod 64 (conflict-free A reads), a spare dword behind the chunks
    const int stage = 2 * 48 * a.SP + 256;                        // (+ what a window may read past the last staged row: zero weights)
    auto lds_of = [&](int segj) { return 4 * RM_WT + 2 * 16 * RM_OP + segj * 3072 + stage; };
    // output groups per workgroup: rounds of workgroups x iterations per workgroup, as blur_mfma_segment does it; the
    // occupancy follows from the LDS a workgroup of that many groups takes
    const double spg = static_cast<double>(srcH) / (16.0 * v.ngroups);       // source slots per output group
    int segj = 1;
    {
        double best = 0;
        for (int cand = 1; cand <= std::min(a.nvg, RM_MAXJ); cand++) {
            const int occ = std::max(1, std::min(3, (160 * 1024) / (lds_of(cand) + 1024)));
            const long slots = static_cast<long>(occ) * ctx->num_cus;
            const long wgs = static_cast<long>(a.tiles_x) * ((a.nvg + cand - 1) / cand) * std::max(1, nimg);
            const double iters = std::max(cand * spg, static_cast<double>(cand)) + 4.0;
            const double cost = static_cast<double>((wgs + slots - 1) / slots) * iters;
            if (cand == 1 || cost < best * 0.98) { best = cost; segj = cand; }
        }
    }
    if (const char *e = dev_env("FNX_RM_SEG")) segj = std::max(1, std::min(RM_MAXJ, atoi(e)));   // experiments
    a.segj = segj;
    a.tiles = a.tiles_x * ((a.nvg + segj - 1) / segj);
    const size_t lds = static_cast<size_t>(lds_of(segj));
    a.hmat = static_cast<const v4i *>(h.mats);
    a.hmeta = reinterpret_cast<const RmH *>(h.meta);
    a.sbase = h.sbase;
    a.vmat = static_cast<const v4i *>(v.mats);
    a.vmeta = reinterpret_cast<const RmV *>(v.meta);
    a.seed_h = h.seed; a.thr_h = h.thr; a.seed_v = v.seed; a.thr_v = v.thr;
    a.sh_h = h.S - 16; a.sh_v = v.S - 16;
    a.exh = static_cast<const RmEx *>(h.ex); a.exv = static_cast<const RmEx *>(v.ex);
#ifdef FNX_DEVELOP                     // a development build only (make DEVELOP=1): no fix-ups, results may be off by one
    { static const bool nofix = dev_env("FNX_RM_NOFIX") != nullptr; if (nofix) a.thr_h = a.thr_v = 0; }
#endif
    a.todo = todo; a.gave_up = gave_up; a.gen = gen; a.old_tw = old_tw; a.old_th = old_th; a.old_gx = old_gx;
    const dim3 grid(8 * ((a.tiles + 7) / 8), std::max(1, nimg));
    *workgroups = a.tiles * std::max(1, nimg);
    if (16 * a.NC <= 512) hipLaunchKernelGGL((resize_mfma_kernel<2>), grid, dim3(256), lds, ctx->stream, a);
    else hipLaunchKernelGGL((resize_mfma_kernel<3>), grid, dim3(256), lds, ctx->stream, a);
    FNX_HIP(hipGetLastError());
    return FNX_OK;
}

}  // namespace fnx
