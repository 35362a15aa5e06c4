// lanczosResize (resize.go:37-161) with both passes on the gfx950 i8 matrix pipe: opaque images, scale ratios up to ~2.3
// (the 4K -> 1080p downscale and its way back up are both inside).  Same idea as blur_mfma.hip, with matrices that change
// from output group to output group.
//
// What the reference computes for an output whose window is opaque (every A = 255; resize.go:93-113 / 137-156):
//     clampF(fl(r * inv)),  r = sum_k fl(R_k * aw_k),  aw_k = fl(255 w_k),  inv = fl(1 / sum_k aw_k)
// whose value is within 1e-12 of X = sum_k R_k W_k, W_k = aw_k inv (a real number).  Here W_k becomes the integer
// Wq_k ~ W_k 2^22, the roundings chosen so that sum_k Wq_k = 2^22 exactly (largest remainders), split into three signed
// base-256 digits; the bytes go in as R - 128 (R ^ 0x80 read as int8), so
//     u = sum_k Wq_k (R_k - 128) + 128 * 2^22 + 2^21 + G  =  X' 2^22 + 2^21 + G
// is an EXACT int32 with |X' - X| <= E = 255 sum_k |Wq_k 2^-22 - W_k| (the host computes it per output; G >= E + 2 units is
// the table's largest).  The output byte is sat_u8(u >> 22) -- floor(X + 1/2) clamped, clampF -- unless u's 22 fraction
// bits lie below 2 G: such a sample (one in ~500 on photographs) is recomputed in fp64 in the reference's own order and
// patched.  Proven, not sampled -- the rounding-guard argument of resize.hip with an integer sum in place of fp32 FMAs.
//
// A workgroup owns 64 output columns and a run of 16-row output groups; it marches down the SOURCE 16 rows ("slot") at a
// time.  H set of a slot: A = staged source bytes (M = row, K = 64 KH bytes of the group's window), B = the group's
// banded weight matrix (kept in registers for the whole march), C = 4 consecutive rows of one output byte per lane = one
// dword of the transposed uint8 intermediate T[byte column][row], a four-slot ring per wave.  V set of an output group,
// run as soon as its last slot is in the ring: A = T (M = byte column, K = the 64 rows of slots vb .. vb + 3), B = the
// group's weights (fetched one group ahead), C = one RGBA pixel of one output row per lane.  Loads and stores are
// workgroup-wide through LDS stages, as in the blur.
//
// What it does NOT handle it hands back: a workgroup that meets a pixel with A != 255, or a set dense with flagged
// samples (a linear ramp at an integer ratio puts every output on an exact tie), stops and marks the tiles of
// resize_fused_kernel (resize.hip) that cover its region; that kernel runs afterwards on marked tiles only and
// overwrites them in the reference's arithmetic.  Both are bit-exact, so the overlap is harmless.
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

constexpr int RM_S = 22;                 // fixed point of the weights
constexpr uint32_t RM_FRAC = (1u << RM_S) - 1;
constexpr int RM_P = 80;                 // ring: 64 rows + 16 per byte column (blur_mfma_wide_kernel's)
constexpr int RM_WT = 64 * RM_P + 256;
constexpr int RM_OP = 272;
constexpr int RM_SP1 = 352, RM_SP2 = 736;   // staged row pitch limits (KH = 1, 2); the pitch in use is = 32 mod 64: conflict-free A reads
constexpr int RM_NC1 = 20, RM_NC2 = 44;
constexpr int RM_MAXJ = 8;               // output groups per workgroup (their matrices live in LDS)
constexpr int RM_MAXTAPS = 16;
constexpr int RM_DENSE = 16;             // flagged lanes in one set from which the workgroup gives up

struct RmH { int hb, mat; };
struct RmV { int vb, need, mat, pad; };

struct RzMfArgs {
    const uint8_t *src;
    uint8_t *dst;
    int sstride, dstride, srcW, srcH, dstW, dstH;
    int tiles_x, tiles, segj, nvg;
    int NC, SP;
    const v4i *hmat;
    const RmH *hmeta;                    // per H group: window byte offset in the staged row, matrix index
    const int32_t *sbase;                // per strip: source byte offset of the staged row's first chunk
    const v4i *vmat;
    const RmV *vmeta;                    // per V group: first slot, last slot, matrix index
    int seed_h, thr_h, seed_v, thr_v;
    RzMfExact eh, ev;
    uint32_t *todo;
    uint32_t gen;
    int old_tw, old_th, old_gx;
};

__device__ __forceinline__ int rm_comb3(int hi, int mid, int lo)
{
    int t = hi * 256 + mid;
    asm volatile("" : "+v"(t));
    return t * 256 + lo;
}

// sat_u8(x >> 22) of two sums: their high halves side by side, an arithmetic shift of both, a saturating pack
__device__ __forceinline__ uint32_t rm_bytes2(int u1, int u0)
{
    const uint32_t hi = __builtin_amdgcn_perm(static_cast<uint32_t>(u1), static_cast<uint32_t>(u0), 0x07060302u);
    v2s h;
    __builtin_memcpy(&h, &hi, 4);
    h = h >> static_cast<short>(RM_S - 16);
    uint32_t hs, o;
    __builtin_memcpy(&hs, &h, 4);
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(o) : "v"(hs));
    return o;
}

// One flagged output channel in the reference's own arithmetic, opaque window (resize.go:95-112 / 139-155):
// aw = 255 w; r = r + R aw, taps ascending; clampF(r * inv).  The bytes are fetched (as R ^ 0x80) from p[((ring0 + t) & mask) * step].
__device__ __noinline__ uint32_t rm_exact(const uint8_t *p, int step, int ring0, int mask, int n, const double *wt, double inv)
{
    uint32_t v[RM_MAXTAPS];
    double w[RM_MAXTAPS];
#pragma unroll
    for (int t = 0; t < RM_MAXTAPS; t++) {
        const int tt = min(t, n - 1);
        v[t] = p[((ring0 + tt) & mask) * step];
        w[t] = wt[tt];
    }
    double r = 0;
#pragma unroll
    for (int t = 0; t < RM_MAXTAPS; t++)
        if (t < n) r = r + u8_to_f64(v[t] ^ 0x80u) * (255.0 * w[t]);
    return clampF_dev(r * inv);
}

#ifndef RM_OCC
#define RM_OCC 2
#endif
template <int KH>
__global__ __launch_bounds__(256, RM_OCC) void resize_mfma_kernel(RzMfArgs a)
{
    constexpr int NL = KH == 1 ? 2 : 3;
    constexpr int SPMAX = KH == 1 ? RM_SP1 : RM_SP2;
    constexpr int P = RM_P, WT = RM_WT, OP = RM_OP;
    __shared__ __attribute__((aligned(16))) uint8_t s_t[4 * WT];
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[2 * 16 * SPMAX];
    __shared__ __attribute__((aligned(16))) uint8_t s_out[2 * 16 * OP];
    __shared__ RmV s_vm[RM_MAXJ];
    __shared__ __attribute__((aligned(16))) v4i s_vmat[RM_MAXJ * 3 * 64];
    __shared__ int s_bad[2];

    const int tile = xcd_tile(blockIdx.x, a.tiles);
    if (tile < 0) return;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int x0 = 64 * tx;
    const int j0 = ty * a.segj, J = min(a.segj, a.nvg - j0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    if (tid < J) s_vm[tid] = a.vmeta[j0 + tid];
    if (tid < 2) s_bad[tid] = 0;
    const int SP = a.SP, NC = a.NC;
    const int sbyte0 = a.sbase[tx];
    const bool xedge = sbyte0 + 16 * NC > 4 * a.srcW;

    v4i bh[4][KH][3];
    int st_r[4];
#pragma unroll
    for (int qq = 0; qq < 4; qq++) {
        const RmH m = a.hmeta[16 * tx + 4 * wave + qq];
        st_r[qq] = r * SP + m.hb + 16 * g;
        const v4i *mp = a.hmat + static_cast<size_t>(m.mat) * (KH * 3 * 64) + lane;
#pragma unroll
        for (int kk = 0; kk < KH; kk++)
#pragma unroll
            for (int l = 0; l < 3; l++) bh[qq][kk][l] = mp[(3 * kk + l) * 64];
    }
    const bool alane = (r & 3) == 3;
    const v4i sh = {a.seed_h, a.seed_h, a.seed_h, a.seed_h}, sv = {a.seed_v, a.seed_v, a.seed_v, a.seed_v};
    const v4i zero = {0, 0, 0, 0};

    uint8_t *tw = s_t + wave * WT;
    int srow[NL], sch[NL], st_w[NL];
    bool valid[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) {
        // every lane loads (lanes past the staged rows' 16 NC chunks fetch the last chunk again and park it in the spare
        // 16 bytes behind row 0): loads under a branch make the compiler wait for ALL outstanding loads at every join
        const int id = min(tid + 256 * k, 16 * NC - 1);
        srow[k] = id / NC;
        sch[k] = id - NC * srow[k];
        valid[k] = tid + 256 * k < 16 * NC;
        st_w[k] = valid[k] ? srow[k] * SP + 16 * sch[k] : 16 * NC;
    }
    uint8_t *t_w = tw + r * P + 4 * g;                              // + (16 qq) P + 64 qq + 16 slot
    const int m4 = r >> 2, mi = r & 3;
    const uint8_t *t_r = tw + (16 * m4 + mi) * P + 64 * m4;         // + 4 q P + 16 ((rel + g) & 3)
    const int o_w = r * OP + 64 * wave + 16 * g;
    const int orow = tid >> 4, och = tid & 15;
    const int o_r = orow * OP + 16 * och;
    const int xo = x0 + 4 * och;

    __syncthreads();
    const int S0 = s_vm[0].vb, SL = s_vm[J - 1].need;                   // source slots S0 .. SL
    if (tid < 192)
        for (int j = 0; j < J; j++) s_vmat[j * 192 + tid] = a.vmat[static_cast<size_t>(s_vm[j].mat) * 192 + tid];
    int it = 0;                                                     // barriers passed

    // Both kinds of set run in two halves (two of the four 16-column groups each, a scheduling barrier between them): the
    // accumulators of one half are dead before the other's are born.
    auto hset = [&](int par, int slot) {
        const uint8_t *sbuf = s_stage + par * 16 * SPMAX;
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
            v4i c2[2], c1[2], c0[2];
#pragma unroll
            for (int q2 = 0; q2 < 2; q2++) { c2[q2] = zero; c1[q2] = zero; c0[q2] = sh; }
#pragma unroll
            for (int kk = 0; kk < KH; kk++)
#pragma unroll
                for (int q2 = 0; q2 < 2; q2++) {
                    const int qq = 2 * hf + q2;
                    const v4i A = *reinterpret_cast<const v4i *>(sbuf + st_r[qq] + 64 * kk);
                    c2[q2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bh[qq][kk][0], c2[q2], 0, 0, 0);
                    c1[q2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bh[qq][kk][1], c1[q2], 0, 0, 0);
                    c0[q2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bh[qq][kk][2], c0[q2], 0, 0, 0);
                }
            v4i u[2];
#pragma unroll
            for (int q2 = 0; q2 < 2; q2++)
#pragma unroll
                for (int k = 0; k < 4; k++) u[q2][k] = rm_comb3(c2[q2][k], c1[q2][k], c0[q2][k]);
            uint32_t m[2];
#pragma unroll
            for (int q2 = 0; q2 < 2; q2++) {
                const int qq = 2 * hf + q2;
                const uint32_t b01 = rm_bytes2(u[q2][1], u[q2][0]), b23 = rm_bytes2(u[q2][3], u[q2][2]);
                const uint32_t wv = __builtin_amdgcn_perm(b23, b01, 0x05040100u) ^ 0x80808080u;
                *reinterpret_cast<uint32_t *>(t_w + (16 * qq) * P + 64 * qq + 16 * slot) = alane ? 0x7f7f7f7fu : wv;
                const uint32_t f0 = static_cast<uint32_t>(u[q2][0]) & RM_FRAC, f1 = static_cast<uint32_t>(u[q2][1]) & RM_FRAC;
                const uint32_t f2 = static_cast<uint32_t>(u[q2][2]) & RM_FRAC, f3 = static_cast<uint32_t>(u[q2][3]) & RM_FRAC;
                m[q2] = min(min(min(f0, f1), f2), f3);
            }
#ifdef RM_NOGUARD
            const unsigned long long bal = 0;
#else
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(min(m[0], m[1]) < static_cast<uint32_t>(a.thr_h));
#endif
            if (bal) {
                if (__builtin_popcountll(bal) > RM_DENSE) {
                    s_bad[(it + 1) & 1] = 1;
                } else {
                    uint32_t fl = 0;
#pragma unroll
                    for (int q2 = 0; q2 < 2; q2++)
#pragma unroll
                        for (int k = 0; k < 4; k++) fl |= ((static_cast<uint32_t>(u[q2][k]) & RM_FRAC) < static_cast<uint32_t>(a.thr_h) ? 1u : 0u) << (4 * q2 + k);
                    while (fl) {
                        const int b = __builtin_ctz(fl), qq = 2 * hf + (b >> 2), k = b & 3;
                        fl &= fl - 1;
                        const int dx = x0 + 16 * wave + 4 * qq + (r >> 2);
                        const int t0 = a.eh.off[dx], n = a.eh.off[dx + 1] - t0, s0 = a.eh.idx[t0];
                        const uint8_t *p = sbuf + (4 * g + k) * SP + (4 * s0 - sbyte0) + (r & 3);
                        const uint32_t e = rm_exact(p, 4, 0, 0xffff, n, a.eh.wt + t0, a.eh.inv[dx]) ^ 0x80u;
                        *(t_w + (16 * qq) * P + 64 * qq + 16 * slot + k) = static_cast<uint8_t>(e);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto vset = [&](int jl, int ob) {
        const int ro = 16 * ((s_vm[jl].vb - S0 + g) & 3);
        uint8_t *op = s_out + ob * 16 * OP + o_w;
        v4i A[4];
#pragma unroll
        for (int q = 0; q < 4; q++) A[q] = *reinterpret_cast<const v4i *>(t_r + (4 * q) * P + ro);
        const v4i b2 = s_vmat[jl * 192 + lane], b1 = s_vmat[jl * 192 + 64 + lane], b0 = s_vmat[jl * 192 + 128 + lane];
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
            v4i c2[2], c1[2], c0[2];
#pragma unroll
            for (int q2 = 0; q2 < 2; q2++) {
                c2[q2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[2 * hf + q2], b2, zero, 0, 0, 0);
                c1[q2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[2 * hf + q2], b1, zero, 0, 0, 0);
                c0[q2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[2 * hf + q2], b0, sv, 0, 0, 0);
            }
            int u[2][3];
#pragma unroll
            for (int q2 = 0; q2 < 2; q2++)
#pragma unroll
                for (int i = 0; i < 3; i++) u[q2][i] = rm_comb3(c2[q2][i], c1[q2][i], c0[q2][i]);
            u32x2 o;
            uint32_t m[2];
#pragma unroll
            for (int q2 = 0; q2 < 2; q2++) {
                const uint32_t b01 = rm_bytes2(u[q2][1], u[q2][0]), bb = rm_bytes2(u[q2][2], u[q2][2]);
                o[q2] = __builtin_amdgcn_perm(bb, b01, 0x0d040100u);
                m[q2] = min(min(static_cast<uint32_t>(u[q2][0]) & RM_FRAC, static_cast<uint32_t>(u[q2][1]) & RM_FRAC), static_cast<uint32_t>(u[q2][2]) & RM_FRAC);
            }
            *reinterpret_cast<u32x2 *>(op + 8 * hf) = o;
#ifdef RM_NOGUARD
            const unsigned long long bal = 0;
#else
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(min(m[0], m[1]) < static_cast<uint32_t>(a.thr_v));
#endif
            if (bal) {
                if (__builtin_popcountll(bal) > RM_DENSE) {
                    s_bad[(it + 1) & 1] = 1;
                } else {
                    uint32_t fl = 0;
#pragma unroll
                    for (int q2 = 0; q2 < 2; q2++)
#pragma unroll
                        for (int i = 0; i < 3; i++) fl |= ((static_cast<uint32_t>(u[q2][i]) & RM_FRAC) < static_cast<uint32_t>(a.thr_v) ? 1u : 0u) << (4 * q2 + i);
                    while (fl) {
                        const int b = __builtin_ctz(fl), q = 2 * hf + (b >> 2), i = b & 3;
                        fl &= fl - 1;
                        const int dy = 16 * (j0 + jl) + r;
                        const int t0 = a.ev.off[dy], n = a.ev.off[dy + 1] - t0, s0 = a.ev.idx[t0];
                        const uint8_t *p = tw + (16 * g + 4 * q + i) * P + 64 * g;
                        op[4 * q + i] = static_cast<uint8_t>(rm_exact(p, 1, (s0 - 16 * S0) & 63, 63, n, a.ev.wt + t0, a.ev.inv[dy]));
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
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

    auto march = [&](auto xedget) -> bool {
        constexpr bool XEDGE = decltype(xedget)::value;
        auto hload = [&](int sabs, u32x4 (&d)[NL]) {
            const int ys = 16 * sabs;
#pragma unroll
            for (int k = 0; k < NL; k++) {
                const int y = min(ys + srow[k], a.srcH - 1);
                const uint8_t *rowp = a.src + static_cast<size_t>(y) * a.sstride;
                if constexpr (!XEDGE) {
                    d[k] = *(g_u32x4 *)(rowp + sbyte0 + 16 * sch[k]);
                } else {
                    const int xc = (sbyte0 >> 2) + 4 * sch[k];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const uint32_t px = ld_px(rowp, min(xc + e, a.srcW - 1));
                        d[k][e] = xc + e < a.srcW ? px : 0xff000000u;
                    }
                }
            }
        };
        auto stage_write = [&](const u32x4 (&d)[NL], int par) {
            uint8_t *sb = s_stage + par * 16 * SPMAX;
            uint32_t m = 0xff000000u;
#pragma unroll
            for (int k = 0; k < NL; k++) {
                m &= d[k][0] & d[k][1] & d[k][2] & d[k][3];
                *reinterpret_cast<u32x4 *>(sb + st_w[k]) = d[k] ^ 0x80808080u;
            }
            if (m != 0xff000000u) s_bad[it & 1] = 1;               // a pixel that is not opaque: read after the coming barrier
        };
        int s = S0, jl = 0, pend = -1, pob = 0, ob = 0;
        // after barrier `it`: s_bad[it & 1] holds what was raised before it; what the sets raise now goes to the other cell
        auto post = [&]() -> bool {
            if (s_bad[it & 1]) return false;
            if (pend >= 0) { out_store(pend, pob); pend = -1; }
            return true;
        };
        auto vready = [&]() { return jl < J && s_vm[jl].need < s; };
        auto vrun = [&]() { vset(jl, ob); pend = jl; pob = ob; ob ^= 1; jl++; };
        auto hstep = [&](u32x4 (&d)[NL], auto part) -> bool {
            constexpr int par = decltype(part)::value;
            stage_write(d, par);
            hload(min(s + 2, SL), d);
            __syncthreads();
            if (!post()) return false;
            hset(par, (s - S0) & 3);
            s++;
            if (vready()) vrun();
            it++;
            return true;
        };
        auto drain = [&]() -> bool {
            while (vready()) {
                __syncthreads();
                if (!post()) return false;
                vrun();
                it++;
            }
            return true;
        };
        u32x4 ra[NL], rb[NL];
        hload(S0, ra);
        __builtin_amdgcn_sched_barrier(0);                          // ra's loads are issued first: the loop's vmcnt waits count on it
        hload(min(S0 + 1, SL), rb);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        while (s <= SL) {
            if (!hstep(ra, std::integral_constant<int, 0>{})) return false;
            if (!drain()) return false;
            if (s > SL) break;
            if (!hstep(rb, std::integral_constant<int, 1>{})) return false;
            if (!drain()) return false;
        }
        __syncthreads();
        return post();
    };
    const bool done = xedge ? march(std::true_type{}) : march(std::false_type{});
    if (!done) {
        // hand the region back: resize_fused_kernel's tiles (old_tw x old_th output px) over columns x0 .. x0 + 63, this run of rows
        const int y_first = 16 * j0, y_last = min(16 * (j0 + J), a.dstH) - 1;
        const int by0 = y_first / a.old_th, by1 = y_last / a.old_th, bx = x0 / a.old_tw;
        for (int b = by0 + tid; b <= by1; b += 256) a.todo[b * a.old_gx + bx] = a.gen;
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

static bool resize_mfma_enabled()
{
    static const bool off = [] { const char *e = getenv("FNX_RESIZE_MFMA"); return e && e[0] == '0'; }();
    return !off;
}

void resize_mfma_free(RzMfTable *t)
{
    if (t->blob) (void)hipFree(t->blob);
    *t = RzMfTable();
}

// The matrix form of one tap table; false (nothing allocated): outside what the kernel covers.
// `inv`: 1 / a per output as the guard form computed it (resize.hip: build_guard, a = sum of 255 w in tap order).
bool resize_mfma_build(const TapTable &t, int srcN, bool vertical, const double *inv, RzMfTable *out)
{
    *out = RzMfTable();
    if (!resize_mfma_enabled()) return false;
    const int nout = t.nout;
    if (nout < 16 || srcN < 16) return false;
    // fixed-point weights per output
    std::vector<std::vector<long long>> wq(nout);
    std::vector<int> first(nout), cnt(nout);
    long double emax = 0;
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
        std::vector<long double> W(n);
        std::vector<std::pair<long double, int>> fr(n);
        long long tot = 0;
        wq[d].resize(n);
        for (int k = 0; k < n; k++) {
            W[k] = static_cast<long double>(255.0 * t.wt[t0 + k]) * static_cast<long double>(inv[d]) * 4194304.0L;
            const long double f = floorl(W[k]);
            wq[d][k] = static_cast<long long>(f);
            fr[k] = {W[k] - f, k};
            tot += wq[d][k];
        }
        const long long rem = 4194304 - tot;
        if (rem < 0 || rem > n) return false;
        std::sort(fr.begin(), fr.end(), [](const auto &x, const auto &y) { return x.first > y.first; });
        for (long long i = 0; i < rem; i++) wq[d][fr[i].second] += 1;
        long double e = 0;
        for (int k = 0; k < n; k++) {
            if (wq[d][k] > 8355711 || wq[d][k] < -8355711) return false;
            e += fabsl(static_cast<long double>(wq[d][k]) - W[k]);
        }
        emax = std::max(emax, 255.0L * e);
    }
    // G: the fixed-point bound, the reference's own fp64 chain (< 1e-11 = 4e-5 units) and two units for this arithmetic
    const long long gq = static_cast<long long>(ceill(emax)) + 2;
    if (gq > (1 << 14)) return false;
    out->seed = static_cast<int>((128u << RM_S) + (1u << (RM_S - 1)) + static_cast<uint32_t>(gq));
    out->thr = static_cast<int>(2 * gq);

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
        const int ntx = (nout + 63) / 64, ng = 16 * ntx;
        std::vector<int> wb(ng, -1);
        int KH = 1;
        for (int hg = 0; 4 * hg < nout; hg++) {
            int s_first = 1 << 30, s_end = 0;
            for (int d = 4 * hg; d < std::min(4 * hg + 4, nout); d++) {
                s_first = std::min(s_first, first[d]);
                s_end = std::max(s_end, first[d] + cnt[d]);
            }
            wb[hg] = (4 * s_first) & ~15;
            const int span = 4 * s_end - wb[hg];
            if (span > 128) return false;
            if (span > 64) KH = 2;
        }
        int NC = 0;
        sbase.assign(ntx, 0);
        for (int tx = 0; tx < ntx; tx++) {
            int lo = 1 << 30, hi = 0;
            for (int hg = 16 * tx; hg < 16 * tx + 16; hg++)
                if (wb[hg] >= 0) { lo = std::min(lo, wb[hg]); hi = std::max(hi, wb[hg]); }
            sbase[tx] = lo;
            NC = std::max(NC, (hi - lo) / 16);                          // + the window's own chunks below
        }
        if (NC + 4 > RM_NC1) KH = 2;
        NC += 4 * KH;
        if (NC > (KH == 1 ? RM_NC1 : RM_NC2)) return false;
        std::vector<int8_t> m(static_cast<size_t>(KH) * 3 * 64 * 16);
        meta.assign(2 * static_cast<size_t>(ng), 0);
        const int zero_id = intern(std::vector<int8_t>(m.size(), 0));
        for (int hg = 0; hg < ng; hg++) {
            if (wb[hg] < 0) { meta[2 * hg] = 0; meta[2 * hg + 1] = zero_id; continue; }
            std::fill(m.begin(), m.end(), 0);
            for (int lane = 0; lane < 64; lane++) {
                const int nn = lane & 15, kc = lane >> 4, pj = nn / 4, c = nn % 4, d = 4 * hg + pj;
                if (c == 3 || d >= nout) continue;
                for (int kk = 0; kk < KH; kk++)
                    for (int b = 0; b < 16; b++) {
                        const int byte = wb[hg] + 64 * kk + 16 * kc + b, px = byte / 4, ch = byte % 4;
                        const int tp = px - first[d];
                        if (ch != c || tp < 0 || tp >= cnt[d]) continue;
                        int dg[3];
                        rm_digits(wq[d][tp], dg);
                        for (int l = 0; l < 3; l++) m[((3 * kk + (2 - l)) * 64 + lane) * 16 + b] = static_cast<int8_t>(dg[l]);
                    }
            }
            meta[2 * hg] = wb[hg] - sbase[hg / 16];
            meta[2 * hg + 1] = intern(m);
        }
        out->KH = KH; out->NC = NC; out->ngroups = ng;
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
    std::vector<unsigned char> host(b_m + b_meta + b_sb + 16, 0);
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
    out->nmat = static_cast<int>(seen.size());
    out->ok = true;
    return true;
}

// Output groups per workgroup: rounds of workgroups x iterations per workgroup, as blur_mfma_segment does it
static int resize_mfma_segment(const fnx_ctx *ctx, int tiles_x, int nvg, double slots_per_group, int occ)
{
    const long slots = static_cast<long>(occ) * ctx->num_cus;
    int best_j = 1, first = 1;
    double best = 0;
    for (int segj = 1; segj <= std::min(nvg, RM_MAXJ); segj++) {
        const long wgs = static_cast<long>(tiles_x) * ((nvg + segj - 1) / segj);
        const double iters = std::max(segj * slots_per_group, static_cast<double>(segj)) + 4.0;
        const double cost = static_cast<double>((wgs + slots - 1) / slots) * iters;
        if (first || cost < best * 0.98 || (cost <= best * 1.02 && segj > best_j)) {
            if (first || cost < best) best = cost;
            best_j = segj;
            first = 0;
        }
    }
    return best_j;
}

int resize_mfma_launch(fnx_ctx *ctx, const RzMfTable &h, const RzMfTable &v, const RzMfExact &eh, const RzMfExact &ev,
                       const uint8_t *src, int sstride, int srcW, int srcH, uint8_t *dst, int dstride, int dstW, int dstH,
                       uint32_t *todo, uint32_t gen, int old_tw, int old_th, int old_gx)
{
    RzMfArgs a{};
    a.src = src; a.dst = dst; a.sstride = sstride; a.dstride = dstride;
    a.srcW = srcW; a.srcH = srcH; a.dstW = dstW; a.dstH = dstH;
    a.tiles_x = (dstW + 63) / 64;
    a.nvg = v.ngroups;
    a.NC = h.NC;
    a.SP = ((16 * h.NC + 16 + 31) / 64) * 64 + 32;              // = 32 mod 64, and 16 spare bytes behind the chunks
    const int occ = h.KH == 1 ? 3 : 2;
    int segj = resize_mfma_segment(ctx, a.tiles_x, a.nvg, static_cast<double>(srcH) / (16.0 * v.ngroups), occ);
    if (const char *e = getenv("FNX_RM_SEG")) segj = std::max(1, std::min(RM_MAXJ, atoi(e)));   // experiments
    a.segj = segj;
    a.tiles = a.tiles_x * ((a.nvg + segj - 1) / segj);
    a.hmat = static_cast<const v4i *>(h.mats);
    a.hmeta = reinterpret_cast<const RmH *>(h.meta);
    a.sbase = h.sbase;
    a.vmat = static_cast<const v4i *>(v.mats);
    a.vmeta = reinterpret_cast<const RmV *>(v.meta);
    a.seed_h = h.seed; a.thr_h = h.thr; a.seed_v = v.seed; a.thr_v = v.thr;
    a.eh = eh; a.ev = ev;
    if (getenv("FNX_RM_NOFIX")) a.thr_h = a.thr_v = 0;                 // experiments: no fix-ups (results may be off by one)
    a.todo = todo; a.gen = gen; a.old_tw = old_tw; a.old_th = old_th; a.old_gx = old_gx;
    const dim3 grid(8 * ((a.tiles + 7) / 8));
    if (h.KH == 1) hipLaunchKernelGGL((resize_mfma_kernel<1>), grid, dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL((resize_mfma_kernel<2>), grid, dim3(256), 0, ctx->stream, a);
    FNX_HIP(hipGetLastError());
    return FNX_OK;
}

}  // namespace fnx
