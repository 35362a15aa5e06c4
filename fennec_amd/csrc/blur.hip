// GaussianBlur (effects.go:146-220) on gfx950.
//
//  * blur_pass_kernel<T,VERT>: one separable pass, any radius, thread per pixel.
//    T=double is the EXACT mode: fp64, unfused mul+add in the reference's tap order
//    (this TU is built with -ffp-contract=off), so H(tmp uint8) then V is bit-exact.
//    T=float is the generic fast mode for radii the fused kernel is not built for.
//  * blur_fused_kernel<R>: the fast path for small radii (sigma=2 -> R=6): one launch,
//    NRGBA tile + halo staged in LDS with 16-byte coalesced loads, horizontal pass from
//    LDS into a uint8 LDS intermediate (the reference rounds the intermediate to uint8,
//    effects.go:186-188), vertical pass from LDS, 16-byte stores.  HBM traffic is
//    read-once/write-once (2*S); fp32 FMA accumulation (<=1 LSB off on <=0.1% samples).
#include "common.hpp"
#include "devutil.hpp"

#include <cstdlib>

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
    return acc + static_cast<double>(v) * wt;   // r += float64(pix) * wt  (effects.go:181)
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
constexpr int FUSED_RMAX = 8;

struct FusedArgs {
    const uint8_t *src;
    const uint8_t *const *srcs;
    uint8_t *dst;
    uint8_t *const *dsts;
    int sstride, dstride, w, h;
    int tiles_x, tiles;   // per image
    int vec_in, vec_out;  // pointers+strides 16-byte aligned
    float wt[2 * FUSED_RMAX + 1];
};

// acc += f * w on two lanes at once (v_pk_fma_f32)
__device__ __forceinline__ v2f fma2(v2f f, float w, v2f acc)
{
    return __builtin_elementwise_fma(f, (v2f){w, w}, acc);
}
// clampF of a fast-mode accumulator in ONE instruction.  v_cvt_pk_u8_f32 saturates to [0,255] and
// rounds per the wave's FP32 round mode (probed on gfx950: nearest-even by default, truncation
// under round-toward-zero).  Accumulators are seeded with 0.5, so truncation is floor(sum + 0.5)
// = clampF's round-half-up; the mode is flipped (one SALU s_setreg each way) only around the
// packing instructions, the FMAs all run in round-to-nearest-even.
__device__ __forceinline__ void fp32_round_toward_zero() { __builtin_amdgcn_s_setreg(0x801, 3); }   // hwreg(MODE, 0, 2)
__device__ __forceinline__ void fp32_round_nearest() { __builtin_amdgcn_s_setreg(0x801, 0); }
__device__ __forceinline__ uint32_t pk8(float x, uint32_t sel, uint32_t old)
{
    return __builtin_amdgcn_cvt_pk_u8_f32(x, sel, old);
}

// Work decomposition of one TW x TH output tile (256 threads):
//   stage : (TH+2R) rows x (TW+2RA) px of NRGBA, 16-byte global loads, into LDS s_in
//   H pass: item = 2 staged rows x 8 output px.  The 2 rows x RGB = 6 accumulators per output
//           are three float2 lanes, so every tap is 3 v_pk_fma_f32 for 2 pixels; input pixels
//           are converted once and scattered into the (<=4) outputs they feed, taps ascending.
//           Results are rounded to uint8 into LDS s_tmp (the reference's uint8 intermediate).
//   V pass: item = 4 columns x Q output rows; one 16-byte LDS read per staged row, 4 px x RGB =
//           6 float2 lanes, scattered into the (<=Q) output rows it feeds, taps ascending.
template <int R, int TW, int TH, int Q, int IWP>
__global__ __launch_bounds__(256, 4) void blur_fused_kernel(FusedArgs a)
{
    constexpr int RA = (R + 3) & ~3;        // halo rounded up to whole 16-byte chunks
    constexpr int D = RA - R;               // px the LDS image is shifted against global chunks
    constexpr int NT = 2 * R + 1;
    constexpr int IH = TH + 2 * R;          // staged rows
    constexpr int HO = 8;                   // outputs per H item and row: each px converted (HO+2R)/HO times
    constexpr int NPX = HO + 2 * R;         // input px one H item needs per row
    constexpr int NV = (NPX + 3) / 4;       // ... as 16-byte LDS reads
    constexpr int GC = (TW + 2 * RA) / 4;   // global 16-byte chunks per staged row
    constexpr int RPP = 256 / GC;           // staged rows per pass of the 256 lanes
    constexpr int NLOAD = (IH + RPP - 1) / RPP;
    constexpr int GROUPS = TW / 4, HGROUPS = TW / HO;
    static_assert(TH % Q == 0 && TW % HO == 0 && IH % 2 == 0, "tile shape");
    static_assert(IWP % 4 == 0 && IWP >= TW - HO + 4 * NV, "LDS pitch");

    // s_in[r][i]  <-> src(x0 - R + i, clamp(y0 - R + r));  s_tmp[r][x] <-> H-pass of that row.
    __shared__ __attribute__((aligned(16))) uint32_t smem[IH * IWP + IH * TW];
    uint32_t *s_in = smem;
    uint32_t *s_tmp = smem + IH * IWP;

    const int tile = xcd_tile(blockIdx.x, a.tiles);
    if (tile < 0) return;
    const int z = blockIdx.y;
    const uint8_t *src = a.srcs ? a.srcs[z] : a.src;
    uint8_t *dst = a.dsts ? a.dsts[z] : a.dst;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int x0 = tx * TW, y0 = ty * TH;
    const int tid = threadIdx.x;

    // ---- stage the tile + halo: all global loads first (memory-level parallelism) ----
    // lane -> (chunk column gc, row lane r0), rows r0, r0+RPP, ...: one division per lane, every
    // further address is a constant stride away (the staging used to cost 18 VALU ops per output
    // pixel in index arithmetic -- on a VALU-bound kernel that was 15 % of the run time)
    const int r0 = tid / GC, gc = tid - r0 * GC;
    const bool stager = r0 < RPP;
    const int xs = x0 - RA + 4 * gc;                     // first px of this lane's chunks
    const bool interior = a.vec_in && x0 - RA >= 0 && x0 + TW + RA <= a.w && y0 - R >= 0 && y0 + TH + R <= a.h;
    u32x4 v[NLOAD];
    if (interior) {                                      // workgroup-uniform: no clamps, no per-load tests
        const uint8_t *p = src + static_cast<size_t>(y0 - R + r0) * a.sstride + 4 * static_cast<size_t>(xs);
#pragma unroll
        for (int it = 0; it < NLOAD; it++)
            if (stager && r0 + it * RPP < IH) v[it] = *(g_u32x4 *)(p + static_cast<size_t>(it * RPP) * a.sstride);
    } else {
#pragma unroll
        for (int it = 0; it < NLOAD; it++) {
            const int r = r0 + it * RPP;
            if (stager && r < IH) {
                const uint8_t *row = src + static_cast<size_t>(clampi(y0 - R + r, 0, a.h - 1)) * a.sstride;
                if (a.vec_in && xs >= 0 && xs + 3 < a.w) {
                    v[it] = *(g_u32x4 *)(row + 4 * static_cast<size_t>(xs));
                } else {   // image border (clamp-to-edge, effects.go:174-178) or unaligned input
                    v[it].x = ld_px(row, clampi(xs, 0, a.w - 1));
                    v[it].y = ld_px(row, clampi(xs + 1, 0, a.w - 1));
                    v[it].z = ld_px(row, clampi(xs + 2, 0, a.w - 1));
                    v[it].w = ld_px(row, clampi(xs + 3, 0, a.w - 1));
                }
            }
        }
    }
    {
        const int i0 = 4 * gc - D;                       // LDS column of the chunk's first px
        uint32_t *colp = s_in + r0 * IWP + i0;
#pragma unroll
        for (int it = 0; it < NLOAD; it++) {
            if (stager && r0 + it * RPP < IH) {
                uint32_t *rowp = colp + it * RPP * IWP;
                if constexpr (D == 0) {
                    if (i0 + 3 < IWP) *reinterpret_cast<u32x4 *>(rowp) = v[it];
                } else if constexpr (D == 2) {
                    if (i0 >= 0 && i0 + 1 < IWP) *reinterpret_cast<u32x2 *>(rowp) = (u32x2){v[it].x, v[it].y};
                    if (i0 + 3 < IWP) *reinterpret_cast<u32x2 *>(rowp + 2) = (u32x2){v[it].z, v[it].w};
                } else {
                    if (i0 >= 0 && i0 < IWP) rowp[0] = v[it].x;
                    if (i0 + 1 >= 0 && i0 + 1 < IWP) rowp[1] = v[it].y;
                    if (i0 + 2 >= 0 && i0 + 2 < IWP) rowp[2] = v[it].z;
                    if (i0 + 3 >= 0 && i0 + 3 < IWP) rowp[3] = v[it].w;
                }
            }
        }
    }
    __syncthreads();

    // ---- horizontal pass (effects.go:169-191): item = 2 staged rows x HO outputs ----
    for (int item = tid; item < (IH / 2) * HGROUPS; item += 256) {
        const int rp = item / HGROUPS, g = item - rp * HGROUPS;
        const uint32_t *row0 = s_in + (2 * rp) * IWP + HO * g;
        const uint32_t *row1 = row0 + IWP;
        v2f acc[HO][3];
#pragma unroll
        for (int j = 0; j < HO; j++) acc[j][0] = acc[j][1] = acc[j][2] = (v2f){0.5f, 0.5f};
#pragma unroll
        for (int q = 0; q < NV; q++) {
            const u32x4 t0 = *reinterpret_cast<const u32x4 *>(row0 + 4 * q);
            const u32x4 t1 = *reinterpret_cast<const u32x4 *>(row1 + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int i = 4 * q + e;
                if (i < NPX) {
                    const uint32_t p0 = t0[e], p1 = t1[e];
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
                    // keep the converts next to their FMAs: hoisting all of them costs 6 VGPRs per px
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        uint32_t o0[HO], o1[HO];
        fp32_round_toward_zero();
#pragma unroll
        for (int j = 0; j < HO; j++) {
            o0[j] = pk8(acc[j][1].x, 2, pk8(acc[j][0].y, 1, pk8(acc[j][0].x, 0, 0)));
            o1[j] = pk8(acc[j][2].y, 2, pk8(acc[j][2].x, 1, pk8(acc[j][1].y, 0, 0)));
        }
        fp32_round_nearest();
#pragma unroll
        for (int b = 0; b < HO / 4; b++) {
            *reinterpret_cast<u32x4 *>(s_tmp + (2 * rp) * TW + HO * g + 4 * b) = (u32x4){o0[4 * b], o0[4 * b + 1], o0[4 * b + 2], o0[4 * b + 3]};
            *reinterpret_cast<u32x4 *>(s_tmp + (2 * rp + 1) * TW + HO * g + 4 * b) = (u32x4){o1[4 * b], o1[4 * b + 1], o1[4 * b + 2], o1[4 * b + 3]};
        }
    }
    __syncthreads();

    // ---- vertical pass (effects.go:195-217) ----
    for (int item = tid; item < GROUPS * (TH / Q); item += 256) {
        const int q = item / GROUPS, g = item - q * GROUPS;
        const int x = x0 + 4 * g;
        if (x >= a.w) continue;   // decided BEFORE the arithmetic so the compiler keeps it in one block
        v2f acc[Q][6];
#pragma unroll
        for (int j = 0; j < Q; j++)
#pragma unroll
            for (int e = 0; e < 6; e++) acc[j][e] = (v2f){0.5f, 0.5f};
        const uint32_t *colp = s_tmp + (q * Q) * TW + 4 * g;
        u32x4 tn = *reinterpret_cast<const u32x4 *>(colp);
#pragma unroll
        for (int i = 0; i < Q + 2 * R; i++) {
            const u32x4 t = tn;
            if (i + 1 < Q + 2 * R) tn = *reinterpret_cast<const u32x4 *>(colp + (i + 1) * TW);   // prefetch next row
            v2f f[6];
            f[0] = (v2f){static_cast<float>(t.x & 0xffu), static_cast<float>((t.x >> 8) & 0xffu)};
            f[1] = (v2f){static_cast<float>((t.x >> 16) & 0xffu), static_cast<float>(t.y & 0xffu)};
            f[2] = (v2f){static_cast<float>((t.y >> 8) & 0xffu), static_cast<float>((t.y >> 16) & 0xffu)};
            f[3] = (v2f){static_cast<float>(t.z & 0xffu), static_cast<float>((t.z >> 8) & 0xffu)};
            f[4] = (v2f){static_cast<float>((t.z >> 16) & 0xffu), static_cast<float>(t.w & 0xffu)};
            f[5] = (v2f){static_cast<float>((t.w >> 8) & 0xffu), static_cast<float>((t.w >> 16) & 0xffu)};
#pragma unroll
            for (int j = 0; j < Q; j++) {
                const int k = i - j;
                if (k >= 0 && k < NT) {
#pragma unroll
                    for (int e = 0; e < 6; e++) acc[j][e] = fma2(f[e], a.wt[k], acc[j][e]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // round + pack every output row first, pinned with an empty asm: otherwise LLVM sinks the
        // whole accumulation into the `y < h` store branches and the live ranges explode
        u32x4 o[Q];
        fp32_round_toward_zero();
#pragma unroll
        for (int j = 0; j < Q; j++) {
            // alpha from the ORIGINAL image (effects.go:215), still in the staged tile: the source
            // pixel seeds the pack chain, bytes 0..2 are overwritten, byte 3 (alpha) survives
            const uint32_t *ap = s_in + (q * Q + j + R) * IWP + 4 * g + R;
            o[j].x = pk8(acc[j][1].x, 2, pk8(acc[j][0].y, 1, pk8(acc[j][0].x, 0, ap[0])));
            o[j].y = pk8(acc[j][2].y, 2, pk8(acc[j][2].x, 1, pk8(acc[j][1].y, 0, ap[1])));
            o[j].z = pk8(acc[j][4].x, 2, pk8(acc[j][3].y, 1, pk8(acc[j][3].x, 0, ap[2])));
            o[j].w = pk8(acc[j][5].y, 2, pk8(acc[j][5].x, 1, pk8(acc[j][4].y, 0, ap[3])));
            asm volatile("" : "+v"(o[j].x), "+v"(o[j].y), "+v"(o[j].z), "+v"(o[j].w));
        }
        fp32_round_nearest();
#pragma unroll
        for (int j = 0; j < Q; j++) {
            const int y = y0 + q * Q + j;
            if (y < a.h) {
                uint8_t *drow = dst + static_cast<size_t>(y) * a.dstride + 4 * static_cast<size_t>(x);
                if (a.vec_out && x + 3 < a.w) {
                    *(g_u32x4w *)(drow) = o[j];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (x + e < a.w) *(g_u32w *)(drow + 4 * e) = o[j][e];
                }
            }
        }
    }
}

template <int R>
static int launch_fused(fnx_ctx *ctx, int n, FusedArgs &fa)
{
    constexpr int TW = 64, TH = ((64 - 2 * R) / 4) * 4, Q = 4;   // TH + 2R <= 64 staged rows
    constexpr int NV = (8 + 2 * R + 3) / 4;
    constexpr int IWP = ((TW - 8 + 4 * NV) + 31) / 32 * 32;   // pitch = 0 mod 128 B
    fa.tiles_x = (fa.w + TW - 1) / TW;
    fa.tiles = fa.tiles_x * ((fa.h + TH - 1) / TH);
    dim3 grid(8 * ((fa.tiles + 7) / 8), n);
    hipLaunchKernelGGL((blur_fused_kernel<R, TW, TH, Q, IWP>), grid, dim3(256), 0, ctx->stream, fa);
    FNX_HIP(hipGetLastError());
    return FNX_OK;
}

template <typename T>
static int launch_generic(fnx_ctx *ctx, int n, const uint8_t *src, const uint8_t *const *srcs,
                          int sstride, int w, int h, const double *kernel, int radius,
                          uint8_t *dst, uint8_t *const *dsts, int dstride)
{
    const int nt = 2 * radius + 1;
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

int launch_blur(fnx_ctx *ctx, int n, const uint8_t *src, const uint8_t *const *srcs, int sstride,
                int w, int h, const double *kernel, int radius, int flags, uint8_t *dst,
                uint8_t *const *dsts, int dstride)
{
    if (w <= 0 || h <= 0 || n <= 0) return FNX_OK;
    if (flags & FNX_BLUR_EXACT)
        return launch_generic<double>(ctx, n, src, srcs, sstride, w, h, kernel, radius, dst, dsts, dstride);
    if (radius < 1 || radius > FUSED_RMAX)
        return launch_generic<float>(ctx, n, src, srcs, sstride, w, h, kernel, radius, dst, dsts, dstride);
    FusedArgs fa{};
    fa.src = src; fa.srcs = srcs; fa.dst = dst; fa.dsts = dsts;
    fa.sstride = sstride; fa.dstride = dstride; fa.w = w; fa.h = h;
    // batched callers guarantee 16-byte aligned images (checked by the entry point)
    fa.vec_in = srcs ? ((sstride & 15) == 0) : aligned16(src, sstride);
    fa.vec_out = dsts ? ((dstride & 15) == 0) : aligned16(dst, dstride);
    for (int i = 0; i < 2 * radius + 1; i++) fa.wt[i] = static_cast<float>(kernel[i]);
    switch (radius) {
    case 1: return launch_fused<1>(ctx, n, fa);
    case 2: return launch_fused<2>(ctx, n, fa);
    case 3: return launch_fused<3>(ctx, n, fa);
    case 4: return launch_fused<4>(ctx, n, fa);
    case 5: return launch_fused<5>(ctx, n, fa);
    case 6: return launch_fused<6>(ctx, n, fa);
    case 7: return launch_fused<7>(ctx, n, fa);
    case 8: return launch_fused<8>(ctx, n, fa);
    }
    return FNX_ERR_INVALID;
}

}  // namespace fnx
