// GaussianBlur (effects.go:146-220) on gfx950.
//
//  * blur_pass_kernel<T,VERT>: one separable pass, any radius, thread per pixel.
//    T=double is the EXACT mode: fp64, unfused mul+add in the reference's tap order
//    (this TU is built with -ffp-contract=off), so H(tmp uint8) then V is bit-exact.
//    T=float is the generic fast mode for radii the fused kernel is not built for.
//  * blur_direct_kernel<R>: the fast path for radii <= 8 (sigma=2 -> R=6): one launch, both
//    passes in one 64 x ~52 tile, horizontal pass fed straight from global memory into a
//    uint8 LDS intermediate (the reference rounds the intermediate to uint8,
//    effects.go:186-188), vertical pass from LDS.  HBM traffic is read-once/write-once
//    (2*S, halo re-reads hit L2); fp32 FMA accumulation (<=1 LSB off on <=0.1% samples).
#include "common.hpp"
#include "devutil.hpp"

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
//   * 128-lane workgroups, 64 x TH output tile (TH = 64-2R rounded down to a multiple of 4): an H item is 2 rows x 8 outputs (2 per lane), a V item is 2 columns x
//     TH/4 output rows (1 per lane): each intermediate pixel is converted (Q+2R)/Q ~ 1.9 times
//     instead of 4, all lanes busy in both passes.
template <int R, int NTH, int IH>
__global__ __launch_bounds__(NTH, 4) void blur_direct_kernel(FusedArgs a)
{
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

    // ---- horizontal pass (effects.go:169-191): item = 2 rows x 8 outputs, window from global ----
    for (int item = tid; item < HITEMS; item += NTH) {
        const int rp = item / HGROUPS, g = item - rp * HGROUPS;
        const int xs = x0 + HO * g - R;                          // first px of the window
        u32x4 t0[NV], t1[NV];
        if (interior) {
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
        v2f acc[HO][3];
#pragma unroll
        for (int j = 0; j < HO; j++) acc[j][0] = acc[j][1] = acc[j][2] = (v2f){0.5f, 0.5f};
#pragma unroll
        for (int q = 0; q < NV; q++) {
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
            const int c = j + R;
            o0[j] = pk8(acc[j][1].x, 2, pk8(acc[j][0].y, 1, pk8(acc[j][0].x, 0, t0[c / 4][c % 4])));
            o1[j] = pk8(acc[j][2].y, 2, pk8(acc[j][2].x, 1, pk8(acc[j][1].y, 0, t1[c / 4][c % 4])));
        }
        fp32_round_nearest();
#pragma unroll
        for (int b = 0; b < HO / 4; b++) {
            *reinterpret_cast<u32x4 *>(s_tmp + (2 * rp) * TW + HO * g + 4 * b) = (u32x4){o0[4 * b], o0[4 * b + 1], o0[4 * b + 2], o0[4 * b + 3]};
            *reinterpret_cast<u32x4 *>(s_tmp + (2 * rp + 1) * TW + HO * g + 4 * b) = (u32x4){o1[4 * b], o1[4 * b + 1], o1[4 * b + 2], o1[4 * b + 3]};
        }
    }
    __syncthreads();

    // ---- vertical pass (effects.go:195-217): item = 2 columns x Q output rows ----
    {
        const int cp = tid & 31, rg = tid >> 5;                  // column pair, row group
        const int x = x0 + 2 * cp;
        const uint32_t *colp = s_tmp + (rg * Q) * TW + 2 * cp;
        v2f acc[Q][3];                                           // (r0,g0) (b0,r1) (g1,b1)
#pragma unroll
        for (int j = 0; j < Q; j++) acc[j][0] = acc[j][1] = acc[j][2] = (v2f){0.5f, 0.5f};
        uint32_t al0[Q], al1[Q];
        u32x2 tn = *reinterpret_cast<const u32x2 *>(colp);
#pragma unroll
        for (int i = 0; i < Q + 2 * R; i++) {
            const u32x2 t = tn;
            if (i + 1 < Q + 2 * R) tn = *reinterpret_cast<const u32x2 *>(colp + (i + 1) * TW);   // prefetch next row
            if (i >= R && i < R + Q) { al0[i - R] = t.x; al1[i - R] = t.y; }   // centre rows carry the alpha
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
        fp32_round_toward_zero();
#pragma unroll
        for (int j = 0; j < Q; j++) {
            o[j].x = pk8(acc[j][1].x, 2, pk8(acc[j][0].y, 1, pk8(acc[j][0].x, 0, al0[j])));
            o[j].y = pk8(acc[j][2].y, 2, pk8(acc[j][2].x, 1, pk8(acc[j][1].y, 0, al1[j])));
            asm volatile("" : "+v"(o[j].x), "+v"(o[j].y));   // keep the accumulation out of the store branches
        }
        fp32_round_nearest();
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
}

template <int R, int NTH, int IH>
static int launch_direct_cfg(fnx_ctx *ctx, int n, FusedArgs &fa)
{
    constexpr int TW = 64, RG = NTH / 32, TH = ((IH - 2 * R) / RG) * RG;
    fa.tiles_x = (fa.w + TW - 1) / TW;
    fa.tiles = fa.tiles_x * ((fa.h + TH - 1) / TH);
    dim3 grid(8 * ((fa.tiles + 7) / 8), n);
    hipLaunchKernelGGL((blur_direct_kernel<R, NTH, IH>), grid, dim3(NTH), 0, ctx->stream, fa);
    FNX_HIP(hipGetLastError());
    return FNX_OK;
}

// Two tile shapes (measured on MI355X, 4K: 23.6 vs 22.9 us; 1080p: 5.6 vs 6.0 us per image):
//   128 lanes x 64 intermediate rows  (TH = 52 at R=6, H-pass row halo 1.23x)
//   256 lanes x (2*TH+2R) rows         (TH = 104,      halo 1.12x) -- wins when the image height
//     does not leave a mostly empty last tile row and there are enough tiles to fill the chip.
// Cost model: H work ~ staged rows, V work ~ output rows (about 55 : 45 of the instructions).
template <int R>
static int launch_direct(fnx_ctx *ctx, int n, FusedArgs &fa)
{
    constexpr int TH0 = ((64 - 2 * R) / 4) * 4, TH1 = 2 * TH0;
    const long ty0 = (fa.h + TH0 - 1) / TH0, ty1 = (fa.h + TH1 - 1) / TH1;
    const double cost0 = ty0 * (0.55 * (TH0 + 2 * R) + 0.45 * TH0);
    const double cost1 = ty1 * (0.55 * (TH1 + 2 * R) + 0.45 * TH1);
    const long tiles1 = ty1 * ((fa.w + 63) / 64) * n;
    if (cost1 < 0.95 * cost0 && tiles1 >= 4L * ctx->num_cus) return launch_direct_cfg<R, 256, TH1 + 2 * R>(ctx, n, fa);
    return launch_direct_cfg<R, 128, 64>(ctx, n, fa);
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
    for (int i = 0; i < 2 * radius + 1; i++) fa.wt[i] = static_cast<float>(kernel[i]);
    switch (radius) {
    case 1: return launch_direct<1>(ctx, n, fa);
    case 2: return launch_direct<2>(ctx, n, fa);
    case 3: return launch_direct<3>(ctx, n, fa);
    case 4: return launch_direct<4>(ctx, n, fa);
    case 5: return launch_direct<5>(ctx, n, fa);
    case 6: return launch_direct<6>(ctx, n, fa);
    case 7: return launch_direct<7>(ctx, n, fa);
    case 8: return launch_direct<8>(ctx, n, fa);
    }
    return FNX_ERR_INVALID;
}

}  // namespace fnx
