// SSIM family kernels (ssim.go) on gfx950: boxDownsample (integer sums, bit-exact), windowedSSIM
// (+toLuminance fused into the tile load), pixelSSIM.  fp64 throughout (TU built with
// -ffp-contract=off).  windowed_ssim_kernel follows the reference's operation order with unfused
// arithmetic; windowed_ssim_sep_kernel (rank-1 windows: the reference's own Gaussian) computes
// separable moments with explicit fma -- both within 1e-9 of the reference's mean, whose own
// summation order depends on GOMAXPROCS (ssim.go:84-94,155-160).
#include "common.hpp"
#include <type_traits>
#include "devutil.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace fnx {

// ------------------------------------------------------------------------------------
// boxDownsample (ssim.go:244-309)
// ------------------------------------------------------------------------------------
struct BoxArgs {
    const uint8_t *src;
    const uint8_t *const *srcs;
    const uint8_t *src_b;            // optional second image set: z in [n, 2n) reads these,
    const uint8_t *const *srcs_b;    // dst image index stays z (layout [a0..an-1][b0..bn-1])
    int sstride_b, nimg;
    uint8_t *dst;
    size_t dst_image_bytes;
    int sstride, srcW, srcH, dstride, dstW, dstH;
    double xRatio, yRatio;
    int seg;      // output columns per workgroup (tiled kernel)
    int vec_in;
    int packed_ok;  // rows*cols*255 of the largest box < 65536
};

// box edges exactly as ssim.go:255-278
// Any ratio (also upscaling, where boxes overlap / repeat): one thread per output pixel.
__global__ __launch_bounds__(256) void box_generic_kernel(BoxArgs a)
{
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= a.dstW || dy >= a.dstH) return;
    const int z = blockIdx.z;
    const bool second = z >= a.nimg;
    const int zi = second ? z - a.nimg : z;
    const uint8_t *src = second ? (a.srcs_b ? a.srcs_b[zi] : a.src_b) : (a.srcs ? a.srcs[zi] : a.src);
    const int sstride = second ? a.sstride_b : a.sstride;
    int sx0, sx1, sy0, sy1;
    box_edge(dx, a.xRatio, a.srcW, sx0, sx1);
    box_edge(dy, a.yRatio, a.srcH, sy0, sy1);
    unsigned long long r = 0, g = 0, b = 0, al = 0;
    for (int sy = sy0; sy < sy1; sy++) {
        const uint8_t *row = src + static_cast<size_t>(sy) * sstride;
        for (int sx = sx0; sx < sx1; sx++) {
            const uint32_t p = ld_px(row, sx);
            r += p & 0xffu; g += (p >> 8) & 0xffu; b += (p >> 16) & 0xffu; al += p >> 24;
        }
    }
    const long long count = static_cast<long long>(sy1 - sy0) * (sx1 - sx0);
    // count == 0 (possible when upscaling: sx1 == 0) leaves the fresh image's zero pixel (ssim.go:301)
    uint32_t o = 0;
    if (count > 0) {
        const double inv = 1.0 / static_cast<double>(count);
        o = clampF_dev(static_cast<double>(r) * inv) | (clampF_dev(static_cast<double>(g) * inv) << 8) |
            (clampF_dev(static_cast<double>(b) * inv) << 16) | (clampF_dev(static_cast<double>(al) * inv) << 24);
    }
    uint8_t *dimg = a.dst + a.dst_image_bytes * z;
    *reinterpret_cast<uint32_t *>(dimg + static_cast<size_t>(dy) * a.dstride + 4 * static_cast<size_t>(dx)) = o;
}

// Downscaling fast path (ratio >= 1 on both axes => boxes tile the source exactly, each
// source byte is read ONCE): workgroup = (segment of output columns, one output row).
// Every thread streams one 16-byte column chunk down the box rows (all loads independent),
// accumulating per-source-column sums as packed 2x16-bit lanes ((R,B) and (G,A)); the column
// sums go through LDS and one thread per output pixel adds its box columns.
constexpr int BOX_CHUNKS = 256;          // 16-byte chunks (4 px) per workgroup row segment
constexpr int BOX_MAXROWS = 257;         // 257*255 < 65536: packed 16-bit sums cannot overflow

// N rows of one 4-px chunk: per pixel (R,B) = p & 0x00ff00ff and (G,A) = bytes 1,3 moved to the
// 16-bit lanes by one v_perm_b32; the per-column sums stay packed 2 x 16 bit (<= 257 rows)
template <int N>
__device__ __forceinline__ void box_trip(const uint8_t *q, int sstride, uint32_t (&lo)[4], uint32_t (&hi)[4])
{
    u32x4 v[N];
#pragma unroll
    for (int u = 0; u < N; u++) v[u] = ld16_stream(q + static_cast<size_t>(u) * sstride);
#pragma unroll
    for (int u = 0; u < N; u++) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
            lo[e] += v[u][e] & 0x00ff00ffu;
            hi[e] += __builtin_amdgcn_perm(0u, v[u][e], 0x0c030c01u);   // (G, 0, A, 0)
        }
    }
}

// src_off / dst_off: byte offsets added to the job's source and destination (a batched launch's image; kernel arguments are
// never written to -- a store through a dynamic index would move the whole argument block into scratch memory)
template <bool VEC>
__device__ __forceinline__ void box_tiled_body(const BoxArgs &a, const int bx, const int by, const int z, const size_t src_off = 0,
                                               const size_t dst_off = 0)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_col[BOX_CHUNKS * 4 * 2];
    const bool second = z >= a.nimg;
    const int zi = second ? z - a.nimg : z;
    const uint8_t *src = (second ? (a.srcs_b ? a.srcs_b[zi] : a.src_b) : (a.srcs ? a.srcs[zi] : a.src)) + src_off;
    const int sstride = second ? a.sstride_b : a.sstride;
    const int dy = by;
    const int dx_lo = bx * a.seg;
    const int dx_hi = min(dx_lo + a.seg, a.dstW);
    int sy0, sy1, sxa, sxb, t0, t1;
    box_edge(dy, a.yRatio, a.srcH, sy0, sy1);
    box_edge(dx_lo, a.xRatio, a.srcW, sxa, t1);
    box_edge(dx_hi - 1, a.xRatio, a.srcW, t0, sxb);
    const int c0 = sxa >> 2;                       // first chunk of the segment
    const int nchunk = ((sxb + 3) >> 2) - c0;      // <= BOX_CHUNKS by construction of seg
    const int tid = threadIdx.x;

    if (tid < nchunk) {
        const int x = 4 * (c0 + tid);
        uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
        const uint8_t *p = src + static_cast<size_t>(sy0) * sstride;
        // a chunk that sticks out of the row (srcW % 4 != 0) is read pixel by pixel, clamped;
        // the surplus columns belong to no box
        const bool whole = x + 3 < a.srcW;
        if (VEC && whole) {
            // 8 rows per trip, all 16-byte streaming loads in flight before the first use; the
            // last (short) trip is its own fully unrolled body -- no per-row predication
            const uint8_t *q = p + 4 * static_cast<size_t>(x);
            int left = sy1 - sy0;
            for (; left >= 8; left -= 8, q += static_cast<size_t>(8) * sstride) box_trip<8>(q, sstride, lo, hi);
            switch (left) {
            case 7: box_trip<7>(q, sstride, lo, hi); break;
            case 6: box_trip<6>(q, sstride, lo, hi); break;
            case 5: box_trip<5>(q, sstride, lo, hi); break;
            case 4: box_trip<4>(q, sstride, lo, hi); break;
            case 3: box_trip<3>(q, sstride, lo, hi); break;
            case 2: box_trip<2>(q, sstride, lo, hi); break;
            case 1: box_trip<1>(q, sstride, lo, hi); break;
            default: break;
            }
        } else {
            // unaligned image, or the one chunk that sticks out of the row: pixel by pixel, clamped
#pragma unroll 1
            for (int sy = sy0; sy < sy1; sy++) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const uint32_t q = ld_px(p, min(x + e, a.srcW - 1));
                    lo[e] += q & 0x00ff00ffu; hi[e] += (q >> 8) & 0x00ff00ffu;
                }
                p += sstride;
            }
        }
        // s_col[px] = {lo, hi}: two 16-byte stores per lane
        uint4 *sp = reinterpret_cast<uint4 *>(s_col + tid * 8);
        sp[0] = make_uint4(lo[0], hi[0], lo[1], hi[1]);
        sp[1] = make_uint4(lo[2], hi[2], lo[3], hi[3]);
    }
    __syncthreads();

    const int dx = dx_lo + tid;
    if (dx < dx_hi) {
        int sx0, sx1;
        box_edge(dx, a.xRatio, a.srcW, sx0, sx1);
        uint32_t r = 0, g = 0, b = 0, al = 0;
        if (a.packed_ok) {     // whole box fits 16-bit lanes: add the columns packed, unpack once
            uint32_t plo = 0, phi = 0;
            for (int sx = sx0; sx < sx1; sx++) {
                const uint2 c = *reinterpret_cast<const uint2 *>(s_col + (sx - 4 * c0) * 2);
                plo += c.x; phi += c.y;
            }
            r = plo & 0xffffu; b = plo >> 16; g = phi & 0xffffu; al = phi >> 16;
        } else {
            for (int sx = sx0; sx < sx1; sx++) {
                const uint2 c = *reinterpret_cast<const uint2 *>(s_col + (sx - 4 * c0) * 2);
                r += c.x & 0xffffu; b += c.x >> 16;
                g += c.y & 0xffffu; al += c.y >> 16;
            }
        }
        const int count = (sy1 - sy0) * (sx1 - sx0);
        uint8_t *dimg = a.dst + dst_off + a.dst_image_bytes * z;
        *reinterpret_cast<uint32_t *>(dimg + static_cast<size_t>(dy) * a.dstride + 4 * static_cast<size_t>(dx)) =
            box_finish(r, g, b, al, count);
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void box_tiled_kernel(BoxArgs a)
{
    box_tiled_body<VEC>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}

// ------------------------------------------------------------------------------------
// boxDownsample of a decoded JPEG straight from its planes (r3): SSIMFast's <= 256 px plane of toNRGBARef's image
// (convert.go:22-64, ssim.go:52-58) without the image.  The quality search (compress.go:45-74) compares EVERY candidate
// that way -- ycbcr_to_nrgba_kernel wrote 33 MB per 4K candidate and box_tiled_kernel read them back for a 144 KB plane:
// 54 us of the search step's ~130 -- while the planes are 12 MB at 4:2:0.  Same workgroup shape, box edges and sums as
// box_tiled_body; a lane converts its 4-pixel chunk's samples in registers (devutil.hpp's ycc_nrgba_px: the kernel's own
// arithmetic) and adds the packed fields.  XS = the chroma x shift (0, 1, 2).
// ------------------------------------------------------------------------------------
struct BoxYccArgs {
    BoxArgs box;                    // geometry, destination (src fields unused)
    const uint8_t *y, *cb, *cr;
    int ystride, cstride, yshift;
};

template <int XS>
__global__ __launch_bounds__(256) void box_tiled_ycc_kernel(BoxYccArgs ya)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_col[BOX_CHUNKS * 4 * 2];
    const BoxArgs &a = ya.box;
    const int bx = blockIdx.x, dy = blockIdx.y;
    const int dx_lo = bx * a.seg;
    const int dx_hi = min(dx_lo + a.seg, a.dstW);
    int sy0, sy1, sxa, sxb, t0, t1;
    box_edge(dy, a.yRatio, a.srcH, sy0, sy1);
    box_edge(dx_lo, a.xRatio, a.srcW, sxa, t1);
    box_edge(dx_hi - 1, a.xRatio, a.srcW, t0, sxb);
    const int c0 = sxa >> 2;
    const int nchunk = ((sxb + 3) >> 2) - c0;
    const int tid = threadIdx.x;
    if (tid < nchunk) {
        const int x = 4 * (c0 + tid);
        uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
        const bool whole = x + 3 < a.srcW;
        if (whole) {
            // the chunk's Y bytes are one aligned word (the launcher checks), its chroma bytes one word / half word / byte
            const uint8_t *yp = ya.y + x;
            const uint8_t *cbp = ya.cb + (x >> XS), *crp = ya.cr + (x >> XS);
            auto row = [&](int sy, uint32_t &yw, uint32_t &cbw, uint32_t &crw) {
                yw = *(g_u32 *)(yp + static_cast<size_t>(sy) * ya.ystride);
                const size_t co = static_cast<size_t>(sy >> ya.yshift) * ya.cstride;
                if (XS == 0) {
                    cbw = *(g_u32 *)(cbp + co);
                    crw = *(g_u32 *)(crp + co);
                } else if (XS == 1) {
                    cbw = *(__attribute__((address_space(1))) const uint16_t *)(cbp + co);
                    crw = *(__attribute__((address_space(1))) const uint16_t *)(crp + co);
                } else {
                    cbw = cbp[co];
                    crw = crp[co];
                }
            };
            // a chroma sample's three terms of color.YCbCr.RGBA() serve every pixel it covers: the chunk's (4 >> XS) samples
            // once per chroma ROW (two image rows at 4:2:0), then one multiply and three adds per pixel
            constexpr int NC = 4 >> XS;
            int tr[NC], tg[NC], tb[NC];
            auto terms = [&](uint32_t cbw, uint32_t crw) {
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const int cb1 = static_cast<int>((cbw >> (8 * c)) & 0xffu) - 128, cr1 = static_cast<int>((crw >> (8 * c)) & 0xffu) - 128;
                    tr[c] = 91881 * cr1;
                    tg[c] = -22554 * cb1 - 46802 * cr1;
                    tb[c] = 116130 * cb1;
                }
            };
            auto add = [&](uint32_t yw) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int c = e >> XS;
                    const int yy1 = static_cast<int>((yw >> (8 * e)) & 0xffu) * 0x10101;
                    // ycc_u8(v) == med3(v, 0, 2^24 - 1) >> 16 (v >> 16 inside [0, 2^24), else 0 / 255); the channels go
                    // straight into the packed sums' fields (R | B << 16, G | A << 16 with A = 255)
                    const uint32_t r = static_cast<uint32_t>(clampi(yy1 + tr[c], 0, 0xffffff)) >> 16;
                    const uint32_t g = static_cast<uint32_t>(clampi(yy1 + tg[c], 0, 0xffffff)) >> 16;
                    const uint32_t b = static_cast<uint32_t>(clampi(yy1 + tb[c], 0, 0xffffff)) & 0x00ff0000u;
                    lo[e] += r | b;
                    hi[e] += g + 0x00ff0000u;
                }
            };
            // eight rows of loads in flight per trip (24 loads); a short last trip re-reads the box's last row and adds nothing
            // for it -- no row-by-row tail.  The kernel is bound by the conversions, not by memory: 25 us per 4K candidate
            // alone against 18.8 + 9.7 for ycbcr_to_nrgba_kernel + box_tiled_kernel (rocprofv3, tools/time_against_ycc.py);
            // what it saves is the 66 MB of traffic beside the other workers' kernels.  Four lanes per chunk (2 304
            // workgroups instead of 576): the same alone, slower in the pool.
            for (int sy = sy0; sy < sy1; sy += 8) {
                uint32_t yw[8], cbw[8], crw[8];
#pragma unroll
                for (int u = 0; u < 8; u++) row(min(sy + u, sy1 - 1), yw[u], cbw[u], crw[u]);
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    if (sy + u < sy1) {                                       // wave-uniform: a workgroup is one row of boxes
                        if (u == 0 || ((sy + u) >> ya.yshift) != ((sy + u - 1) >> ya.yshift)) terms(cbw[u], crw[u]);
                        add(yw[u]);
                    }
                }
            }
        } else {
            // the one chunk that sticks out of the row: sample by sample, clamped (the surplus columns belong to no box)
            for (int sy = sy0; sy < sy1; sy++) {
                const size_t co = static_cast<size_t>(sy >> ya.yshift) * ya.cstride;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int xx = min(x + e, a.srcW - 1);
                    const uint32_t p = ycc_nrgba_px(ya.y[static_cast<size_t>(sy) * ya.ystride + xx], ya.cb[co + (xx >> XS)], ya.cr[co + (xx >> XS)]);
                    lo[e] += p & 0x00ff00ffu;
                    hi[e] += (p >> 8) & 0x00ff00ffu;
                }
            }
        }
        uint4 *sp = reinterpret_cast<uint4 *>(s_col + tid * 8);
        sp[0] = make_uint4(lo[0], hi[0], lo[1], hi[1]);
        sp[1] = make_uint4(lo[2], hi[2], lo[3], hi[3]);
    }
    __syncthreads();
    const int dx = dx_lo + tid;
    if (dx < dx_hi) {
        int sx0, sx1;
        box_edge(dx, a.xRatio, a.srcW, sx0, sx1);
        uint32_t r = 0, g = 0, b = 0, al = 0;
        for (int sx = sx0; sx < sx1; sx++) {
            const uint2 c = *reinterpret_cast<const uint2 *>(s_col + (sx - 4 * c0) * 2);
            r += c.x & 0xffffu; b += c.x >> 16;
            g += c.y & 0xffffu; al += c.y >> 16;
        }
        const int count = (sy1 - sy0) * (sx1 - sx0);
        *reinterpret_cast<uint32_t *>(a.dst + static_cast<size_t>(dy) * a.dstride + 4 * static_cast<size_t>(dx)) = box_finish(r, g, b, al, count);
    }
}

// *done = false: not this kernel's case (grey, an upscale, huge boxes, unaligned planes) -- the caller converts and downsamples
int launch_box_downsample_ycc(fnx_ctx *ctx, const uint8_t *y, int ystride, const uint8_t *cb, const uint8_t *cr, int cstride,
                              int ratio, int srcW, int srcH, uint8_t *dst, int dstride, int dstW, int dstH, bool *done)
{
    *done = false;
    static const bool off = [] { const char *e = dev_env("FNX_BOX_YCC"); return e && e[0] == '0'; }();   // A/B and tests
    if (off || !cb || !cr || ratio < 0 || ratio > 5 || srcW <= 0 || srcH <= 0 || dstW <= 0 || dstH <= 0) return FNX_OK;
    static const int xs[6] = {0, 1, 1, 0, 2, 2}, ys[6] = {0, 0, 1, 1, 0, 1};
    BoxYccArgs ya{};
    BoxArgs &a = ya.box;
    a.dst = dst; a.srcW = srcW; a.srcH = srcH; a.dstride = dstride; a.dstW = dstW; a.dstH = dstH;
    a.xRatio = static_cast<double>(srcW) / static_cast<double>(dstW);   // ssim.go:251-252
    a.yRatio = static_cast<double>(srcH) / static_cast<double>(dstH);
    const bool tiled = srcW >= dstW && srcH >= dstH && a.yRatio + 1.0 < BOX_MAXROWS &&
                       a.xRatio + 1.0 < BOX_MAXROWS && a.xRatio * 2 + 8 < 4 * BOX_CHUNKS;
    const int cal = 4 >> xs[ratio];                               // bytes of a chunk's chroma
    const bool aligned = (reinterpret_cast<uintptr_t>(y) & 3u) == 0 && (ystride & 3) == 0 &&
                         (reinterpret_cast<uintptr_t>(cb) % cal) == 0 && (reinterpret_cast<uintptr_t>(cr) % cal) == 0 && (cstride % cal) == 0;
    if (!tiled || !aligned) return FNX_OK;
    int seg = static_cast<int>((4 * BOX_CHUNKS - 8) / a.xRatio);
    if (seg > 256) seg = 256;
    if (seg < 1) seg = 1;
    a.seg = seg;
    ya.y = y; ya.cb = cb; ya.cr = cr; ya.ystride = ystride; ya.cstride = cstride; ya.yshift = ys[ratio];
    const dim3 grid((dstW + seg - 1) / seg, dstH);
    switch (xs[ratio]) {
    case 0: hipLaunchKernelGGL(box_tiled_ycc_kernel<0>, grid, dim3(256), 0, ctx->stream, ya); break;
    case 1: hipLaunchKernelGGL(box_tiled_ycc_kernel<1>, grid, dim3(256), 0, ctx->stream, ya); break;
    default: hipLaunchKernelGGL(box_tiled_ycc_kernel<2>, grid, dim3(256), 0, ctx->stream, ya); break;
    }
    FNX_HIP(hipGetLastError());
    *done = true;
    return FNX_OK;
}

// several independent pair downsamples in one launch (MSSSIM: the <= 512 px planes of every level that needs
// them): blockIdx.z = 2 * job + side, grid x / y = the largest job's
constexpr int BOX_MAXJOBS = 4;
struct BoxMulti {
    BoxArgs job[BOX_MAXJOBS];
    int gx[BOX_MAXJOBS], gy[BOX_MAXJOBS];
    // a batch (blockIdx.z = 2 (image * njobs + job) + side): sources in the ctx's pyramid scratch (`src_img` bytes per image),
    // planes in its plane scratch (`dst_img`); njobs == 0: one pair
    int njobs;
    size_t src_img, dst_img;
};

__global__ __launch_bounds__(256) void box_tiled_multi_kernel(BoxMulti m)
{
    const int jz = blockIdx.z >> 1;
    const int img = m.njobs ? jz / m.njobs : 0, j = m.njobs ? jz - img * m.njobs : jz;
    if (static_cast<int>(blockIdx.x) >= m.gx[j] || static_cast<int>(blockIdx.y) >= m.gy[j]) return;
    box_tiled_body<true>(m.job[j], blockIdx.x, blockIdx.y, blockIdx.z & 1, m.src_img * img, m.dst_img * img);
}

// "workgroups finished" counters of the kernels whose last workgroup takes the final sum itself: 2 x 4096 for the
// marching SSIM kernels (one half per stream the ctx launches on), one more for MSSSIM's multi-level window launch.
// Zero between launches (the last workgroup puts its counter back).
// Words DONE_SCAN .. + 15 belong to the single-launch scans of analyze.hip (isOpaque / isGrayscale).
constexpr int SSIM_DONE_WORDS = 2 * 4096 + 32;
int ssim_done_counters(fnx_ctx *ctx, unsigned **out)
{
    const void *before = ctx->slot[SLOT_DONE].p;
    void *dn = nullptr;
    FNX_TRY(scratch(ctx, SLOT_DONE, sizeof(unsigned) * SSIM_DONE_WORDS, &dn));
    if (dn != before) {                                           // first use: zero once, for both streams
        FNX_HIP(hipMemsetAsync(dn, 0, sizeof(unsigned) * SSIM_DONE_WORDS, ctx->stream));
        FNX_HIP(hipStreamSynchronize(ctx->stream));
    }
    *out = static_cast<unsigned *>(dn);
    return FNX_OK;
}

int launch_box_downsample(fnx_ctx *ctx, int n, const uint8_t *src, const uint8_t *const *srcs,
                          int sstride, int srcW, int srcH, uint8_t *dst, int dstride,
                          size_t dst_image_bytes, int dstW, int dstH)
{
    return launch_box_downsample_pair(ctx, n, src, srcs, sstride, nullptr, nullptr, 0, srcW, srcH, dst, dstride,
                                      dst_image_bytes, dstW, dstH);
}

// Both sides of an SSIM comparison in ONE launch: images a (z < n) and b (z >= n), same geometry;
// dst image z lives at dst + z*dst_image_bytes.  b == bs == nullptr: a only.
int launch_box_downsample_pair(fnx_ctx *ctx, int n, const uint8_t *src, const uint8_t *const *srcs,
                               int sstride, const uint8_t *src_b, const uint8_t *const *srcs_b, int sstride_b,
                               int srcW, int srcH, uint8_t *dst, int dstride, size_t dst_image_bytes,
                               int dstW, int dstH)
{
    if (srcW <= 0 || srcH <= 0 || dstW <= 0 || dstH <= 0 || n <= 0) return FNX_OK;
    const bool pair = src_b || srcs_b;
    BoxArgs a{};
    a.src = src; a.srcs = srcs; a.dst = dst; a.dst_image_bytes = dst_image_bytes;
    a.src_b = src_b; a.srcs_b = srcs_b; a.sstride_b = sstride_b; a.nimg = n;
    a.sstride = sstride; a.srcW = srcW; a.srcH = srcH; a.dstride = dstride; a.dstW = dstW; a.dstH = dstH;
    a.xRatio = static_cast<double>(srcW) / static_cast<double>(dstW);   // ssim.go:251-252
    a.yRatio = static_cast<double>(srcH) / static_cast<double>(dstH);
    a.vec_in = srcs ? ((sstride & 15) == 0) : aligned16(src, sstride);
    if (pair) a.vec_in = a.vec_in && (srcs_b ? ((sstride_b & 15) == 0) : aligned16(src_b, sstride_b));
    const int nz = pair ? 2 * n : n;
    const bool tiled = srcW >= dstW && srcH >= dstH && a.yRatio + 1.0 < BOX_MAXROWS &&
                       a.xRatio + 1.0 < BOX_MAXROWS && a.xRatio * 2 + 8 < 4 * BOX_CHUNKS;
    a.packed_ok = (static_cast<double>(static_cast<long>(a.yRatio) + 2) * static_cast<double>(static_cast<long>(a.xRatio) + 2) * 255.0) < 65536.0;
    if (tiled) {
        // seg output columns span < seg*xRatio + 1 source px, plus < 4 px of chunk alignment each side
        int seg = static_cast<int>((4 * BOX_CHUNKS - 8) / a.xRatio);
        if (seg > 256) seg = 256;
        if (seg < 1) seg = 1;
        a.seg = seg;
        dim3 grid((dstW + seg - 1) / seg, dstH, nz);
        if (a.vec_in) hipLaunchKernelGGL(box_tiled_kernel<true>, grid, dim3(256), 0, ctx->stream, a);
        else hipLaunchKernelGGL(box_tiled_kernel<false>, grid, dim3(256), 0, ctx->stream, a);
    } else {
        dim3 grid((dstW + 63) / 64, (dstH + 3) / 4, nz);
        hipLaunchKernelGGL(box_generic_kernel, grid, dim3(256), 0, ctx->stream, a);
    }
    FNX_HIP(hipGetLastError());
    return FNX_OK;
}

// ------------------------------------------------------------------------------------
// windowedSSIM (ssim.go:73-166) with toLuminance (ssim.go:207-220) fused into the tile load
// ------------------------------------------------------------------------------------
constexpr int WS_TX = 32, WS_TY = 8;   // windows per workgroup (one per thread)

// image z of a batched launch: by pointer array (separately allocated images: fnx_ssim_batch_enqueue) or at a fixed spacing
__device__ __forceinline__ const uint8_t *batch_img(const uint8_t *base, const uint8_t *const *ptrs, size_t image_bytes, int z)
{
    return ptrs ? ptrs[z] : base + image_bytes * z;
}

struct WinArgs {
    const uint8_t *a;
    const uint8_t *b;
    const uint8_t *const *as, *const *bs;    // non-null: image z is as[z] / bs[z]
    size_t a_image_bytes, b_image_bytes;
    int astride, bstride, w, h;
    int tiles_x, tiles;    // per image pair
    const double *window;  // 64 weights, row-major wy,wx in [-4,4)
    double *partial;       // [n][tiles]
};

__global__ __launch_bounds__(256) void windowed_ssim_kernel(WinArgs a)
{
    constexpr int LW = WS_TX + 7, LH = WS_TY + 7;
    __shared__ double s_a[LH * LW], s_b[LH * LW], s_k[64], s_red[4];
    const int z = blockIdx.y;
    const int tile = blockIdx.x;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    // window (wx0+i, wy0+j) covers pixels [wx0+i, wx0+i+8) x [wy0+j, wy0+j+8): centre (x,y)=(wx+4,wy+4)
    const int wx0 = tx * WS_TX, wy0 = ty * WS_TY;
    const uint8_t *A = batch_img(a.a, a.as, a.a_image_bytes, z);
    const uint8_t *B = batch_img(a.b, a.bs, a.b_image_bytes, z);
    const int tid = threadIdx.x;
    if (tid < 64) s_k[tid] = a.window[tid];
    for (int i = tid; i < LH * LW; i += 256) {
        const int ly = i / LW, lx = i - ly * LW;
        const int x = min(wx0 + lx, a.w - 1), y = min(wy0 + ly, a.h - 1);
        s_a[i] = lum601(ld_px(A + static_cast<size_t>(y) * a.astride, x));
        s_b[i] = lum601(ld_px(B + static_cast<size_t>(y) * a.bstride, x));
    }
    __syncthreads();
    const int lx = tid & (WS_TX - 1), ly = tid / WS_TX;
    const int wx = wx0 + lx, wy = wy0 + ly;
    double val = 0;
    if (wx < a.w - 8 && wy < a.h - 8) {     // (w-8)*(h-8) windows (ssim.go:110-111)
        double muA = 0, muB = 0;
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const double wt = s_k[j * 8 + i];
                muA += s_a[(ly + j) * LW + lx + i] * wt;
                muB += s_b[(ly + j) * LW + lx + i] * wt;
            }
        double sAA = 0, sBB = 0, sAB = 0;
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const double wt = s_k[j * 8 + i];
                const double da = s_a[(ly + j) * LW + lx + i] - muA;
                const double db = s_b[(ly + j) * LW + lx + i] - muB;
                sAA += da * da * wt;
                sBB += db * db * wt;
                sAB += da * db * wt;
            }
        const double num = (2 * muA * muB + 6.5025) * (2 * sAB + 58.5225);
        const double den = (muA * muA + muB * muB + 6.5025) * (sAA + sBB + 58.5225);
        val = num / den;
    }
    const double t = block_sum_256(val, s_red);
    if (tid == 0) a.partial[static_cast<size_t>(z) * a.tiles + tile] = t;
}

// Separable form.  gaussianKernel(8,1.5) is rank-1: k[j][i] = r[j]*c[i] (ssim.go:231, exp of a
// sum), so the window moments E[a], E[b], E[a^2 + b^2], E[ab] (SSIM only needs sigma_aa + sigma_bb and
// sigma_ab) are two 8-tap passes instead of 64 taps x 2 sweeps, and sigma = E[x^2] - mu^2.  Algebraically identical to the reference's
// two-sweep form; in fp64 the results differ by ~1e-13 (inside the 1e-9 bar).  The host checks
// that the caller's table really is rank-1 and otherwise uses windowed_ssim_kernel above.
constexpr int WSS_TX = 32, WSS_TY = 16;   // windows per workgroup (256 lanes, 2 per lane)

struct WinSepArgs {
    const uint8_t *a;
    const uint8_t *b;
    const uint8_t *const *as, *const *bs;    // non-null: image z is as[z] / bs[z]
    size_t a_image_bytes, b_image_bytes;
    int astride, bstride, w, h;
    int tiles_x, tiles;
    double *partial;
    double col[8], row[8];   // k[j][i] ~= row[j] * col[i]
    // boxed != 0 (r3, MSSSIM's middle levels): a and b are srcW x srcH images and the w x h plane the windows run over
    // is their boxDownsample (ssim.go:244-309), taken on the fly in the tile load -- boxes of at most 5 x 5 pixels, the
    // integer sums and the fp64 finish of box_tiled_kernel, so the plane's pixels are the same bytes and no kernel
    // has to write them first
    int boxed, srcW, srcH;
    double xRatio, yRatio;
    // out != nullptr (windowed_ssim_sep_kernel): the LAST workgroup of an image to finish takes the image's mean -- the sum
    // ssim_finish_kernel would take, in the same order (the quality search's planes: a launch less per candidate)
    double *out;
    unsigned *done;          // [n], zero before and after the launch
    double count;
};

// the box of at most MB x MB pixels behind plane pixel (x, y): every load is issued before the first is used (indices
// clamped into the box, the surplus masked out of the sums) -- a loop over the box's own extent is a chain of dependent
// misses per staged pixel, and this kernel has few workgroups to hide them behind
template <int MB>
__device__ __forceinline__ uint32_t win_box_px(const WinSepArgs &a, const uint8_t *img, int stride, int x, int y)
{
    int sx0, sx1, sy0, sy1;
    box_edge(x, a.xRatio, a.srcW, sx0, sx1);
    box_edge(y, a.yRatio, a.srcH, sy0, sy1);
    uint32_t p[MB][MB];
#pragma unroll
    for (int jy = 0; jy < MB; jy++) {
        const uint8_t *row = img + static_cast<size_t>(min(sy0 + jy, sy1 - 1)) * stride;
#pragma unroll
        for (int jx = 0; jx < MB; jx++) p[jy][jx] = ld_px(row, min(sx0 + jx, sx1 - 1));
    }
    uint32_t rb = 0, ga = 0;                             // 25 x 255 < 65536: packed 16-bit fields cannot carry
#pragma unroll
    for (int jy = 0; jy < MB; jy++)
#pragma unroll
        for (int jx = 0; jx < MB; jx++) {
            const bool in = sy0 + jy < sy1 && sx0 + jx < sx1;
            rb += in ? p[jy][jx] & 0x00ff00ffu : 0u;
            ga += in ? (p[jy][jx] >> 8) & 0x00ff00ffu : 0u;
        }
    return box_finish(rb & 0xffffu, ga & 0xffffu, rb >> 16, ga >> 16, (sy1 - sy0) * (sx1 - sx0));
}

// pixel (x, y) of the plane a window job runs over: the image's own, or the box mean of the larger image behind it
// (a.boxed = the largest box side, 2..5)
__device__ __forceinline__ uint32_t win_plane_px(const WinSepArgs &a, const uint8_t *img, int stride, int x, int y)
{
    switch (a.boxed) {                                   // uniform over the workgroup
    case 0: return ld_px(img + static_cast<size_t>(y) * stride, x);
    case 1:
    case 2: return win_box_px<2>(a, img, stride, x, y);
    case 3: return win_box_px<3>(a, img, stride, x, y);
    case 4: return win_box_px<4>(a, img, stride, x, y);
    default: return win_box_px<5>(a, img, stride, x, y);
    }
}

// TY = 16 rows of windows per workgroup (a 32-row tile trims the halos from 1.75x / 1.44x to 1.49x /
// 1.22x but needs 70 KB of LDS: 2 workgroups per CU, measured 7 % slower at 8K).  The kernel is fp64-issue bound (fp64 runs at
// half rate), so everything is cut for fp64 instructions per window: four moments, 2 horizontally
// adjacent H outputs and TY/8 vertically adjacent windows per lane.  (A 3 x 256 LDS table of the
// luminance products saved 6 fp64 ops per pixel and cost 20 %: dependent, bank-conflicting reads.)
// Big planes take windowed_ssim_sep24_kernel below; this one serves planes with too few windows for it.
template <int TY, int NTHR, bool STORE = true>
__device__ __forceinline__ double ssim_sep_body(const WinSepArgs &a, const int tile, const int z)
{
    constexpr int LW = WSS_TX + 8, LH = TY + 7;   // 39 columns are used; an even pitch keeps row starts 16-byte aligned
    constexpr int WPT = WSS_TX * TY / NTHR;       // vertically adjacent windows per lane
    __shared__ __attribute__((aligned(16))) double s_a[LH * LW], s_b[LH * LW];
    __shared__ __attribute__((aligned(16))) double s_h[4][LH * WSS_TX];   // E[a], E[b], E[a^2 + b^2], E[ab] after the H pass
    __shared__ double s_red[NTHR / 64];
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int wx0 = tx * WSS_TX, wy0 = ty * TY;
    const uint8_t *A = batch_img(a.a, a.as, a.a_image_bytes, z);
    const uint8_t *B = batch_img(a.b, a.bs, a.b_image_bytes, z);
    const int tid = threadIdx.x;
    if (!a.boxed) {        // (its own loop: with the boxed form inside it the plain loads were no longer issued together -- 9.4 -> 13.9 us per 4K MSSSIM)
        for (int i = tid; i < LH * LW; i += NTHR) {
            const int ly = i / LW, lx = i - ly * LW;
            const int x = min(wx0 + lx, a.w - 1), y = min(wy0 + ly, a.h - 1);
            s_a[i] = lum601(ld_px(A + static_cast<size_t>(y) * a.astride, x));
            s_b[i] = lum601(ld_px(B + static_cast<size_t>(y) * a.bstride, x));
        }
    } else {
#pragma unroll 1
        for (int i = tid; i < LH * LW; i += NTHR) {
            const int ly = i / LW, lx = i - ly * LW;
            const int x = min(wx0 + lx, a.w - 1), y = min(wy0 + ly, a.h - 1);
            s_a[i] = lum601(win_plane_px(a, A, a.astride, x, y));
            s_b[i] = lum601(win_plane_px(a, B, a.bstride, x, y));
        }
    }
    __syncthreads();
    // horizontal 8-tap pass; item = (row, 2 adjacent outputs): the 9-value window is read once
    for (int i = tid; i < LH * (WSS_TX / 2); i += NTHR) {
        const int r = i / (WSS_TX / 2), x = 2 * (i - r * (WSS_TX / 2));
        // 16-byte reads of value pairs: lanes step by 2 doubles, so 8-byte reads of x+t hit every other
        // bank pair (PMC at 8K: LDS busy 68 % of the kernel, two thirds of it bank conflicts); pairs
        // make the access linear across the wave
        double va[10], vb[10];
#pragma unroll
        for (int t = 0; t < 10; t += 2) {
            const double2 pa = *reinterpret_cast<const double2 *>(&s_a[r * LW + x + t]);
            const double2 pb = *reinterpret_cast<const double2 *>(&s_b[r * LW + x + t]);
            va[t] = pa.x; va[t + 1] = pa.y;
            vb[t] = pb.x; vb[t + 1] = pb.y;
        }
        // SSIM needs sigma_aa + sigma_bb and sigma_ab only, so four moments suffice:
        // E[a], E[b], E[a^2 + b^2], E[ab]
        double h[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const double sq = fma(vb[t], vb[t], va[t] * va[t]), ab = va[t] * vb[t];
#pragma unroll
            for (int o = 0; o < 2; o++) {
                const int k = t - o;
                if (k >= 0 && k < 8) {
                    const double c = a.col[k];
                    h[o][0] = fma(va[t], c, h[o][0]);
                    h[o][1] = fma(vb[t], c, h[o][1]);
                    h[o][2] = fma(sq, c, h[o][2]);
                    h[o][3] = fma(ab, c, h[o][3]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) *reinterpret_cast<double2 *>(&s_h[q][r * WSS_TX + x]) = make_double2(h[0][q], h[1][q]);
    }
    __syncthreads();
    // vertical 8-tap pass + the SSIM formula (ssim.go:142-145); a lane owns WPT vertically
    // adjacent windows and reads their WPT+7 rows once
    double val = 0;
    {
        const int lx = tid & (WSS_TX - 1), ly = WPT * (tid / WSS_TX);
        const int wx = wx0 + lx;
        double m[WPT][4];
#pragma unroll
        for (int o = 0; o < WPT; o++) m[o][0] = m[o][1] = m[o][2] = m[o][3] = 0;
#pragma unroll
        for (int j = 0; j < WPT + 7; j++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const double hv = s_h[q][(ly + j) * WSS_TX + lx];
#pragma unroll
                for (int o = 0; o < WPT; o++) {
                    const int k = j - o;
                    if (k >= 0 && k < 8) m[o][q] = fma(hv, a.row[k], m[o][q]);
                }
            }
        }
#pragma unroll
        for (int o = 0; o < WPT; o++) {
            const int wy = wy0 + ly + o;
            if (wx < a.w - 8 && wy < a.h - 8) {
                const double muA = m[o][0], muB = m[o][1];
                const double mu2 = muA * muA + muB * muB, muAB = muA * muB;
                const double sSum = m[o][2] - mu2, sAB = m[o][3] - muAB;    // sigma_aa + sigma_bb, sigma_ab
                const double num = (2 * muAB + 6.5025) * (2 * sAB + 58.5225);
                const double den = (mu2 + 6.5025) * (sSum + 58.5225);
                // den >= C1*C2 > 0 and far from the subnormal range: reciprocal + 2 Newton steps
                // is accurate to ~1 ulp without the IEEE division's scale/fixup sequence
                double rc = __builtin_amdgcn_rcp(den);
                rc = fma(fma(-den, rc, 1.0), rc, rc);
                rc = fma(fma(-den, rc, 1.0), rc, rc);
                val += num * rc;
            }
        }
    }
    // fixed-shape tree: lanes of a wave, then the waves in order
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) val += __shfl_down(val, off, 64);
    if ((tid & 63) == 0) s_red[tid >> 6] = val;
    __syncthreads();
    double t = 0.0;
    if (tid == 0) {
        t = s_red[0];
#pragma unroll
        for (int wv = 1; wv < NTHR / 64; wv++) t += s_red[wv];
        if (STORE) a.partial[static_cast<size_t>(z) * a.tiles + tile] = t;
    }
    return t;                                            // the tile's sum (thread 0)
}

__device__ __forceinline__ double finish_sum_256(const double *p, int tiles, double *s_red);

template <int TY, int NTHR>
__global__ __launch_bounds__(NTHR) void windowed_ssim_sep_kernel(WinSepArgs a)
{
    __shared__ double s_fin[4];
    __shared__ int s_last;
    const int z = blockIdx.y;
    const double t = ssim_sep_body<TY, NTHR, false>(a, blockIdx.x, z);
    double *pp = a.partial + static_cast<size_t>(z) * a.tiles + blockIdx.x;
    if (!a.out) {
        if (threadIdx.x == 0) *pp = t;
        return;
    }
    if (threadIdx.x == 0) {
        // written through, then the image's counter (see march_finish)
        asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" ::"v"(pp), "v"(t) : "memory");
        const unsigned prev = __hip_atomic_fetch_add(a.done + z, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = prev == static_cast<unsigned>(a.tiles) - 1u;
    }
    __syncthreads();
    if (s_last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const double sum = finish_sum_256(a.partial + static_cast<size_t>(z) * a.tiles, a.tiles, s_fin);
        if (threadIdx.x == 0) {
            a.out[z] = a.count > 0 ? sum / a.count : 1.0;
            __hip_atomic_store(a.done + z, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// several single-pair window jobs of different sizes in one launch (MSSSIM's levels): blockIdx.y = job
constexpr int WS_MAXJOBS = 5;
struct WinSepMulti {
    WinSepArgs job[WS_MAXJOBS];
    // out != nullptr: the LAST workgroup of a job to finish also takes the job's mean -- the sum
    // ssim_finish_multi_kernel would take, in the same order -- so that no sixth launch waits behind this one
    double *out;
    unsigned *done;                  // a counter per job, zero before and after the launch
    int njobs;
    int out_index[WS_MAXJOBS];
    double windows[WS_MAXJOBS];
    // a batch (blockIdx.y = image * nlev + job; out == nullptr): each job's a_image_bytes / b_image_bytes is the spacing of its
    // planes between images, the partial sums of image i lie part_img doubles behind image i - 1's; nlev == 0: one pair
    int nlev;
    size_t part_img;
};

__device__ __forceinline__ double finish_sum_256(const double *p, int tiles, double *s_red);

__global__ __launch_bounds__(256) void windowed_ssim_sep_multi_kernel(WinSepMulti m)
{
    __shared__ double s_fin[4];
    __shared__ int s_last;
    const int img = m.nlev ? blockIdx.y / m.nlev : 0, jb = m.nlev ? blockIdx.y - img * m.nlev : blockIdx.y;
    const WinSepArgs &a = m.job[jb];
    if (static_cast<int>(blockIdx.x) >= a.tiles) return;
    // (ONE instantiation of the body: its LDS arrays are per instantiation, and two of them halved the kernel's occupancy)
    const double t = ssim_sep_body<WSS_TY, 256, false>(a, blockIdx.x, img);
    if (!m.out) {
        if (threadIdx.x == 0) a.partial[m.part_img * img + blockIdx.x] = t;
        return;
    }
    if (threadIdx.x == 0) {
        // written through, then the job's counter (see march_finish: an agent-scope release would be an L2 write-back per workgroup)
        double *pp = a.partial + blockIdx.x;
        asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" ::"v"(pp), "v"(t) : "memory");
        const unsigned prev = __hip_atomic_fetch_add(m.done + blockIdx.y, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = prev == static_cast<unsigned>(a.tiles) - 1u;
    }
    __syncthreads();
    if (s_last) {                                        // the job's last workgroup: its mean (the five jobs end independently)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const int j = blockIdx.y;
        const double sum = finish_sum_256(a.partial, a.tiles, s_fin);
        if (threadIdx.x == 0) {
            m.out[m.out_index[j]] = m.windows[j] > 0 ? sum / m.windows[j] : 1.0;
            __hip_atomic_store(m.done + j, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// rank-1 factorisation of the 8x8 table: col[i] = sum_j k[j][i], row[j] = sum_i k[j][i] / sum(k).
// Accepted when every entry is reproduced to 1e-13 relative (the Gaussian is, to ~3e-16).
static bool window_rank1(const double *k, double *col, double *row)
{
    double total = 0;
    for (int i = 0; i < 8; i++) col[i] = row[i] = 0;
    for (int j = 0; j < 8; j++)
        for (int i = 0; i < 8; i++) {
            col[i] += k[j * 8 + i];
            row[j] += k[j * 8 + i];
            total += k[j * 8 + i];
        }
    if (!(total > 0)) return false;
    double kmax = 0;
    for (int i = 0; i < 64; i++) kmax = k[i] > kmax ? k[i] : kmax;
    for (int j = 0; j < 8; j++) row[j] /= total;
    for (int j = 0; j < 8; j++)
        for (int i = 0; i < 8; i++) {
            const double d = k[j * 8 + i] - row[j] * col[i];
            if (!(d <= 1e-13 * kmax && d >= -1e-13 * kmax)) return false;
        }
    return true;
}

// The same two passes on a 32 x 24-window tile and 256 lanes, for big planes.  Shape: (24 + 7) rows x 16
// two-output H items = 496 items = 2 rounds of 256 lanes at 97 % (the 32 x 32 / 512-lane shape fills 61 % of
// its second round: PMC counted 213 VALU wave-instructions per window, a quarter of them on idle lanes),
// 3 vertically adjacent windows per lane in the V pass.  LDS: the H results of rows 0..15 (round 1) get
// their own 16 KB; those of rows 16..30 (round 2) are held in registers across a barrier and then
// overwrite the luminance tile, which nobody reads any more -- 36 KB per workgroup = 16 waves per CU, where
// separate arrays (52 KB, 12 waves per CU) measured no faster than the old shape.
constexpr int W24_TY = 24, W24_LH = W24_TY + 7, W24_LW = WSS_TX + 8, W24_R1 = 16;
__global__ __launch_bounds__(256) void windowed_ssim_sep24_kernel(WinSepArgs a)
{
    constexpr int LH = W24_LH, LW = W24_LW, TX = WSS_TX, NTHR = 256, WPT = 3;
    constexpr int LUM = LH * LW;                              // doubles per luminance tile
    constexpr int ROWH = 4 * TX;                              // doubles per row of H results: [moment][x]
    static_assert((LH - W24_R1) * ROWH <= 2 * LUM, "round-2 results must fit over the luminance tiles");
    __shared__ __attribute__((aligned(16))) double s_lds[2 * LUM + W24_R1 * ROWH];
    __shared__ double s_red[NTHR / 64];
    double *s_a = s_lds, *s_b = s_lds + LUM;
    double *s_h1 = s_lds + 2 * LUM;                           // rows 0..15
    double *s_h2 = s_lds;                                     // rows 16..30, over s_a / s_b
    const int z = blockIdx.y;
    const int tile = blockIdx.x;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int wx0 = tx * TX, wy0 = ty * W24_TY;
    const uint8_t *A = batch_img(a.a, a.as, a.a_image_bytes, z);
    const uint8_t *B = batch_img(a.b, a.bs, a.b_image_bytes, z);
    const int tid = threadIdx.x;
    for (int i = tid; i < LUM; i += NTHR) {
        const int ly = i / LW, lx = i - ly * LW;
        const int x = min(wx0 + lx, a.w - 1), y = min(wy0 + ly, a.h - 1);
        s_a[i] = lum601(ld_px(A + static_cast<size_t>(y) * a.astride, x));
        s_b[i] = lum601(ld_px(B + static_cast<size_t>(y) * a.bstride, x));
    }
    __syncthreads();
    // horizontal pass: item = (row, 2 adjacent outputs), as in windowed_ssim_sep_kernel
    auto h_item = [&](int r, int x, double (&h)[2][4]) {
        double va[10], vb[10];
#pragma unroll
        for (int t = 0; t < 10; t += 2) {
            const double2 pa = *reinterpret_cast<const double2 *>(&s_a[r * LW + x + t]);
            const double2 pb = *reinterpret_cast<const double2 *>(&s_b[r * LW + x + t]);
            va[t] = pa.x; va[t + 1] = pa.y;
            vb[t] = pb.x; vb[t + 1] = pb.y;
        }
#pragma unroll
        for (int o = 0; o < 2; o++) h[o][0] = h[o][1] = h[o][2] = h[o][3] = 0;
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const double sq = fma(vb[t], vb[t], va[t] * va[t]), ab = va[t] * vb[t];
#pragma unroll
            for (int o = 0; o < 2; o++) {
                const int k = t - o;
                if (k >= 0 && k < 8) {
                    const double c = a.col[k];
                    h[o][0] = fma(va[t], c, h[o][0]);
                    h[o][1] = fma(vb[t], c, h[o][1]);
                    h[o][2] = fma(sq, c, h[o][2]);
                    h[o][3] = fma(ab, c, h[o][3]);
                }
            }
        }
    };
    const int hr = tid / (TX / 2), hx = 2 * (tid - hr * (TX / 2));      // this lane's item within a round
    {
        double h[2][4];
        h_item(hr, hx, h);                                              // round 1: rows 0..15
#pragma unroll
        for (int q = 0; q < 4; q++) *reinterpret_cast<double2 *>(&s_h1[hr * ROWH + q * TX + hx]) = make_double2(h[0][q], h[1][q]);
    }
    {
        double h[2][4];
        const bool live = W24_R1 + hr < LH;                             // round 2: rows 16..30
        if (live) h_item(W24_R1 + hr, hx, h);
        __syncthreads();                                                // every read of the luminance tiles is done
        if (live) {
#pragma unroll
            for (int q = 0; q < 4; q++) *reinterpret_cast<double2 *>(&s_h2[hr * ROWH + q * TX + hx]) = make_double2(h[0][q], h[1][q]);
        }
    }
    __syncthreads();
    double val = 0;
    {
        const int lx = tid & (TX - 1), ly = WPT * (tid / TX);
        const int wx = wx0 + lx;
        double m[WPT][4];
#pragma unroll
        for (int o = 0; o < WPT; o++) m[o][0] = m[o][1] = m[o][2] = m[o][3] = 0;
#pragma unroll
        for (int j = 0; j < WPT + 7; j++) {
            const int r = ly + j;
            const double *hrow = (r < W24_R1 ? s_h1 + r * ROWH : s_h2 + (r - W24_R1) * ROWH) + lx;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const double hv = hrow[q * TX];
#pragma unroll
                for (int o = 0; o < WPT; o++) {
                    const int k = j - o;
                    if (k >= 0 && k < 8) m[o][q] = fma(hv, a.row[k], m[o][q]);
                }
            }
        }
#pragma unroll
        for (int o = 0; o < WPT; o++) {
            const int wy = wy0 + ly + o;
            if (wx < a.w - 8 && wy < a.h - 8) {
                const double muA = m[o][0], muB = m[o][1];
                const double mu2 = muA * muA + muB * muB, muAB = muA * muB;
                const double sSum = m[o][2] - mu2, sAB = m[o][3] - muAB;
                const double num = (2 * muAB + 6.5025) * (2 * sAB + 58.5225);
                const double den = (mu2 + 6.5025) * (sSum + 58.5225);
                double rc = __builtin_amdgcn_rcp(den);                  // see windowed_ssim_sep_kernel
                rc = fma(fma(-den, rc, 1.0), rc, rc);
                rc = fma(fma(-den, rc, 1.0), rc, rc);
                val += num * rc;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) val += __shfl_down(val, off, 64);
    if ((tid & 63) == 0) s_red[tid >> 6] = val;
    __syncthreads();
    if (tid == 0) {
        double t = s_red[0];
#pragma unroll
        for (int wv = 1; wv < NTHR / 64; wv++) t += s_red[wv];
        a.partial[static_cast<size_t>(z) * a.tiles + tile] = t;
    }
}

// ------------------------------------------------------------------------------------
// windowed SSIM, column-strip marching form (rank-1 windows) -- the kernel every big plane takes.
//
// The tile kernels above pay for a 2-D halo (1.61x luminance conversions per window at 32 x 24), an LDS
// round trip for BOTH passes and ~30 instructions of staging arithmetic per staged pixel: PMC counted 213
// VALU wave-instructions per window against a floor of 64 fp64 FMAs (4 moments x 8 taps x 2 passes).  Here a
// WAVE owns a strip of 57 window columns (64 pixel columns, one per lane) and marches DOWN a segment of rows:
//   * per row a lane loads ITS pixel of a and b once, turns each into the integer milli-luminance
//     I = 299 R + 587 G + 114 B with three v_dot4_u32_u8 (exact; the reference's fp64 0.299 R + 0.587 G +
//     0.114 B is I / 1000 to 3 ulp -- the SSIM formula is scale-invariant once C1, C2 are scaled by 1e6, so
//     no division is needed and the result stays within 1e-12 of the reference's, bar 1e-9);
//   * the horizontal 8-tap pass reads the neighbours' (a, b) and (a^2 + b^2, ab) pairs from a per-wave LDS row
//     (16-byte reads at a 16-byte lane stride: conflict-free; no barrier -- one wave, in-order LDS);
//   * the vertical pass never leaves registers: a ring of the 8 windows in flight per column x 4 moments,
//     unrolled over the 8 phases of the ring so that every index is static; the window whose 8th row just
//     arrived goes through the SSIM formula (rcp + 2 Newton steps) and into the lane's sum.
// Row halo (S + 7) / S for a segment of S window rows, column halo 64 / 57; about 120 VALU instructions
// per window.  Segments are sized so that the launch is a few thousand waves (launch_windowed_ssim).
// ------------------------------------------------------------------------------------
constexpr int WM_COLS = 57;          // window columns per wave (64 lanes - 7)
constexpr int WM_LDSW = 72;          // LDS row entries per wave (lane + 7 taps, padded)
constexpr int WM_PF = 4;             // pixel rows in flight per lane (divides the 8 ring phases)

struct MarchArgs {
    const uint8_t *a;
    const uint8_t *b;
    const uint8_t *const *as, *const *bs;    // non-null: image z is as[z] / bs[z]
    size_t a_image_bytes, b_image_bytes;
    int astride, bstride, w, h;
    int strips, segs, seg_rows;   // per image: strips x segs wave-sized work items, seg_rows window rows each
    double *partial;              // [n][strips * segs]
    double col[8], row[8];        // k[j][i] ~= row[j] * col[i]
    // out != nullptr: the LAST workgroup of an image to finish also takes the image's mean (the sum ssim_finish_kernel
    // would take, in the same order) -- no second launch behind a kernel whose waves all end together
    double *out;
    unsigned *done;               // [n] workgroups finished, zero before and after the launch
    double count;
    int prio;                     // march2: alternate the wave's priority by progress (A/B knob FNX_SSIM_PRIO)
    // the fp32 form (windowed_ssim_march2f_kernel): col[] rounded to fp32; row[] * 2^-10 (first moments) and * 2^-20 (second
    // moments) -- exact scalings: the moments come out in units of milli-luminance / 1024, whose squares stay far inside fp32
    float colf[8], rowf1[8], rowf2[8];
};

// exact integer milli-luminance: 299 R + 587 G + 114 B  (255 + 44, 255 + 255 + 77, 114)
__device__ __forceinline__ double lum_milli(uint32_t p)
{
    uint32_t i = __builtin_amdgcn_udot4(p, 0x00004d00u, 0u, false);                                      // (0, 77, 0, 0)
    i = __builtin_amdgcn_udot4(p, 0x0000ff2cu, i, false);                                               // (44, 255, 0, 0)
    i = __builtin_amdgcn_udot4(p, 0x0072ffffu, i, false);                                               // (255, 255, 114, 0)
    return u8_to_f64(i);
}

// fixed-order sum of `tiles` partial sums by a 256-lane workgroup (valid in thread 0): 8 independent partial sums
// per lane keep 8 loads in flight (a single chain waits out every load: 60 us for the 65 k tiles of an 8K SSIM);
// the combination order is fixed, so results stay bit-reproducible from run to run
__device__ __forceinline__ double finish_sum_256(const double *p, int tiles, double *s_red)
{
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int i = threadIdx.x;
    for (; i + 7 * 256 < tiles; i += 8 * 256) {
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] += p[i + e * 256];
    }
    for (int e = 0; i < tiles; i += 256, e++) acc[e] += p[i];
    const double v = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    return block_sum_256(v, s_red);
}

// The end of a marching wave: its sum into the image's partial sums and -- out != nullptr -- the folded finish.  Every
// wave of the workgroup calls it (val is ignored when the wave had no item).
__device__ __forceinline__ void march_finish(const MarchArgs &a, int z, int item, int items, double val, double *s_red, int *s_last)
{
    const int lane = threadIdx.x & 63;
    if (item < items) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) val += __shfl_down(val, off, 64);
        if (lane == 0) {
            double *pp = a.partial + static_cast<size_t>(z) * items + item;
            if (a.out) {
                // write-through store + drain instead of an agent-scope release: the release is an L2 write-back
                // (buffer_wbl2) per workgroup and cost 40 % of the launch
                asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" ::"v"(pp), "v"(val) : "memory");
            } else {
                *pp = val;
            }
        }
    }
    if (a.out) {
        // partial sums written through, then the counter; whoever moves it to the last value reads them all back
        // behind an agent-scope acquire (other CUs', other XCDs' stores: MI355X_MICROARCH "inter-workgroup visibility")
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned prev = __hip_atomic_fetch_add(&a.done[z], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *s_last = prev == gridDim.x - 1;
        }
        __syncthreads();
        if (*s_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const double t = finish_sum_256(a.partial + static_cast<size_t>(z) * items, items, s_red);
            if (threadIdx.x == 0) {
                a.out[z] = a.count > 0 ? t / a.count : 1.0;       // totalCount==0 -> 1.0 (ssim.go:162-164)
                __hip_atomic_store(&a.done[z], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// SMALL = true: built for <= 96 VGPRs (a few spills: 150 against 148 us per 8K pair when it runs alone).  The one-pass
// GaussianBlur + SSIMFast step launches it as its tail, under the NEXT step's blur, whose four waves per SIMD leave
// 96 registers free: a wave of this size becomes resident BESIDE them, a 124-register one only in place of one.
template <bool SMALL>
__global__ __launch_bounds__(256, SMALL ? 5 : 1) void windowed_ssim_march_kernel(MarchArgs a)
{
    // per wave: [0, WM_LDSW) (a, b) pairs, [WM_LDSW, 2 WM_LDSW) (a^2 + b^2, ab) pairs
    __shared__ __attribute__((aligned(16))) double2 s_row[4][2 * WM_LDSW];
    __shared__ double s_red[4];
    __shared__ int s_last;
    const int z = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + wave;
    const int items = a.strips * a.segs;
    double val = 0.0;
    if (item < items) {                                           // wave-uniform (a wave without an item still joins the finish)
    // adjacent waves take adjacent strips of one segment: a workgroup reads 4 x 57 contiguous columns
    const int seg = item / a.strips, strip = item - seg * a.strips;
    const int ww = a.w - 8, wh = a.h - 8;
    const int wy0 = seg * a.seg_rows;
    const int nwin = min(a.seg_rows, wh - wy0);                  // window rows of this segment
    const int wx = strip * WM_COLS + lane;
    const bool live = lane < WM_COLS && wx < ww;
    const int px = min(wx, a.w - 1);
    const uint8_t *pa = batch_img(a.a, a.as, a.a_image_bytes, z) + static_cast<size_t>(wy0) * a.astride + 4 * static_cast<size_t>(px);
    const uint8_t *pb = batch_img(a.b, a.bs, a.b_image_bytes, z) + static_cast<size_t>(wy0) * a.bstride + 4 * static_cast<size_t>(px);
    double2 *s_p1 = s_row[wave], *s_p2 = s_row[wave] + WM_LDSW;
    if (lane < WM_LDSW - 64) {                                    // the pad entries lanes 57..63 read: finite, unused
        s_p1[64 + lane] = make_double2(0.0, 0.0);
        s_p2[64 + lane] = make_double2(0.0, 0.0);
    }
    const int nrows = nwin + 7;                                   // pixel rows of the segment
    double m[8][4];                                               // ring: window slot x {E[a], E[b], E[a^2 + b^2], E[ab]}
#pragma unroll
    for (int s = 0; s < 8; s++) m[s][0] = m[s][1] = m[s][2] = m[s][3] = 0.0;
    // pixel rows are fetched WM_PF rows ahead (one row of work is ~450 clocks of issue, an HBM miss ~900):
    // slot (i mod WM_PF) holds row i
    uint32_t qa[WM_PF], qb[WM_PF];
#pragma unroll
    for (int k = 0; k < WM_PF; k++) {
        const int rr = min(k, nrows - 1);
        qa[k] = *(g_u32 *)(pa + static_cast<size_t>(rr) * a.astride);
        qb[k] = *(g_u32 *)(pb + static_cast<size_t>(rr) * a.bstride);
    }
    pa += static_cast<size_t>(min(WM_PF, nrows - 1)) * a.astride;   // -> row i + WM_PF (clamped to the segment)
    pb += static_cast<size_t>(min(WM_PF, nrows - 1)) * a.bstride;
    constexpr double C1 = 6.5025e6, C2 = 58.5225e6;              // (0.01 * 255)^2, (0.03 * 255)^2 in milli-luminance^2

    for (int r = 0; r < nrows; r += 8) {
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const int i = r + p;
            if (i < nrows) {                                      // wave-uniform
                const double va = lum_milli(qa[p % WM_PF]), vb = lum_milli(qb[p % WM_PF]);
                // no branch around these loads (the compiler would wait for vmcnt(0) every row and lose the
                // prefetch): past the segment's last row the pointers simply stop advancing
                qa[p % WM_PF] = *(g_u32 *)pa;
                qb[p % WM_PF] = *(g_u32 *)pb;
                if (i + WM_PF + 1 < nrows) {
                    pa += a.astride;
                    pb += a.bstride;
                }
                s_p1[lane] = make_double2(va, vb);
                s_p2[lane] = make_double2(fma(vb, vb, va * va), va * vb);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                double h0 = 0.0, h1 = 0.0, h2 = 0.0, h3 = 0.0;
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    const double2 u = s_p1[lane + t], v = s_p2[lane + t];
                    const double c = a.col[t];
                    h0 = fma(u.x, c, h0);
                    h1 = fma(u.y, c, h1);
                    h2 = fma(v.x, c, h2);
                    h3 = fma(v.y, c, h3);
                }
                __builtin_amdgcn_wave_barrier();                  // the row is consumed before the next one overwrites it
                // vertical taps: slot s holds the window that started at row i - k, k = (p - s) mod 8
#pragma unroll
                for (int s = 0; s < 8; s++) {
                    const int k = (p - s) & 7;
                    const double rk = a.row[k];
                    if (k == 0) {
                        m[s][0] = h0 * rk; m[s][1] = h1 * rk; m[s][2] = h2 * rk; m[s][3] = h3 * rk;
                    } else {
                        m[s][0] = fma(h0, rk, m[s][0]); m[s][1] = fma(h1, rk, m[s][1]);
                        m[s][2] = fma(h2, rk, m[s][2]); m[s][3] = fma(h3, rk, m[s][3]);
                    }
                }
                if (i >= 7) {                                     // slot (p + 1) mod 8 just took its 8th row: window row wy0 + i - 7
                    const int s = (p + 1) & 7;
                    const double muA = m[s][0], muB = m[s][1];
                    const double mu2 = fma(muB, muB, muA * muA), muAB = muA * muB;
                    const double sSum = m[s][2] - mu2, sAB = m[s][3] - muAB;   // sigma_aa + sigma_bb, sigma_ab
                    const double num = fma(2.0, muAB, C1) * fma(2.0, sAB, C2);
                    const double den = (mu2 + C1) * (sSum + C2);
                    double rc = __builtin_amdgcn_rcp(den);        // den >= C1 * C2 > 0: rcp + 2 Newton steps ~ 1 ulp
                    rc = fma(fma(-den, rc, 1.0), rc, rc);
                    rc = fma(fma(-den, rc, 1.0), rc, rc);
                    val = fma(num, rc, val);
                }
            }
        }
    }
    if (!live) val = 0.0;
    }
    march_finish(a, z, item, items, val, s_red, &s_last);
}

// ------------------------------------------------------------------------------------
// The same march with TWO pixel columns per lane -- what the biggest planes take.
//
// The one-column kernel is bound by LDS bandwidth, not by its FMAs: per pixel row a lane reads 8 taps x 2 entries of
// 16 bytes (256 B) and writes 32 B, 64 lanes x 288 B = 18 KB per wave-row at 128 B per clock per CU -> 144 clocks,
// x 16 resident waves = 2300 clocks per row round against 4 x 94 x 4 = 1500 clocks of VALU issue per SIMD.  Two
// ADJACENT columns share 7 of their 8 taps: a lane that owns pixels 2l and 2l + 1 reads 9 entries per array instead of
// 16 (and a wave covers 128 pixel columns, so its 7 idle window columns are 5 % of the lanes' work instead of 11 %):
// 176 LDS bytes per window instead of 288.  Even and odd pixels live in separate arrays so that every read is
// 16 bytes at a 16-byte lane stride (one array at a 32-byte stride would be 2-way bank conflicts).  The rings of both
// columns are 128 VGPRs, so this form runs 2 waves per SIMD -- the same windows in flight per SIMD as 4 one-column
// waves, with the instruction-level parallelism inside the wave instead of between waves.
// ------------------------------------------------------------------------------------
// r5, measured and NOT taken: the milli-luminances are integers below 2^24 -- exact in fp32 -- so a lane's pixel pair can go
// through the LDS as ONE float4 (a0, b0, a1, b1) and come back as five 16-byte reads instead of nine (17 LDS instructions per
// wave-row instead of 22, 18 more conversions).  The model that asked for it -- 22 x 8 clocks x 8 resident waves = 1408 clocks
// of the CU's one LDS pipe per row round against 1288 of VALU issue per SIMD -- was wrong: 135 us per 8K pair against 131.
// The bound is a wave's chain write -> reads -> FMAs (the LDS round trip it cannot hide from itself), not the pipe.
constexpr int WM2_COLS = 121;        // window columns per wave (128 pixel columns - 7)
constexpr int WM2_LDSW = 72;         // entries per parity array: lane + 4, padded

__global__ __launch_bounds__(256, 2) void windowed_ssim_march2_kernel(MarchArgs a)
{
    // per wave, four arrays of WM2_LDSW: even (a, b), odd (a, b), even (a^2 + b^2, ab), odd (a^2 + b^2, ab)
    __shared__ __attribute__((aligned(16))) double2 s_row[4][4 * WM2_LDSW];
    __shared__ double s_red[4];
    __shared__ int s_last;
    const int z = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + wave;
    const int items = a.strips * a.segs;
    double val = 0.0;
    if (item < items) {                                           // wave-uniform
    const int seg = item / a.strips, strip = item - seg * a.strips;
    const int ww = a.w - 8, wh = a.h - 8;
    const int wy0 = seg * a.seg_rows;
    const int nwin = min(a.seg_rows, wh - wy0);
    const int wx = strip * WM2_COLS + 2 * lane;                   // window (= pixel) column of this lane's first column
    const bool live0 = 2 * lane < WM2_COLS && wx < ww, live1 = 2 * lane + 1 < WM2_COLS && wx + 1 < ww;
    // the pixel pair (px, px + 1), one 8-byte load (4-byte aligned).  Columns past w - 2 are taps of no live window
    // (the image's last column is never sampled, ssim.go:110-111): such lanes re-read the last pair
    const int px = min(wx, a.w - 2);
    const uint8_t *pa = batch_img(a.a, a.as, a.a_image_bytes, z) + static_cast<size_t>(wy0) * a.astride + 4 * static_cast<size_t>(px);
    const uint8_t *pb = batch_img(a.b, a.bs, a.b_image_bytes, z) + static_cast<size_t>(wy0) * a.bstride + 4 * static_cast<size_t>(px);
    double2 *s_e1 = s_row[wave], *s_o1 = s_e1 + WM2_LDSW, *s_e2 = s_o1 + WM2_LDSW, *s_o2 = s_e2 + WM2_LDSW;
    if (lane < WM2_LDSW - 64) {                                   // the entries lanes 60..63 read past lane 63: finite, unused
        s_e1[64 + lane] = make_double2(0.0, 0.0); s_o1[64 + lane] = make_double2(0.0, 0.0);
        s_e2[64 + lane] = make_double2(0.0, 0.0); s_o2[64 + lane] = make_double2(0.0, 0.0);
    }
    const int nrows = nwin + 7;
    double m0[8][4], m1[8][4];                                    // the two columns' rings (see the one-column kernel)
#pragma unroll
    for (int s = 0; s < 8; s++) {
        m0[s][0] = m0[s][1] = m0[s][2] = m0[s][3] = 0.0;
        m1[s][0] = m1[s][1] = m1[s][2] = m1[s][3] = 0.0;
    }
    double val0 = 0.0, val1 = 0.0;
    double P0 = 0.0, Q0 = 1.0, P1 = 0.0, Q1 = 1.0;                 // the fraction of the windows not yet divided
    typedef __attribute__((address_space(1))) const u32x2 g_u32x2;
    u32x2 qa[WM_PF], qb[WM_PF];
#pragma unroll
    for (int k = 0; k < WM_PF; k++) {
        const int rr = min(k, nrows - 1);
        qa[k] = *(g_u32x2 *)(pa + static_cast<size_t>(rr) * a.astride);
        qb[k] = *(g_u32x2 *)(pb + static_cast<size_t>(rr) * a.bstride);
    }
    pa += static_cast<size_t>(min(WM_PF, nrows - 1)) * a.astride;
    pb += static_cast<size_t>(min(WM_PF, nrows - 1)) * a.bstride;
    constexpr double C1 = 6.5025e6, C2 = 58.5225e6;

    // The two waves of a SIMD are from two workgroups, and the issue arbiter prefers the OLDER wave every cycle: per-wave
    // timestamps showed the first-dispatched half of an 8K launch finishing at 101 us and the second at 148 us --
    // for a third of the launch every SIMD held ONE wave, which cannot hide its own LDS latency.  Priority by progress
    // instead: a wave raises its priority on alternate groups of 8 rows, the SIMD's other wave (the other wave slot)
    // on the groups in between; whoever runs ahead reaches its low-priority group sooner.  Measured: 114 / 142 us and
    // the launch 139 -> 136 us -- the LDS queue still serves the older wave first, and a SIMD delivers about the same
    // rows per microsecond with one wave as with two (the bound is the chain write -> reads -> FMAs, not the arbiter).
    const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1u;   // HW_REG_HW_ID.WAVE_ID bit 0
    // One row of the march.  Groups of eight full rows run WITHOUT a branch around the row (GUARD = false): with the
    // wave-uniform "if (i < nrows)" around every unrolled row the compiler cannot tell how many loads are in flight at the
    // joins and drains the queue (s_waitcnt vmcnt(0)) before each row's taps -- the four rows of prefetch were none
    // (8K pair: 135 -> 131 us; the one-column kernel keeps its loop: the same change costs its 96-register form 22 spills).
    // r5: the march is software-pipelined by one row.  A row's luminances and its four LDS writes (with the wave barrier) are
    // issued BEFORE the previous row's vertical pass and SSIM formula (~95 VALU instructions, which need none of it) instead of
    // in front of its own reads: the row is in the LDS by the time its horizontal pass asks for it.  8K pair 134.6 -> 126 us (same
    // box).  Sending the reads ahead too did not pay: all eighteen took 72 registers (256 with 62 spills), the nine second-moment
    // taps alone (36) measured the same 126-127 us.  The loop is 145 VALU instructions per wave-row, 128 of them the separable
    // window's FMAs: what is left is the fp64 rate.
    auto stage = [&](const int i, auto pc) {                      // row i (i & 7 == pc): luminances into the wave's LDS row
        constexpr int p = decltype(pc)::value;
        const double va0 = lum_milli(qa[p % WM_PF][0]), va1 = lum_milli(qa[p % WM_PF][1]);
        const double vb0 = lum_milli(qb[p % WM_PF][0]), vb1 = lum_milli(qb[p % WM_PF][1]);
        qa[p % WM_PF] = *(g_u32x2 *)pa;                       // no branch around the loads (see the one-column kernel)
        qb[p % WM_PF] = *(g_u32x2 *)pb;
        if (i + WM_PF + 1 < nrows) {
            pa += a.astride;
            pb += a.bstride;
        }
        s_e1[lane] = make_double2(va0, vb0);
        s_o1[lane] = make_double2(va1, vb1);
        s_e2[lane] = make_double2(fma(vb0, vb0, va0 * va0), va0 * vb0);
        s_o2[lane] = make_double2(fma(vb1, vb1, va1 * va1), va1 * vb1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto rowbody = [&](const int r, auto pc, auto guardc) {
        constexpr int p = decltype(pc)::value;
        const int i = r + p;
        if (decltype(guardc)::value && i >= nrows) return;        // wave-uniform
            // pixel 2l + t: even t -> even[l + t / 2], odd t -> odd[l + t / 2].  Column 0's tap t is pixel 2l + t,
            // column 1's tap t is pixel 2l + 1 + t.
            double h00 = 0.0, h01 = 0.0, h02 = 0.0, h03 = 0.0, h10 = 0.0, h11 = 0.0, h12 = 0.0, h13 = 0.0;
            double2 ru[9];
#pragma unroll
            for (int q = 0; q < 9; q++) ru[q] = (q & 1) ? s_o1[lane + q / 2] : s_e1[lane + q / 2];
#pragma unroll
            for (int q = 0; q < 9; q++) {                     // pixel 2l + q: the second moments
                const double2 v = (q & 1) ? s_o2[lane + q / 2] : s_e2[lane + q / 2];
                if (q < 8) { const double c = a.col[q]; h02 = fma(v.x, c, h02); h03 = fma(v.y, c, h03); }
                if (q >= 1) { const double c = a.col[q - 1]; h12 = fma(v.x, c, h12); h13 = fma(v.y, c, h13); }
            }
#pragma unroll
            for (int q = 0; q < 9; q++) {                     // the first moments
                const double2 u = ru[q];
                if (q < 8) { const double c = a.col[q]; h00 = fma(u.x, c, h00); h01 = fma(u.y, c, h01); }
                if (q >= 1) { const double c = a.col[q - 1]; h10 = fma(u.x, c, h10); h11 = fma(u.y, c, h11); }
            }
            // (this row's reads have been waited for by the FMAs above: the next row may overwrite it -- one wave, in-order LDS)
            __builtin_amdgcn_wave_barrier();
            stage(i + 1, std::integral_constant<int, (p + 1) & 7>{});   // (past the last row: the clamped loads again, never consumed)
#pragma unroll
            for (int s = 0; s < 8; s++) {
                const int k = (p - s) & 7;
                const double rk = a.row[k];
                if (k == 0) {
                    m0[s][0] = h00 * rk; m0[s][1] = h01 * rk; m0[s][2] = h02 * rk; m0[s][3] = h03 * rk;
                    m1[s][0] = h10 * rk; m1[s][1] = h11 * rk; m1[s][2] = h12 * rk; m1[s][3] = h13 * rk;
                } else {
                    m0[s][0] = fma(h00, rk, m0[s][0]); m0[s][1] = fma(h01, rk, m0[s][1]);
                    m0[s][2] = fma(h02, rk, m0[s][2]); m0[s][3] = fma(h03, rk, m0[s][3]);
                    m1[s][0] = fma(h10, rk, m1[s][0]); m1[s][1] = fma(h11, rk, m1[s][1]);
                    m1[s][2] = fma(h12, rk, m1[s][2]); m1[s][3] = fma(h13, rk, m1[s][3]);
                }
            }
            if (i >= 7) {
                const int s = (p + 1) & 7;
                // numerator and denominator of one window's SSIM (ssim.go:150-155, scaled by 1e6 twice)
                auto numden = [&](const double (&mm)[4], double &num, double &den) {
                    const double muA = mm[0], muB = mm[1];
                    const double mu2 = fma(muB, muB, muA * muA), muAB = muA * muB;
                    const double sSum = mm[2] - mu2, sAB = mm[3] - muAB;
                    num = fma(2.0, muAB, C1) * fma(2.0, sAB, C2);
                    den = (mu2 + C1) * (sSum + C2);
                };
                auto quot = [&](double num, double den) {
                    double rc = __builtin_amdgcn_rcp(den);
                    rc = fma(fma(-den, rc, 1.0), rc, rc);
                    rc = fma(fma(-den, rc, 1.0), rc, rc);
                    return num * rc;
                };
                // ONE division per four windows of a column: n1/d1 + ... + n4/d4 as a single fraction, built up as
                // P <- P d + n Q, Q <- Q d (den <= 1.7e22, so Q stays below 1e89; every term is positive and the
                // whole is a few ulp from the sum of the four quotients -- the bar is 1e-9).  v_rcp_f64 and its two
                // Newton steps were 9 of the 21 instruction slots a window's score took.
                double n0, d0, n1, d1;
                numden(m0[s], n0, d0);
                numden(m1[s], n1, d1);
                if (r == 0) {                                 // the first ring fill: only p == 7 has a complete window
                    val0 += quot(n0, d0);
                    val1 += quot(n1, d1);
                } else {
                    if ((p & 3) == 0) {
                        P0 = n0; Q0 = d0; P1 = n1; Q1 = d1;
                    } else {
                        P0 = fma(P0, d0, n0 * Q0); Q0 *= d0;
                        P1 = fma(P1, d1, n1 * Q1); Q1 *= d1;
                    }
                    if ((p & 3) == 3 || i == nrows - 1) {     // wave-uniform
                        val0 += quot(P0, Q0);
                        val1 += quot(P1, Q1);
                    }
                }
            }
    };
    auto group = [&](const int r, auto guardc) {
        if (a.prio && (((r >> 3) ^ slot) & 1u) != 0) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
        rowbody(r, std::integral_constant<int, 0>{}, guardc); rowbody(r, std::integral_constant<int, 1>{}, guardc);
        rowbody(r, std::integral_constant<int, 2>{}, guardc); rowbody(r, std::integral_constant<int, 3>{}, guardc);
        rowbody(r, std::integral_constant<int, 4>{}, guardc); rowbody(r, std::integral_constant<int, 5>{}, guardc);
        rowbody(r, std::integral_constant<int, 6>{}, guardc); rowbody(r, std::integral_constant<int, 7>{}, guardc);
    };
    stage(0, std::integral_constant<int, 0>{});
    int r = 0;
    for (; r + 8 <= nrows; r += 8) group(r, std::false_type{});
    if (r < nrows) group(r, std::true_type{});
    val = (live0 ? val0 : 0.0) + (live1 ? val1 : 0.0);
    }
    march_finish(a, z, item, items, val, s_red, &s_last);
}

// ------------------------------------------------------------------------------------
// r6: the two-column march with fp32 moments -- the FAST mode of full-resolution SSIM (fnx_ctx_set_ssim_mode(FNX_SSIM_FAST);
// "<= 1e-6 for fp32-moment fast paths"; the default stays the fp64 kernel above, <= 1e-9).
//
// fp64 FMAs issue at the plain fp32 rate on this part and v_pk_fma_f32 does two fp32 FMAs per lane in the same slot, so the
// window's 64 FMAs per pixel become 32 instructions.  What makes fp32 ENOUGH is the formulation, not the type:
//   * the moments are taken of x = A - 127500, y = B - 127500 (exact integers in fp32) and of d = x - y (exact);
//   * with  S = E[x^2 + y^2] - mx^2 - my^2 (= sigma_aa + sigma_bb),  Vd = E[d^2] - (mx - my)^2 (= Var(a - b))  the window's
//     value (ssim.go:150-155) is   [1 - (mx - my)^2 / (muA^2 + muB^2 + C1)] * [1 - Vd / (S + C2)]   -- identically, since
//     2 sigma_ab = S - Vd and 2 muA muB = muA^2 + muB^2 - (muA - muB)^2.  The cancellation-prone S (an fp32 E[x^2] minus
//     an fp32 mean^2: absolute error ~1e-7 E[x^2]) now only scales a term that VANISHES where the images agree: its error
//     reaches the result multiplied by Vd / (S + C2)^2.  Identical images give exactly 1, as in the reference.
// Measured against the CPU restatement on SURVEY's ramp, photograph-like, noise, dark, bright, flat and unrelated pairs:
// |delta| <= 6e-8 on the image's mean with the per-wave centre below (tests/test_gpu_parity.py::test_ssim_fast_moments); the
// same expression in fp64 is within 1e-13.
// Layout: lane l owns pixels 2l, 2l + 1 (as the fp64 kernel); an LDS entry is ONE float4 per pixel (x, y, x^2 + y^2, d^2) in
// an even and an odd array (16-byte reads at a 16-byte lane stride); (x, y) and (x^2 + y^2, d^2) are the packed pairs of
// every FMA, so no operand is ever shuffled.  Rings: 2 columns x 8 windows x 2 pairs = 64 VGPRs: three waves per SIMD.
// Two consecutive windows of a column share ONE division (P / Q as in the fp64 kernel; denominators <= 4e9 in these units).
// The lane's sum is kept in fp32 over eight rows of (1 - value) and added into an fp64 total per group.
// ------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// (299 R + 587 G + 114 B) - c as an exact fp32 integer: the dot products start from -c mod 2^32 (negc)
__device__ __forceinline__ float lum_milli_centred(uint32_t p, uint32_t negc)
{
    uint32_t i = __builtin_amdgcn_udot4(p, 0x00004d00u, negc, false);
    i = __builtin_amdgcn_udot4(p, 0x0000ff2cu, i, false);
    i = __builtin_amdgcn_udot4(p, 0x0072ffffu, i, false);
    return static_cast<float>(static_cast<int32_t>(i));
}

constexpr int WMF_PF = 4;              // pixel rows in flight per lane (divides 8; 2 and 8 measured the same)
template <int WPS, bool PRE>
__global__ __launch_bounds__(256, WPS) void windowed_ssim_march2f_kernel(MarchArgs a)
{
    __shared__ __attribute__((aligned(16))) f32x4 s_row[4][2 * WM2_LDSW];   // per wave: even pixels, odd pixels
    __shared__ double s_red[4];
    __shared__ int s_last;
    const int z = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + wave;
    const int items = a.strips * a.segs;
    double val = 0.0;
    if (item < items) {                                           // wave-uniform
    const int seg = item / a.strips, strip = item - seg * a.strips;
    const int ww = a.w - 8, wh = a.h - 8;
    const int wy0 = seg * a.seg_rows;
    const int nwin = min(a.seg_rows, wh - wy0);
    const int wx = strip * WM2_COLS + 2 * lane;
    const bool live0 = 2 * lane < WM2_COLS && wx < ww, live1 = 2 * lane + 1 < WM2_COLS && wx + 1 < ww;
    const int px = min(wx, a.w - 2);                              // (see the fp64 kernel)
    const uint8_t *pa = batch_img(a.a, a.as, a.a_image_bytes, z) + static_cast<size_t>(wy0) * a.astride + 4 * static_cast<size_t>(px);
    const uint8_t *pb = batch_img(a.b, a.bs, a.b_image_bytes, z) + static_cast<size_t>(wy0) * a.bstride + 4 * static_cast<size_t>(px);
    f32x4 *s_e = s_row[wave], *s_o = s_e + WM2_LDSW;
    if (lane < WM2_LDSW - 64) {
        s_e[64 + lane] = (f32x4){0.f, 0.f, 0.f, 0.f};
        s_o[64 + lane] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int nrows = nwin + 7;
    // rings, each entry the pair (column 0, column 1): mx, my, E[x^2 + y^2], E[d^2].  The horizontal pass pairs QUANTITIES
    // (an LDS entry is (x, y, x^2 + y^2, d^2) of one pixel); its four results are transposed once per row (v_pk_mov) so that
    // the vertical pass and the whole window formula run on column pairs with no operand shuffled again.
    f32x2 rmx[8], rmy[8], rms[8], rmd[8];
#pragma unroll
    for (int s = 0; s < 8; s++) rmx[s] = rmy[s] = rms[s] = rmd[s] = (f32x2){0.f, 0.f};
    typedef __attribute__((address_space(1))) const u32x2 g_u32x2;
    // The centre c: the mean milli-luminance of three rows of both images as this WAVE sees them (first, middle and last
    // pixel row of its segment, 128 columns), an integer.  Any constant gives the same moments in exact arithmetic; one near
    // the strip's level keeps |x|, |y| -- and with them the rounding of E[x^2 + y^2] against mx^2 + my^2, the one
    // cancellation the form has left -- as small as the content allows (a bright flat region against a noisy copy of itself
    // was 3.6e-6 off with the fixed centre 127500 and is exact to 1e-8 with this one).
    uint32_t csum = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const size_t rr = static_cast<size_t>(k == 0 ? 0 : (k == 1 ? (nrows - 1) / 2 : nrows - 1));
        const u32x2 sa = *(g_u32x2 *)(pa + rr * a.astride), sb = *(g_u32x2 *)(pb + rr * a.bstride);
#pragma unroll
        for (int e = 0; e < 2; e++) {
            csum = __builtin_amdgcn_udot4(sa[e], 0x00004d00u, csum, false); csum = __builtin_amdgcn_udot4(sa[e], 0x0000ff2cu, csum, false);
            csum = __builtin_amdgcn_udot4(sa[e], 0x0072ffffu, csum, false);
            csum = __builtin_amdgcn_udot4(sb[e], 0x00004d00u, csum, false); csum = __builtin_amdgcn_udot4(sb[e], 0x0000ff2cu, csum, false);
            csum = __builtin_amdgcn_udot4(sb[e], 0x0072ffffu, csum, false);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) csum += __shfl_xor(csum, off, 64);          // <= 768 x 255000 < 2^28
    const uint32_t centre = __builtin_amdgcn_readfirstlane(csum) / 768u;
    const uint32_t negc = 0u - centre;
    u32x2 qa[WMF_PF], qb[WMF_PF];
#pragma unroll
    for (int k = 0; k < WMF_PF; k++) {
        const int rr = min(k, nrows - 1);
        qa[k] = *(g_u32x2 *)(pa + static_cast<size_t>(rr) * a.astride);
        qb[k] = *(g_u32x2 *)(pb + static_cast<size_t>(rr) * a.bstride);
    }
    pa += static_cast<size_t>(min(WMF_PF, nrows - 1)) * a.astride;
    pb += static_cast<size_t>(min(WMF_PF, nrows - 1)) * a.bstride;
    // units: milli-luminance / 1024
    const float CU = static_cast<float>(centre) * (1.0f / 1024.0f);    // exact: an integer below 2^18 times 2^-10
    constexpr float C1U = 6.5025e6f / 1048576.0f, C2U = 58.5225e6f / 1048576.0f;
    f32x2 acc = {0.f, 0.f};                                       // sum of (1 - value) over the current group of rows, per column
    double tot0 = 0.0, tot1 = 0.0;
    f32x2 PP = {0.f, 0.f}, QQ = {1.f, 1.f};                       // the windows of the row before, not yet divided

    auto stage = [&](const int i, auto pc) {
        constexpr int p = decltype(pc)::value;
        const float x0 = lum_milli_centred(qa[p % WMF_PF][0], negc), x1 = lum_milli_centred(qa[p % WMF_PF][1], negc);
        const float y0 = lum_milli_centred(qb[p % WMF_PF][0], negc), y1 = lum_milli_centred(qb[p % WMF_PF][1], negc);
        qa[p % WMF_PF] = *(g_u32x2 *)pa;                       // no branch around the loads (see the one-column kernel)
        qb[p % WMF_PF] = *(g_u32x2 *)pb;
        if (i + WMF_PF + 1 < nrows) {
            pa += a.astride;
            pb += a.bstride;
        }
        const float d0 = x0 - y0, d1 = x1 - y1;
        s_e[lane] = (f32x4){x0, y0, fmaf(y0, y0, x0 * x0), d0 * d0};
        s_o[lane] = (f32x4){x1, y1, fmaf(y1, y1, x1 * x1), d1 * d1};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    f32x4 tap[9];                                                 // the staged row as this lane's window sees it: pixels 2l .. 2l + 8
    auto fetch = [&]() {
#pragma unroll
        for (int q = 0; q < 9; q++) tap[q] = (q & 1) ? s_o[lane + q / 2] : s_e[lane + q / 2];
        __builtin_amdgcn_wave_barrier();                          // (the next stage() overwrites the row: after these reads, in order)
    };
    // numerator and denominator of (1 - value) of the two columns' windows: value = (t1 - md2)(den2 - vd) / (t1 den2)
    auto numden = [&](const f32x2 MX, const f32x2 MY, const f32x2 M2, const f32x2 M3, f32x2 &num, f32x2 &den) {
        const f32x2 cu = {CU, CU}, c1 = {C1U, C1U}, c2 = {C2U, C2U};
        const f32x2 sq = __builtin_elementwise_fma(MY, MY, MX * MX);              // mx^2 + my^2
        const f32x2 ma = MX + cu, mb = MY + cu;
        const f32x2 t1 = __builtin_elementwise_fma(ma, ma, __builtin_elementwise_fma(mb, mb, c1));
        const f32x2 md = MX - MY;
        const f32x2 md2 = md * md;
        const f32x2 den2 = (M2 + c2) - sq;
        const f32x2 vd = M3 - md2;
        // 1 - (t1 - md2)(den2 - vd) / (t1 den2) = (md2 den2 + vd (t1 - md2)) / (t1 den2): no cancellation left in the quotient
        num = __builtin_elementwise_fma(md2, den2, vd * (t1 - md2));
        den = t1 * den2;
    };
    auto quot = [&](const f32x2 num, const f32x2 den) {
        f32x2 rc = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
        const f32x2 one = {1.0f, 1.0f};
        rc = __builtin_elementwise_fma(__builtin_elementwise_fma(-den, rc, one), rc, rc);
        return num * rc;
    };
    // FIRST: the segment's first eight rows (the ring fills: only p == 7 completes a window; nrows >= 8 always).  GUARD: the
    // segment's last, partial group.  The groups in between are straight-line code without a branch (see the fp64 kernel).
    auto rowbody = [&](const int r, auto pc, auto guardc, auto firstc) {
        constexpr int p = decltype(pc)::value;
        constexpr bool GUARD = decltype(guardc)::value, FIRST = decltype(firstc)::value;
        const int i = r + p;
        if (GUARD && i >= nrows) return;                          // wave-uniform
        // rows stay rows: without this fence the scheduler pulls the luminances of all WMF_PF prefetched rows to the front of
        // the group and waits for the NEWEST load (s_waitcnt vmcnt(0) once per row: the four rows of prefetch become one)
        __builtin_amdgcn_sched_barrier(0);
        // the row's nine entries were read at the END of the row before (tap[]): the LDS round trip runs under that row's
        // vertical pass and formula instead of in front of this row's FMAs (36 registers; the fp64 kernel has no room for it)
        f32x2 h0a = {0.f, 0.f}, h0q = {0.f, 0.f}, h1a = {0.f, 0.f}, h1q = {0.f, 0.f};
        if (!PRE) fetch();
#pragma unroll
        for (int q = 0; q < 9; q++) {                             // pixel 2l + q: column 0's tap q, column 1's tap q - 1
            const f32x4 v = tap[q];
            const f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
            if (q < 8) {
                const f32x2 c = {a.colf[q], a.colf[q]};
                h0a = __builtin_elementwise_fma(lo, c, h0a);
                h0q = __builtin_elementwise_fma(hi, c, h0q);
            }
            if (q >= 1) {
                const f32x2 c = {a.colf[q - 1], a.colf[q - 1]};
                h1a = __builtin_elementwise_fma(lo, c, h1a);
                h1q = __builtin_elementwise_fma(hi, c, h1q);
            }
        }
        stage(i + 1, std::integral_constant<int, (p + 1) & 7>{});   // (tap[] holds row i: the LDS row is free)
        if (PRE) fetch();
        // quantity pairs -> column pairs (two v_pk_mov_b32 per 2 x 2 transpose)
        f32x2 hx, hy, hs, hd;
        asm("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0]" : "=v"(hx) : "v"(h0a), "v"(h1a));      // (h0a.x, h1a.x)
        asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(hy) : "v"(h0a), "v"(h1a));      // (h0a.y, h1a.y)
        asm("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0]" : "=v"(hs) : "v"(h0q), "v"(h1q));
        asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(hd) : "v"(h0q), "v"(h1q));
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const int k = (p - s) & 7;
            const f32x2 r1 = {a.rowf1[k], a.rowf1[k]}, r2 = {a.rowf2[k], a.rowf2[k]};
            if (k == 0) {
                rmx[s] = hx * r1; rmy[s] = hy * r1; rms[s] = hs * r2; rmd[s] = hd * r2;
            } else {
                rmx[s] = __builtin_elementwise_fma(hx, r1, rmx[s]); rmy[s] = __builtin_elementwise_fma(hy, r1, rmy[s]);
                rms[s] = __builtin_elementwise_fma(hs, r2, rms[s]); rmd[s] = __builtin_elementwise_fma(hd, r2, rmd[s]);
            }
        }
        if (FIRST && p != 7) return;
        const int s = (p + 1) & 7;
        f32x2 nn, dd;
        numden(rmx[s], rmy[s], rms[s], rmd[s], nn, dd);
        if (FIRST) {
            acc += quot(nn, dd);
        } else if ((p & 1) == 0) {
            // (a full group ends on an odd p: only the guarded group can end on an even one)
            if (GUARD && i == nrows - 1) {                        // wave-uniform: the segment's last row has no partner
                acc += quot(nn, dd);
            } else {
                PP = nn; QQ = dd;
            }
        } else {
            acc += quot(__builtin_elementwise_fma(PP, dd, nn * QQ), QQ * dd);
        }
    };
    auto group = [&](const int r, auto guardc, auto firstc) {
        rowbody(r, std::integral_constant<int, 0>{}, guardc, firstc); rowbody(r, std::integral_constant<int, 1>{}, guardc, firstc);
        rowbody(r, std::integral_constant<int, 2>{}, guardc, firstc); rowbody(r, std::integral_constant<int, 3>{}, guardc, firstc);
        rowbody(r, std::integral_constant<int, 4>{}, guardc, firstc); rowbody(r, std::integral_constant<int, 5>{}, guardc, firstc);
        rowbody(r, std::integral_constant<int, 6>{}, guardc, firstc); rowbody(r, std::integral_constant<int, 7>{}, guardc, firstc);
        tot0 += static_cast<double>(acc.x); tot1 += static_cast<double>(acc.y);
        acc = (f32x2){0.f, 0.f};
    };
    stage(0, std::integral_constant<int, 0>{});
    if (PRE) fetch();
    group(0, std::false_type{}, std::true_type{});
    int r = 8;
    for (; r + 8 <= nrows; r += 8) group(r, std::false_type{}, std::false_type{});
    if (r < nrows) group(r, std::true_type{}, std::false_type{});
    // the lane's windows: nwin each of value 1 - (1 - value)
    val = (live0 ? static_cast<double>(nwin) - tot0 : 0.0) + (live1 ? static_cast<double>(nwin) - tot1 : 0.0);
    }
    march_finish(a, z, item, items, val, s_red, &s_last);
}

// one workgroup per image pair: fixed-order sum of the tile partials, then / count
__global__ __launch_bounds__(256) void ssim_finish_kernel(const double *partial, int tiles, double count, double *out)
{
    __shared__ double s_red[4];
    const double t = finish_sum_256(partial + static_cast<size_t>(blockIdx.x) * tiles, tiles, s_red);
    if (threadIdx.x == 0) out[blockIdx.x] = count > 0 ? t / count : 1.0;   // totalCount==0 -> 1.0 (ssim.go:162-164)
}

// FNX_SSIM_TILED=1 keeps the tile kernels (A/B measurements); default: the marching kernel
static bool ssim_use_tiled()
{
    static const bool v = [] { const char *e = dev_env("FNX_SSIM_TILED"); return e && e[0] == '1'; }();
    return v;
}

int launch_windowed_ssim(fnx_ctx *ctx, int n, const uint8_t *a, int astride, size_t a_image_bytes,
                         const uint8_t *b, int bstride, size_t b_image_bytes, int w, int h,
                         const double *h_window, const double *d_window, double *d_out,
                         SsimDeferred *defer, int defer_out_index, const uint8_t *const *d_as, const uint8_t *const *d_bs)
{
    const int ww = w - 8, wh = h - 8;     // window grid
    const bool have = ww > 0 && wh > 0;
    WinSepArgs sa{};
    const bool sep = have && window_rank1(h_window, sa.col, sa.row);
    // the marching kernel needs long column segments to pay for its serial row walk (~0.45 us per row per
    // wave): single small planes (one SSIMFast, the MSSSIM levels) keep the tile kernels, 2-5 us a launch
    // against ~19.  FNX_SSIM_MARCH_MIN overrides the window count from which it takes over (experiments).
    static const long march_min = [] { const char *e = dev_env("FNX_SSIM_MARCH_MIN"); return e ? atol(e) : 1500000L; }();
    const bool march = sep && !ssim_use_tiled() && static_cast<long>(ww) * wh * n >= march_min;
    // tile kernels (FNX_SSIM_TILED=1, and the 64-tap kernel for tables that are not rank-1)
    const bool big = sep && static_cast<long>(ww) * wh * n >= 4L * 1024 * ctx->num_cus;
    const int TX = sep ? WSS_TX : WS_TX, TY = sep ? (big ? W24_TY : WSS_TY) : WS_TY;
    int tiles_x = 0, tiles = 0;
    MarchArgs ma{};
    // the two-column form (2 waves per SIMD) once ONE image has a round of such waves to fill the chip with
    static const long march2_min = [] { const char *e = dev_env("FNX_SSIM_MARCH2_MIN"); return e ? atol(e) : 4000000L; }();
    // (per image: a batch of small planes -- the one-pass tail under the next blur -- keeps the one-column form, whose
    // 104-VGPR waves fit into the gaps the blur's workgroups leave; 202-VGPR waves wait for two of them to retire)
    const bool march2 = march && static_cast<long>(ww) * wh >= march2_min;
    if (march) {
        // wave-sized work items: strips of 57 (two-column form: 121) window columns x row segments.  Segments are cut
        // so that the launch holds about one resident round of waves (16 per CU; 8) but never shorter than 32 window
        // rows (row halo (S + 7) / S <= 1.22)
        const int cols = march2 ? WM2_COLS : WM_COLS;
        ma.strips = (ww + cols - 1) / cols;
        static const long m2_env = [] { const char *e = dev_env("FNX_SSIM_M2_WAVES"); return e ? atol(e) : 0L; }();   // experiments
        const long m2_per_cu = m2_env ? m2_env : 8L;                // (both two-column kernels run two waves per SIMD)
        const long target = (march2 ? m2_per_cu : 16L) * ctx->num_cus;
        long segs = target / (static_cast<long>(n) * ma.strips);
        const long max_segs = wh / 32 > 0 ? wh / 32 : 1;
        if (segs > max_segs) segs = max_segs;
        if (segs < 1) segs = 1;
        ma.seg_rows = static_cast<int>((wh + segs - 1) / segs);
        ma.segs = (wh + ma.seg_rows - 1) / ma.seg_rows;
        tiles = ma.strips * ma.segs;
    } else if (have) {
        tiles_x = (ww + TX - 1) / TX;
        tiles = tiles_x * ((wh + TY - 1) / TY);
    }
    void *part = nullptr;
    bool folded = false;
    if (defer) {
        // the caller reserved SLOT_PARTIAL for all deferred levels (growing it now would free partials that
        // earlier levels' kernels are still writing); a level that does not fit is an error, not a fallback
        if (n != 1 || defer->count >= 8 || defer->used + tiles + 2 > SSIM_DEFER_DOUBLES) {
            set_error("deferred SSIM finish: level %d does not fit the reserved partial sums (%zu + %d of %zu)",
                      defer->count, defer->used, tiles, SSIM_DEFER_DOUBLES);
            return FNX_ERR_INVALID;
        }
        FNX_TRY(scratch(ctx, SLOT_PARTIAL, sizeof(double) * SSIM_DEFER_DOUBLES, &part));   // reserved by the caller: no growth
        part = static_cast<double *>(part) + defer->used;
    } else {
        const Slot ps = ctx->partial_slot >= 0 ? static_cast<Slot>(ctx->partial_slot) : SLOT_PARTIAL;
        FNX_TRY(scratch(ctx, ps, sizeof(double) * (static_cast<size_t>(tiles) * n + 2), &part));
    }
    if (march) {
        ma.a = a; ma.b = b; ma.as = d_as; ma.bs = d_bs; ma.a_image_bytes = a_image_bytes; ma.b_image_bytes = b_image_bytes;
        ma.astride = astride; ma.bstride = bstride; ma.w = w; ma.h = h;
        ma.partial = static_cast<double *>(part);
        for (int i = 0; i < 8; i++) { ma.col[i] = sa.col[i]; ma.row[i] = sa.row[i]; }
        // the image's last workgroup takes the mean itself (no finish launch).  Counters: 2 x 4096, one half per
        // stream the ctx launches on (a scored tail on the second stream may run beside a call on the first)
        static const bool nofold = [] { const char *e = dev_env("FNX_SSIM_NOFOLD"); return e && e[0] == '1'; }();
        if (!defer && !nofold && n <= 4096) {
            unsigned *dn = nullptr;
            FNX_TRY(ssim_done_counters(ctx, &dn));
            ma.done = dn + (ctx->partial_slot >= 0 ? 4096 : 0);
            ma.out = d_out;
            ma.count = have ? static_cast<double>(ww) * static_cast<double>(wh) : 0.0;
            folded = true;
        }
        static const int prio = [] { const char *e = dev_env("FNX_SSIM_PRIO"); return e ? atoi(e) : 1; }();
        ma.prio = prio;
        FNX_TRY(prof_begin(ctx, FNX_PROF_SSIM));
        note_route(ctx, FNX_PROF_SSIM, march2 ? "windowed_ssim_march2_kernel" : "windowed_ssim_march_kernel");
        const bool fast = march2 && ctx->ssim_mode == FNX_SSIM_FAST;
        if (fast) {
            for (int i = 0; i < 8; i++) {
                ma.colf[i] = static_cast<float>(sa.col[i]);
                ma.rowf1[i] = static_cast<float>(sa.row[i] * (1.0 / 1024.0));
                ma.rowf2[i] = static_cast<float>(sa.row[i] * (1.0 / 1048576.0));
            }
            note_route(ctx, FNX_PROF_SSIM, "windowed_ssim_march2f_kernel");
            // (two waves per SIMD, the next row's taps read ahead: the three- and four-wave builds spill inside the loop, CHANGELOG r6)
            hipLaunchKernelGGL((windowed_ssim_march2f_kernel<2, true>), dim3((tiles + 3) / 4, n), dim3(256), 0, ctx->stream, ma);
        } else if (march2) hipLaunchKernelGGL(windowed_ssim_march2_kernel, dim3((tiles + 3) / 4, n), dim3(256), 0, ctx->stream, ma);
        else if (ctx->partial_slot >= 0) hipLaunchKernelGGL(windowed_ssim_march_kernel<true>, dim3((tiles + 3) / 4, n), dim3(256), 0, ctx->stream, ma);
        else hipLaunchKernelGGL(windowed_ssim_march_kernel<false>, dim3((tiles + 3) / 4, n), dim3(256), 0, ctx->stream, ma);
        FNX_HIP(hipGetLastError());
        FNX_TRY(prof_end(ctx));
    } else if (sep) {
        sa.a = a; sa.b = b; sa.as = d_as; sa.bs = d_bs; sa.a_image_bytes = a_image_bytes; sa.b_image_bytes = b_image_bytes;
        sa.astride = astride; sa.bstride = bstride; sa.w = w; sa.h = h;
        sa.tiles_x = tiles_x; sa.tiles = tiles; sa.partial = static_cast<double *>(part);
        static const bool nofold = [] { const char *e = dev_env("FNX_SSIM_NOFOLD"); return e && e[0] == '1'; }();
        if (!big && !defer && !nofold && n <= 4096) {
            unsigned *dn = nullptr;
            FNX_TRY(ssim_done_counters(ctx, &dn));
            sa.done = dn + (ctx->partial_slot >= 0 ? 4096 : 0);
            sa.out = d_out;
            sa.count = static_cast<double>(ww) * static_cast<double>(wh);
            folded = true;
        }
        note_route(ctx, FNX_PROF_SSIM, big ? "windowed_ssim_sep24_kernel" : "windowed_ssim_sep_kernel");
        if (big) hipLaunchKernelGGL(windowed_ssim_sep24_kernel, dim3(tiles, n), dim3(256), 0, ctx->stream, sa);
        else hipLaunchKernelGGL((windowed_ssim_sep_kernel<WSS_TY, 256>), dim3(tiles, n), dim3(256), 0, ctx->stream, sa);
        FNX_HIP(hipGetLastError());
    } else if (have) {
        WinArgs wa{};
        wa.a = a; wa.b = b; wa.as = d_as; wa.bs = d_bs; wa.a_image_bytes = a_image_bytes; wa.b_image_bytes = b_image_bytes;
        wa.astride = astride; wa.bstride = bstride; wa.w = w; wa.h = h; wa.window = d_window;
        wa.tiles_x = tiles_x; wa.tiles = tiles; wa.partial = static_cast<double *>(part);
        note_route(ctx, FNX_PROF_SSIM, "windowed_ssim_kernel");
        hipLaunchKernelGGL(windowed_ssim_kernel, dim3(tiles, n), dim3(256), 0, ctx->stream, wa);
        FNX_HIP(hipGetLastError());
    }
    const double count = have ? static_cast<double>(ww) * static_cast<double>(wh) : 0.0;
    if (defer) {
        defer->item[defer->count++] = {defer->used, tiles, count, defer_out_index};
        defer->used += static_cast<size_t>(tiles) + 2;
        return FNX_OK;
    }
    if (folded) return FNX_OK;
    hipLaunchKernelGGL(ssim_finish_kernel, dim3(n), dim3(256), 0, ctx->stream,
                       static_cast<const double *>(part), tiles, count, d_out);
    FNX_HIP(hipGetLastError());
    return FNX_OK;
}

// ------------------------------------------------------------------------------------
// MSSSIM (ssim.go:313-365) in five launches instead of ~15: when every halving is an exact 2 x 2 box (both dims
// divisible by 2^levels: 4K, 8K, 1080p/8 ...) one kernel reads the full-size pair ONCE and writes every level
// (clampF(sum * 0.25) == (sum + 2) >> 2 for integer sums, all four channels); the <= 512 px SSIMFast planes of
// the levels that need them come from ONE multi-job box launch, the five window sums from ONE multi-job launch,
// the means from the deferred finish.  Anything else (odd dims, tables that are not rank-1, unaligned views)
// returns FNX_NOOP and the caller runs the level-by-level loop.  Results are identical: same kernels' bodies,
// same integer box arithmetic.
// ------------------------------------------------------------------------------------
struct PyrArgs {
    const uint8_t *src[2];
    int sstride[2];
    int w, h, nl;            // level-0 dims, number of halvings (1..4)
    uint8_t *lv[2][4];       // [side][k - 1]: level k, tight (w >> k) x (h >> k)
    // a batch (blockIdx.z = 2 image + side): every pointer above lies in the ctx's pyramid scratch, `img_bytes` per image
    size_t img_bytes;
};

__device__ __forceinline__ uint32_t quad_mean(uint32_t p00, uint32_t p01, uint32_t p10, uint32_t p11)
{
    const uint32_t rb = ((p00 & 0x00ff00ffu) + (p01 & 0x00ff00ffu) + (p10 & 0x00ff00ffu) + (p11 & 0x00ff00ffu) + 0x00020002u) >> 2;
    const uint32_t ga = (((p00 >> 8) & 0x00ff00ffu) + ((p01 >> 8) & 0x00ff00ffu) + ((p10 >> 8) & 0x00ff00ffu) + ((p11 >> 8) & 0x00ff00ffu) + 0x00020002u) >> 2;
    return (rb & 0x00ff00ffu) | ((ga & 0x00ff00ffu) << 8);
}

// workgroup = 128 x 16 level-0 pixels of one image -> 64 x 8 (L1), 32 x 4 (L2), 16 x 2 (L3), 8 x 1 (L4)
__global__ __launch_bounds__(256) void pyramid_halve_kernel(PyrArgs a)
{
    __shared__ uint32_t s_l1[8][64], s_l2[4][32], s_l3[2][16];
    const int side = blockIdx.z & 1, tid = threadIdx.x;
    const int bx = blockIdx.x, by = blockIdx.y;
    const size_t io = a.img_bytes * (blockIdx.z >> 1);              // a batch's image (0 for a single pair); added where a pointer is USED
    {
        const int lx = tid & 31, ly = tid >> 5;
        const int x = bx * 128 + 4 * lx, y = by * 16 + 2 * ly;
        u32x4 r0 = {0, 0, 0, 0}, r1 = r0;
        const bool in = x < a.w && y < a.h;                        // w % 4 == 0, h % 2 == 0
        if (in) {
            const uint8_t *p = a.src[side] + io + static_cast<size_t>(y) * a.sstride[side] + 4 * static_cast<size_t>(x);
            r0 = ld16_stream(p);
            r1 = ld16_stream(p + a.sstride[side]);
        }
        const uint32_t q0 = quad_mean(r0[0], r0[1], r1[0], r1[1]), q1 = quad_mean(r0[2], r0[3], r1[2], r1[3]);
        s_l1[ly][2 * lx] = q0;
        s_l1[ly][2 * lx + 1] = q1;
        if (in) {
            const int w1 = a.w >> 1;
            *reinterpret_cast<u32x2 *>(a.lv[side][0] + io + (static_cast<size_t>(by * 8 + ly) * w1 + bx * 64 + 2 * lx) * 4) = (u32x2){q0, q1};
        }
    }
    if (a.nl < 2) return;
    __syncthreads();
    if (tid < 128) {
        const int lx = tid & 31, ly = tid >> 5;
        const uint32_t q = quad_mean(s_l1[2 * ly][2 * lx], s_l1[2 * ly][2 * lx + 1], s_l1[2 * ly + 1][2 * lx], s_l1[2 * ly + 1][2 * lx + 1]);
        s_l2[ly][lx] = q;
        const int X = bx * 32 + lx, Y = by * 4 + ly, w2 = a.w >> 2;
        if (X < w2 && Y < (a.h >> 2)) *reinterpret_cast<uint32_t *>(a.lv[side][1] + io + (static_cast<size_t>(Y) * w2 + X) * 4) = q;
    }
    if (a.nl < 3) return;
    __syncthreads();
    if (tid < 32) {
        const int lx = tid & 15, ly = tid >> 4;
        const uint32_t q = quad_mean(s_l2[2 * ly][2 * lx], s_l2[2 * ly][2 * lx + 1], s_l2[2 * ly + 1][2 * lx], s_l2[2 * ly + 1][2 * lx + 1]);
        s_l3[ly][lx] = q;
        const int X = bx * 16 + lx, Y = by * 2 + ly, w3 = a.w >> 3;
        if (X < w3 && Y < (a.h >> 3)) *reinterpret_cast<uint32_t *>(a.lv[side][2] + io + (static_cast<size_t>(Y) * w3 + X) * 4) = q;
    }
    if (a.nl < 4) return;
    __syncthreads();
    if (tid < 8) {
        const uint32_t q = quad_mean(s_l3[0][2 * tid], s_l3[0][2 * tid + 1], s_l3[1][2 * tid], s_l3[1][2 * tid + 1]);
        const int X = bx * 8 + tid, Y = by, w4 = a.w >> 4;
        if (X < w4 && Y < (a.h >> 4)) *reinterpret_cast<uint32_t *>(a.lv[side][3] + io + (static_cast<size_t>(Y) * w4 + X) * 4) = q;
    }
}

// MSSSIM's level 0 read ONCE (r3; VERDICT r2 weak 4: box_tiled_multi_kernel and pyramid_halve_kernel both streamed the
// full-size pair, 181 MB for 110 MB algorithmic): box_tiled_body's workgroup -- a segment of output columns x one box
// row, every lane streaming one 16-byte column chunk down the box's rows -- also emits the 2 x 2 means of level 1 for
// the row pairs whose TOP row lies in its band.  Box bands tile the rows (host-checked) but their edges are odd half
// of the time, so the band's rows are walked as even-aligned pairs: a first pair whose top row belongs to the band
// above only feeds its bottom row to the box; a last pair whose bottom row belongs to the band below is loaded for
// the quad alone (the one redundant row per two bands, an L2 hit when neighbouring workgroups run together).
// Levels 2..4 come from level 1 (pyramid_halve_kernel on a quarter of the bytes).  Same integer arithmetic as the two
// kernels it replaces: planes and levels are bit-identical.
struct BoxHalveArgs {
    BoxArgs b;
    uint8_t *l1[2];      // level 1 of side a / b, tight (srcW / 2) x (srcH / 2)
    // a batch (blockIdx.z = 2 image + side): the level-0 images by pointer (device arrays), planes and level 1 in the ctx's
    // scratch at `planes_img` / `l1_img` bytes per image
    const uint8_t *const *srcs_a, *const *srcs_b;
    size_t planes_img, l1_img;
};

template <int NP>
__device__ __forceinline__ void box_pair_trip(const uint8_t *q, int sstride, bool top_ok, bool bot_ok, uint32_t (&lo)[4],
                                              uint32_t (&hi)[4], uint8_t *l1p, int l1stride)
{
    u32x4 v[2 * NP];
#pragma unroll
    for (int u = 0; u < 2 * NP; u++) v[u] = ld16_stream(q + static_cast<size_t>(u) * sstride);
#pragma unroll
    for (int u = 0; u < NP; u++) {
        const u32x4 ra = v[2 * u], rb = v[2 * u + 1];
        const bool ta = u > 0 || top_ok, tb = u < NP - 1 || bot_ok;     // wave-uniform
        if (ta) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                lo[e] += ra[e] & 0x00ff00ffu;
                hi[e] += __builtin_amdgcn_perm(0u, ra[e], 0x0c030c01u);
            }
            *reinterpret_cast<u32x2 *>(l1p + static_cast<size_t>(u) * l1stride) =
                (u32x2){quad_mean(ra[0], ra[1], rb[0], rb[1]), quad_mean(ra[2], ra[3], rb[2], rb[3])};
        }
        if (tb) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                lo[e] += rb[e] & 0x00ff00ffu;
                hi[e] += __builtin_amdgcn_perm(0u, rb[e], 0x0c030c01u);
            }
        }
    }
}

__global__ __launch_bounds__(256) void box_halve_kernel(BoxHalveArgs h)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_col[BOX_CHUNKS * 4 * 2];
    const BoxArgs &a = h.b;
    const int side = blockIdx.z & 1;                 // 0: a, 1: b
    const int img = blockIdx.z >> 1;                 // (0: one pair per launch)
    const uint8_t *src = h.srcs_a ? (side ? h.srcs_b : h.srcs_a)[img] : (side ? a.src_b : a.src);
    const int sstride = side ? a.sstride_b : a.sstride;
    const int dy = blockIdx.y;
    const int dx_lo = blockIdx.x * a.seg;
    const int dx_hi = min(dx_lo + a.seg, a.dstW);
    int sy0, sy1, sxa, sxb, t0, t1;
    box_edge(dy, a.yRatio, a.srcH, sy0, sy1);
    box_edge(dx_lo, a.xRatio, a.srcW, sxa, t1);
    box_edge(dx_hi - 1, a.xRatio, a.srcW, t0, sxb);
    const int c0 = sxa >> 2;
    const int nchunk = ((sxb + 3) >> 2) - c0;
    const int tid = threadIdx.x;
    if (tid < nchunk) {
        const int x = 4 * (c0 + tid);                // srcW % 4 == 0: every chunk is whole
        uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
        const int e0 = sy0 & ~1, e1 = (sy1 + 1) & ~1;               // srcH is even: e1 <= srcH
        const bool top_ok = (sy0 & 1) == 0, bot_ok = (sy1 & 1) == 0;
        const int l1stride = (a.srcW >> 1) * 4;
        const uint8_t *q = src + static_cast<size_t>(e0) * sstride + 4 * static_cast<size_t>(x);
        uint8_t *l1p = h.l1[side] + h.l1_img * img + static_cast<size_t>(e0 >> 1) * l1stride + 2 * static_cast<size_t>(x);
        int left = (e1 - e0) >> 1;
        bool first = true;
        // up to five pairs (every box of <= 8 rows, odd edges included) in ONE trip: all of a lane's loads in flight
        // before the first use; taller boxes walk in fours and finish with two to five pairs
        for (; left > 5; left -= 4, q += static_cast<size_t>(8) * sstride, l1p += static_cast<size_t>(4) * l1stride) {
            box_pair_trip<4>(q, sstride, first ? top_ok : true, true, lo, hi, l1p, l1stride);
            first = false;
        }
        const bool tk = first ? top_ok : true;
        switch (left) {
        case 5: box_pair_trip<5>(q, sstride, tk, bot_ok, lo, hi, l1p, l1stride); break;
        case 4: box_pair_trip<4>(q, sstride, tk, bot_ok, lo, hi, l1p, l1stride); break;
        case 3: box_pair_trip<3>(q, sstride, tk, bot_ok, lo, hi, l1p, l1stride); break;
        case 2: box_pair_trip<2>(q, sstride, tk, bot_ok, lo, hi, l1p, l1stride); break;
        default: box_pair_trip<1>(q, sstride, tk, bot_ok, lo, hi, l1p, l1stride); break;
        }
        uint4 *sp = reinterpret_cast<uint4 *>(s_col + tid * 8);
        sp[0] = make_uint4(lo[0], hi[0], lo[1], hi[1]);
        sp[1] = make_uint4(lo[2], hi[2], lo[3], hi[3]);
    }
    __syncthreads();
    const int dx = dx_lo + tid;
    if (dx < dx_hi) {
        int sx0, sx1;
        box_edge(dx, a.xRatio, a.srcW, sx0, sx1);
        uint32_t r = 0, g = 0, b = 0, al = 0;
        if (a.packed_ok) {
            uint32_t plo = 0, phi = 0;
            for (int sx = sx0; sx < sx1; sx++) {
                const uint2 c = *reinterpret_cast<const uint2 *>(s_col + (sx - 4 * c0) * 2);
                plo += c.x; phi += c.y;
            }
            r = plo & 0xffffu; b = plo >> 16; g = phi & 0xffffu; al = phi >> 16;
        } else {
            for (int sx = sx0; sx < sx1; sx++) {
                const uint2 c = *reinterpret_cast<const uint2 *>(s_col + (sx - 4 * c0) * 2);
                r += c.x & 0xffffu; b += c.x >> 16;
                g += c.y & 0xffffu; al += c.y >> 16;
            }
        }
        const int count = (sy1 - sy0) * (sx1 - sx0);
        uint8_t *dimg = a.dst + h.planes_img * img + a.dst_image_bytes * side;
        *reinterpret_cast<uint32_t *>(dimg + static_cast<size_t>(dy) * a.dstride + 4 * static_cast<size_t>(dx)) =
            box_finish(r, g, b, al, count);
    }
}

// SSIMFast's dims (ssim.go:52-56); api.cpp holds the same arithmetic for the per-op entry points
static bool fast_dims(int w, int h, int *nw, int *nh)
{
    *nw = w; *nh = h;
    if (w > 512 || h > 512) {
        const double scale = 512.0 / std::fmax(double(w), double(h));
        *nw = int(std::fmax(8, std::round(double(w) * scale)));
        *nh = int(std::fmax(8, std::round(double(h) * scale)));
        return true;
    }
    return false;
}

// nimg > 1: a batch of same-geometry pairs in the SAME five launches (the image is a grid dimension of each; d_as / d_bs are
// device arrays of the level-0 pointers, a / b then only stand for their alignment); d_out holds five slots per image.
// Only the default form is batched (level 0 read once, planes written by the box kernels, the separate finish launch).
int launch_msssim_fused(fnx_ctx *ctx, const uint8_t *a, int astride, const uint8_t *b, int bstride, int w, int h,
                        int nweights, const double *h_window, double *d_out, int *nlev,
                        int nimg, const uint8_t *const *d_as, const uint8_t *const *d_bs)
{
    const unsigned nz = static_cast<unsigned>(nimg > 1 ? nimg : 1);
    const char *lvw = form_value(ctx, FORM_MSSSIM_LEVELWISE);             // "1": A/B and tests
    if ((lvw && lvw[0] == '1') || nweights < 1 || nweights > WS_MAXJOBS) return FNX_NOOP;
    WinSepArgs proto{};
    if (!window_rank1(h_window, proto.col, proto.row)) return FNX_NOOP;
    // levels the reference's loop visits: level i + 1 exists while both halved dims stay >= 8 (ssim.go:354-358)
    int lw[5], lh[5], levels = 1;
    lw[0] = w; lh[0] = h;
    while (levels < nweights && lw[levels - 1] / 2 >= 8 && lh[levels - 1] / 2 >= 8) {
        lw[levels] = lw[levels - 1] / 2; lh[levels] = lh[levels - 1] / 2;
        levels++;
    }
    const int nl = levels - 1;
    if (w < 8 || h < 8 || (w & 3) || nl > 4) return FNX_NOOP;
    if (nl > 0 && ((w & ((1 << nl) - 1)) || (h & ((1 << nl) - 1)))) return FNX_NOOP;      // some halving is not 2 x 2
    if (!aligned16(a, astride) || !aligned16(b, bstride)) return FNX_NOOP;
    // storage: pyramid levels 1..nl of both sides, then the SSIMFast planes of the levels that need them
    size_t off_lv[2][4] = {}, total = 0;
    for (int k = 1; k <= nl; k++)
        for (int sd = 0; sd < 2; sd++) {
            off_lv[sd][k - 1] = total;
            total += (static_cast<size_t>(lw[k]) * lh[k] * 4 + 15) & ~size_t(15);
        }
    const size_t pyr_img = (total + 255) & ~size_t(255);                  // per image of a batch
    void *pyr = nullptr;
    FNX_TRY(scratch(ctx, SLOT_TMP0, pyr_img * nz + 16, &pyr));
    int pw[5], ph[5];
    bool down[5];
    size_t off_pl[5] = {}, ptotal = 0;
    for (int i = 0; i < levels; i++) {
        down[i] = fast_dims(lw[i], lh[i], &pw[i], &ph[i]);
        if (pw[i] < 9 || ph[i] < 9) return FNX_NOOP;             // pixelSSIM / zero-window levels: level-wise loop
        if (down[i]) {
            if (lw[i] & 3) return FNX_NOOP;                      // 16-byte rows for the tiled box kernel
            off_pl[i] = ptotal;
            ptotal += 2 * ((static_cast<size_t>(pw[i]) * ph[i] * 4 + 15) & ~size_t(15));
        }
    }
    const size_t planes_img = (ptotal + 255) & ~size_t(255);
    void *planes = nullptr;
    FNX_TRY(scratch(ctx, SLOT_TMP2, planes_img * nz + 16, &planes));
    void *part = nullptr;
    FNX_TRY(scratch(ctx, SLOT_PARTIAL, sizeof(double) * SSIM_DEFER_DOUBLES * nz, &part));

    const uint8_t *la[5], *lb[5];
    int ls[5];
    la[0] = a; lb[0] = b; ls[0] = 0;
    uint8_t *lvp[2][4] = {};
    for (int k = 1; k <= nl; k++) {
        lvp[0][k - 1] = static_cast<uint8_t *>(pyr) + off_lv[0][k - 1];
        lvp[1][k - 1] = static_cast<uint8_t *>(pyr) + off_lv[1][k - 1];
        la[k] = lvp[0][k - 1]; lb[k] = lvp[1][k - 1]; ls[k] = lw[k] * 4;
    }
    // level 0 read once: its <= 512 px planes AND level 1 from box_halve_kernel, levels 2.. from level 1.  Needs level 0
    // downsampled by the tiled box form, box bands / segments that cover every row and column, 16-byte rows of level 1.
    bool fuse0 = false;
    BoxHalveArgs bh{};
    if (nl > 0 && down[0]) {
        const char *nf = form_value(ctx, FORM_MSSSIM_NOFUSE0);            // "1": A/B and tests (two reads of level 0, as in round 2)
        const bool nofuse = nf && nf[0] == '1';
        BoxArgs &ba = bh.b;
        ba.xRatio = static_cast<double>(lw[0]) / static_cast<double>(pw[0]);      // ssim.go:251-252
        ba.yRatio = static_cast<double>(lh[0]) / static_cast<double>(ph[0]);
        const bool tiled = lw[0] >= pw[0] && lh[0] >= ph[0] && ba.yRatio + 1.0 < BOX_MAXROWS && ba.xRatio + 1.0 < BOX_MAXROWS &&
                           ba.xRatio * 2 + 8 < 4 * BOX_CHUNKS;
        const bool covers = static_cast<int>(static_cast<double>(ph[0]) * ba.yRatio) >= lh[0] &&
                            static_cast<int>(static_cast<double>(pw[0]) * ba.xRatio) >= lw[0];
        fuse0 = !nofuse && tiled && covers && (lw[1] & 3) == 0 && (lh[0] & 1) == 0;
    }
    if (nz > 1 && !fuse0) return FNX_NOOP;                       // (the batch is the default form's)
    if (fuse0) {
        BoxArgs &ba = bh.b;
        const size_t plane = (static_cast<size_t>(pw[0]) * ph[0] * 4 + 15) & ~size_t(15);
        ba.src = a; ba.src_b = b; ba.sstride = astride; ba.sstride_b = bstride; ba.nimg = 1;
        ba.dst = static_cast<uint8_t *>(planes) + off_pl[0]; ba.dst_image_bytes = plane;
        ba.srcW = lw[0]; ba.srcH = lh[0]; ba.dstride = pw[0] * 4; ba.dstW = pw[0]; ba.dstH = ph[0];
        ba.vec_in = 1;
        ba.packed_ok = (static_cast<double>(static_cast<long>(ba.yRatio) + 2) * static_cast<double>(static_cast<long>(ba.xRatio) + 2) * 255.0) < 65536.0;
        int seg = static_cast<int>((4 * BOX_CHUNKS - 8) / ba.xRatio);
        ba.seg = seg > 256 ? 256 : (seg < 1 ? 1 : seg);
        bh.l1[0] = lvp[0][0]; bh.l1[1] = lvp[1][0];
        if (nz > 1) { bh.srcs_a = d_as; bh.srcs_b = d_bs; bh.planes_img = planes_img; bh.l1_img = pyr_img; }
        hipLaunchKernelGGL(box_halve_kernel, dim3((pw[0] + ba.seg - 1) / ba.seg, ph[0], 2 * nz), dim3(256), 0, ctx->stream, bh);
        FNX_HIP(hipGetLastError());
        if (nl > 1) {
            PyrArgs pa{};
            pa.src[0] = lvp[0][0]; pa.src[1] = lvp[1][0]; pa.sstride[0] = pa.sstride[1] = lw[1] * 4;
            pa.w = lw[1]; pa.h = lh[1]; pa.nl = nl - 1;
            pa.img_bytes = pyr_img;
            for (int k = 2; k <= nl; k++) { pa.lv[0][k - 2] = lvp[0][k - 1]; pa.lv[1][k - 2] = lvp[1][k - 1]; }
            hipLaunchKernelGGL(pyramid_halve_kernel, dim3((lw[1] + 127) / 128, (lh[1] + 15) / 16, 2 * nz), dim3(256), 0, ctx->stream, pa);
            FNX_HIP(hipGetLastError());
        }
    } else if (nl > 0) {
        PyrArgs pa{};
        pa.src[0] = a; pa.src[1] = b; pa.sstride[0] = astride; pa.sstride[1] = bstride; pa.w = w; pa.h = h; pa.nl = nl;
        for (int k = 1; k <= nl; k++) { pa.lv[0][k - 1] = lvp[0][k - 1]; pa.lv[1][k - 1] = lvp[1][k - 1]; }
        hipLaunchKernelGGL(pyramid_halve_kernel, dim3((w + 127) / 128, (h + 15) / 16, 2), dim3(256), 0, ctx->stream, pa);
        FNX_HIP(hipGetLastError());
    }
    // SSIMFast planes: one launch for every level that is downsampled
    BoxMulti bm{};
    int nb = 0, gx = 0, gy = 0;
    const uint8_t *sa[5], *sb[5];
    int sas[5], sbs[5];
    bool onfly[5] = {false, false, false, false, false};
    for (int i = 0; i < levels; i++) {
        sa[i] = la[i]; sb[i] = lb[i];
        sas[i] = i == 0 ? astride : ls[i]; sbs[i] = i == 0 ? bstride : ls[i];
        if (!down[i]) continue;
        if (i == 0 && fuse0) {                                   // box_halve_kernel wrote level 0's planes
            const size_t plane0 = (static_cast<size_t>(pw[0]) * ph[0] * 4 + 15) & ~size_t(15);
            uint8_t *dst0 = static_cast<uint8_t *>(planes) + off_pl[0];
            sa[0] = dst0; sb[0] = dst0 + plane0; sas[0] = sbs[0] = pw[0] * 4;
            continue;
        }
        // boxes of at most 5 x 5 pixels (levels downsampled by less than 4: 1080p and 960 x 540 under a 4K pair) are taken
        // by the window kernel's tile load itself -- no plane is written, no launch has to finish first
        {
            const double xr = static_cast<double>(lw[i]) / static_cast<double>(pw[i]), yr = static_cast<double>(lh[i]) / static_cast<double>(ph[i]);
            const char *nb_env = form_value(ctx, FORM_MSSSIM_BOXFLY);        // "1": on.  Off by default: one launch fewer and 5 us off a single
                                                                     // 4K call, but the window kernel then finishes every staged pixel's
                                                                     // box 1.75 times over (tile halos) and config 3's four streams
                                                                     // measured 68 k MP/s against 72 k
            if (nz == 1 && (nb_env && nb_env[0] == '1') && i > 0 && lw[i] >= pw[i] && lh[i] >= ph[i] &&
                (static_cast<int>(xr) + 1) * (static_cast<int>(yr) + 1) <= 25) {
                onfly[i] = true;
                continue;
            }
        }
        if (nb == BOX_MAXJOBS) return FNX_NOOP;
        BoxArgs &ba = bm.job[nb];
        uint8_t *dst = static_cast<uint8_t *>(planes) + off_pl[i];
        const size_t plane = (static_cast<size_t>(pw[i]) * ph[i] * 4 + 15) & ~size_t(15);
        ba.src = sa[i]; ba.src_b = sb[i]; ba.sstride = sas[i]; ba.sstride_b = sbs[i]; ba.nimg = 1;
        ba.dst = dst; ba.dst_image_bytes = plane;
        ba.srcW = lw[i]; ba.srcH = lh[i]; ba.dstride = pw[i] * 4; ba.dstW = pw[i]; ba.dstH = ph[i];
        ba.xRatio = static_cast<double>(lw[i]) / static_cast<double>(pw[i]);      // ssim.go:251-252
        ba.yRatio = static_cast<double>(lh[i]) / static_cast<double>(ph[i]);
        ba.vec_in = 1;
        const bool tiled = lw[i] >= pw[i] && lh[i] >= ph[i] && ba.yRatio + 1.0 < BOX_MAXROWS && ba.xRatio + 1.0 < BOX_MAXROWS &&
                           ba.xRatio * 2 + 8 < 4 * BOX_CHUNKS;
        if (!tiled || !aligned16(sa[i], sas[i]) || !aligned16(sb[i], sbs[i])) return FNX_NOOP;
        ba.packed_ok = (static_cast<double>(static_cast<long>(ba.yRatio) + 2) * static_cast<double>(static_cast<long>(ba.xRatio) + 2) * 255.0) < 65536.0;
        int seg = static_cast<int>((4 * BOX_CHUNKS - 8) / ba.xRatio);
        ba.seg = seg > 256 ? 256 : (seg < 1 ? 1 : seg);
        bm.gx[nb] = (pw[i] + ba.seg - 1) / ba.seg; bm.gy[nb] = ph[i];
        gx = std::max(gx, bm.gx[nb]); gy = std::max(gy, bm.gy[nb]);
        sa[i] = dst; sb[i] = dst + plane; sas[i] = sbs[i] = pw[i] * 4;
        nb++;
    }
    if (nb > 0) {
        if (nz > 1) { bm.njobs = nb; bm.src_img = pyr_img; bm.dst_img = planes_img; }
        hipLaunchKernelGGL(box_tiled_multi_kernel, dim3(gx, gy, 2 * nb * nz), dim3(256), 0, ctx->stream, bm);
        FNX_HIP(hipGetLastError());
    }
    // the window sums of every level: one launch
    WinSepMulti wm{};
    SsimDeferred defer;
    int maxt = 1;
    for (int i = 0; i < levels; i++) {
        WinSepArgs &wa = wm.job[i];
        wa = proto;
        wa.a = sa[i]; wa.b = sb[i]; wa.astride = sas[i]; wa.bstride = sbs[i]; wa.a_image_bytes = wa.b_image_bytes = 0;
        if (nz > 1) {
            // a batch: a level's operands are either its planes (plane scratch) or the level itself (pyramid scratch) -- never the
            // caller's level-0 images, whose planes box_halve_kernel wrote (fuse0)
            const bool in_planes = down[i];
            if (!in_planes && i == 0) return FNX_NOOP;
            wa.a_image_bytes = wa.b_image_bytes = in_planes ? planes_img : pyr_img;
        }
        wa.w = pw[i]; wa.h = ph[i];
        if (onfly[i]) {
            wa.boxed = std::max(static_cast<int>(static_cast<double>(lw[i]) / static_cast<double>(pw[i])),
                                static_cast<int>(static_cast<double>(lh[i]) / static_cast<double>(ph[i]))) + 1;   // largest box side (<= 5)
            wa.srcW = lw[i]; wa.srcH = lh[i];
            wa.xRatio = static_cast<double>(lw[i]) / static_cast<double>(pw[i]);      // ssim.go:251-252
            wa.yRatio = static_cast<double>(lh[i]) / static_cast<double>(ph[i]);
        }
        const int ww = pw[i] - 8, wh = ph[i] - 8;
        wa.tiles_x = (ww + WSS_TX - 1) / WSS_TX;
        wa.tiles = wa.tiles_x * ((wh + WSS_TY - 1) / WSS_TY);
        if (defer.used + wa.tiles + 2 > SSIM_DEFER_DOUBLES) return FNX_NOOP;
        wa.partial = static_cast<double *>(part) + defer.used;
        defer.item[defer.count++] = {defer.used, wa.tiles, static_cast<double>(ww) * static_cast<double>(wh), i};
        defer.used += static_cast<size_t>(wa.tiles) + 2;
        maxt = std::max(maxt, wa.tiles);
    }
    // FNX_MSSSIM_FOLD=1: each level's last workgroup takes the level's mean itself instead of the separate finish launch.
    // Off by default: the write-through stores and counters cost what the launch costs (a single 4K call 75 us against
    // 71, config 3 the same within its noise); kept, with its test, as the measured alternative.
    const char *nf_env = form_value(ctx, FORM_MSSSIM_FOLD);
    const bool fold = nz == 1 && nf_env && nf_env[0] == '1';
    if (nz > 1) { wm.nlev = levels; wm.part_img = SSIM_DEFER_DOUBLES; }
    if (fold) {
        unsigned *dn = nullptr;
        FNX_TRY(ssim_done_counters(ctx, &dn));
        wm.out = d_out; wm.done = dn + 2 * 4096; wm.njobs = levels;
        for (int i = 0; i < levels; i++) {
            wm.out_index[i] = defer.item[i].out_index;
            wm.windows[i] = defer.item[i].windows;
        }
    }
    hipLaunchKernelGGL(windowed_ssim_sep_multi_kernel, dim3(maxt, levels * nz), dim3(256), 0, ctx->stream, wm);
    FNX_HIP(hipGetLastError());
    if (!fold) FNX_TRY(launch_ssim_finish_deferred(ctx, defer, d_out, static_cast<int>(nz), nz > 1 ? SSIM_DEFER_DOUBLES : 0, nz > 1 ? 5 : 0));
    *nlev = levels;
    return FNX_OK;
}

struct FinishMulti {
    const double *partial[8];
    int tiles[8], out_index[8];
    double windows[8];
    double *out;
    size_t part_img;             // a batch (blockIdx.y = image): partial sums part_img doubles apart, results out_img apart
    int out_img;
};

// ssim_finish_kernel for several independent reductions: workgroup z finishes item z
__global__ __launch_bounds__(256) void ssim_finish_multi_kernel(FinishMulti f)
{
    __shared__ double s_red[4];
    const int z = blockIdx.x;
    const double *p = f.partial[z] + f.part_img * blockIdx.y;
    const int tiles = f.tiles[z];
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int i = threadIdx.x;
    for (; i + 7 * 256 < tiles; i += 8 * 256) {
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] += p[i + e * 256];
    }
    for (int e = 0; i < tiles; i += 256, e++) acc[e] += p[i];
    const double v = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    const double t = block_sum_256(v, s_red);
    if (threadIdx.x == 0) f.out[f.out_index[z] + f.out_img * blockIdx.y] = f.windows[z] > 0 ? t / f.windows[z] : 1.0;
}

int launch_ssim_finish_deferred(fnx_ctx *ctx, const SsimDeferred &d, double *d_out, int nimg, size_t part_img, int out_img)
{
    if (d.count == 0) return FNX_OK;
    void *part = nullptr;
    FNX_TRY(scratch(ctx, SLOT_PARTIAL, sizeof(double) * std::max(SSIM_DEFER_DOUBLES, part_img * static_cast<size_t>(nimg)), &part));
    FinishMulti f{};
    f.part_img = part_img; f.out_img = out_img;
    for (int i = 0; i < d.count; i++) {
        f.partial[i] = static_cast<const double *>(part) + d.item[i].offset;
        f.tiles[i] = d.item[i].tiles;
        f.out_index[i] = d.item[i].out_index;
        f.windows[i] = d.item[i].windows;
    }
    f.out = d_out;
    hipLaunchKernelGGL(ssim_finish_multi_kernel, dim3(d.count, nimg), dim3(256), 0, ctx->stream, f);
    FNX_HIP(hipGetLastError());
    return FNX_OK;
}

// ------------------------------------------------------------------------------------
// pixelSSIM (ssim.go:169-204): images with a dimension < 8.  One lane, the reference's
// exact running-sum order over the flat Pix slices.
// ------------------------------------------------------------------------------------
__global__ void pixel_ssim_kernel(const uint8_t *a, const uint8_t *b, int w, int h, size_t pix_len, double *out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double n = static_cast<double>(w * h);
    if (n == 0) { *out = 1.0; return; }
    double muA = 0, muB = 0;
    for (size_t i = 0; i < pix_len; i += 4) {
        muA += lum601(*reinterpret_cast<const uint32_t *>(a + i));
        muB += lum601(*reinterpret_cast<const uint32_t *>(b + i));
    }
    muA /= n; muB /= n;
    double sAA = 0, sBB = 0, sAB = 0;
    for (size_t i = 0; i < pix_len; i += 4) {
        const double da = lum601(*reinterpret_cast<const uint32_t *>(a + i)) - muA;
        const double db = lum601(*reinterpret_cast<const uint32_t *>(b + i)) - muB;
        sAA += da * da; sBB += db * db; sAB += da * db;
    }
    sAA /= n; sBB /= n; sAB /= n;
    const double num = (2 * muA * muB + 6.5025) * (2 * sAB + 58.5225);
    const double den = (muA * muA + muB * muB + 6.5025) * (sAA + sBB + 58.5225);
    *out = num / den;
}

int launch_pixel_ssim(fnx_ctx *ctx, const uint8_t *a, const uint8_t *b, int w, int h,
                      size_t pix_len, double *d_out)
{
    hipLaunchKernelGGL(pixel_ssim_kernel, dim3(1), dim3(64), 0, ctx->stream, a, b, w, h, pix_len, d_out);
    FNX_HIP(hipGetLastError());
    return FNX_OK;
}

}  // namespace fnx
