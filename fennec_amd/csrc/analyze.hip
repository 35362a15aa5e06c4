// Analyze (analyze.go:26-124) and the flat scans isOpaque / isGrayscale (convert.go:66-84) on gfx950.
// SURVEY 8(f) item 3 -- the first row widened after the hot path proper.
//
//  * analyze_pass_kernel: THE full-image pass (analyze.go:53-80): luminance histogram
//    (bin int(lum + 0.5)), luminance sum, has-alpha and all-grey flags.  One 16-byte
//    non-temporal load = 4 px per lane; fp64 luminance in the reference's operation order
//    (so every bin is exact); histogram in LDS, one copy per wave; per-workgroup partials
//    (no global atomics: a thousand workgroups hammering 256 addresses serialise).
//  * analyze_finish_kernel: column sums of the histogram partials (exact integers), the
//    brightness partials in a fixed order (bit-reproducible; the reference's own sum is one long
//    serial fp64 chain, so its last bits are an accident of order -- tolerance 1e-12 relative).
//  * analyze_sampled_kernel: the three sampled statistics -- distinct colours among every
//    step-th pixel (device hash set, order-free: the reference only reads len() capped at 1024),
//    contrast on the <=100x100 grid (needs the mean: runs after the finish), Sobel edge count on
//    the <=~200x200 grid (integer count; fp64 Sobel + IEEE sqrt in the reference's order: exact).
#include "common.hpp"
#include "devutil.hpp"

#include <algorithm>
#include <cmath>

namespace fnx {

constexpr int AN_WG_MAX = 1024;        // workgroups of the full pass per image: 4 per CU for one image, fewer in a batch
constexpr int AN_CBLOCKS = 8;          // workgroups of the contrast grid per image (each one is latency-bound)
constexpr int AN_HASH_CAP = 1 << 17;   // >= 2x the 50 000 + colour samples

struct PassArgs2 {
    const uint8_t *src;
    const uint8_t *const *srcs;
    int sstride, w, h;
    int upr;                 // 4-px units per row (tight images: ONE row of w*h px)
    uint32_t upr_magic;      // ceil(2^32 / upr)
    int rows, row_px;        // iteration space: rows x row_px pixels
    int vec;                 // 16-byte loads allowed
    long long units;
    int G;                   // workgroups per image
    uint32_t *hist_part;     // [n][G][256]
    double *bright_part;     // [n][G]
    uint32_t *flag_part;     // [n][G]   bit 0: some alpha < 255, bit 1: some r != g || g != b
};

// luminance bin, brightness and flags of one pixel (analyze.go:57-72)
__device__ __forceinline__ int an_pixel(uint32_t p, double &bright, uint32_t &flags)
{
    const double lum = lum601(p);                              // analyze.go:62
    bright += lum;
    const uint32_t r = p & 0xffu, g = (p >> 8) & 0xffu, b = (p >> 16) & 0xffu;
    flags |= ((p >> 24) < 255u ? 1u : 0u) | ((r != g || g != b) ? 2u : 0u);
    return static_cast<int>(lum + 0.5);                        // analyze.go:64
}

// two histogram increments; equal bins (flat image areas) become ONE atomic of 2
__device__ __forceinline__ void an_count2(uint32_t *hist, int b0, int b1)
{
    atomicAdd(&hist[b0], b0 == b1 ? 2u : 1u);
    if (b0 != b1) atomicAdd(&hist[b1], 1u);
}

// LDS histograms: AN_COPIES per wave, picked by lane parity.  Same-address LDS atomics serialise, and
// in flat image areas every lane of a wave hits the same bin: with one histogram per wave and one
// atomic per pixel a solid-colour 4K image took 27.9 us against 6.9 us for a noisy one.  Merging
// equal neighbours (an_count2) and 2 copies give 8.6 us solid / 7.3 us noisy; 4 and 8 copies were no
// better on solid and cost the noisy case occupancy (7.4 / 8.5 us).
constexpr int AN_COPIES = 2;

__global__ __launch_bounds__(256) void analyze_pass_kernel(PassArgs2 a)
{
    __shared__ uint32_t s_hist[4 * AN_COPIES][256];
    __shared__ double s_red[4];
    __shared__ uint32_t s_flag[4];
    const int tid = threadIdx.x, wave = tid >> 6, z = blockIdx.y;
    const uint8_t *src = a.srcs ? a.srcs[z] : a.src;
    for (int i = tid; i < 4 * AN_COPIES * 256; i += 256) (&s_hist[0][0])[i] = 0;
    __syncthreads();
    uint32_t *hist = s_hist[wave * AN_COPIES + (tid & (AN_COPIES - 1))];
    double bright = 0.0;
    uint32_t flags = 0;
    for (long long u = static_cast<long long>(blockIdx.x) * 256 + tid; u < a.units; u += static_cast<long long>(a.G) * 256) {
        int y = 0;
        long long c = u;
        if (a.rows > 1) {
            y = __umulhi(static_cast<uint32_t>(u), a.upr_magic);   // u < 2^32 / upr is checked by the host
            c = u - static_cast<long long>(y) * a.upr;
        }
        const long long x = 4 * c;
        const uint8_t *p = src + static_cast<size_t>(y) * a.sstride + 4 * x;
        const long long left = a.row_px - x;
        if (a.vec && left >= 4) {
            const u32x4 v = ld16_stream(p);
            const int b0 = an_pixel(v.x, bright, flags), b1 = an_pixel(v.y, bright, flags);
            const int b2 = an_pixel(v.z, bright, flags), b3 = an_pixel(v.w, bright, flags);
            an_count2(hist, b0, b1);
            an_count2(hist, b2, b3);
        } else {
            for (int e = 0; e < 4 && e < left; e++) atomicAdd(&hist[an_pixel(*(g_u32 *)(p + 4 * e), bright, flags)], 1u);
        }
    }
    // brightness: fixed tree (lane order inside the wave, then waves 0..3)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        bright += __shfl_down(bright, off, 64);
        flags |= __shfl_down(flags, off, 64);
    }
    if ((tid & 63) == 0) { s_red[wave] = bright; s_flag[wave] = flags; }
    __syncthreads();
    const size_t part = static_cast<size_t>(z) * a.G + blockIdx.x;
    uint32_t cnt = 0;
#pragma unroll 8
    for (int k = 0; k < 4 * AN_COPIES; k++) cnt += s_hist[k][tid];
    a.hist_part[part * 256 + tid] = cnt;
    if (tid == 0) {
        a.bright_part[part] = ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
        a.flag_part[part] = s_flag[0] | s_flag[1] | s_flag[2] | s_flag[3];
    }
}

// grid (16, n): workgroup j sums bins [16j, 16j+16) over the G partial histograms (thread = bin x 16-way
// split of g, LDS tree); workgroup 0 also sums the brightness partials (fixed order) and the flags.
__global__ __launch_bounds__(256) void analyze_finish_kernel(const uint32_t *hist_part, const double *bright_part,
                                                             const uint32_t *flag_part, int G, fnx_analysis *res)
{
    __shared__ unsigned long long s_c[16][17];
    __shared__ double s_b[256];
    __shared__ uint32_t s_f;
    const int tid = threadIdx.x, z = blockIdx.y, bin = blockIdx.x * 16 + (tid & 15), lane = tid >> 4;
    fnx_analysis *r = res + z;
    unsigned long long cnt = 0;
    for (int g = lane; g < G; g += 16) cnt += hist_part[(static_cast<size_t>(z) * G + g) * 256 + bin];
    s_c[lane][tid & 15] = cnt;
    __syncthreads();
    if (tid < 16) {
        unsigned long long t = 0;
        for (int l = 0; l < 16; l++) t += s_c[l][tid];
        r->histogram[blockIdx.x * 16 + tid] = t;
    }
    if (blockIdx.x != 0) return;
    if (tid == 0) s_f = 0;
    __syncthreads();
    double b = 0.0;
    uint32_t f = 0;
    for (int g = tid; g < G; g += 256) {
        b += bright_part[static_cast<size_t>(z) * G + g];
        f |= flag_part[static_cast<size_t>(z) * G + g];
    }
    s_b[tid] = b;
    atomicOr(&s_f, f);
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) s_b[tid] += s_b[tid + off];
        __syncthreads();
    }
    if (tid == 0) {
        r->bright_sum = s_b[0];
        r->has_alpha = s_f & 1u;
        r->is_grayscale = (s_f & 2u) ? 0 : 1;
        r->variance_sum = 0.0;
        r->sample_count = 0;
        r->edge_count = 0;
        r->edge_total = 0;
        r->unique_colors = 0;
        r->pad = 0;
    }
}

struct SampArgs {
    const uint8_t *src;
    const uint8_t *const *srcs;
    int sstride, w, h;
    long long color_samples, color_step;       // analyze.go:46-50,73-77
    long long color_k0;                        // first sample of this launch
    int color_blocks;
    int colors_only;                           // second colour launch: skip images that already hold 1024
    int cstep_x, cstep_y, cnx, cny;            // contrast grid (analyze.go:93-109)
    int estep_x, estep_y, enx, eny;            // edge grid (analyze.go:150-156)
    unsigned long long *hash;                  // [n][AN_HASH_CAP], zeroed
    uint32_t *count_part;                      // [n][gridDim.x]: per-workgroup counts (new colours / edges)
    double *var_part;                          // [n][AN_CBLOCKS]: per-workgroup sums of (lum - mean)^2
    fnx_analysis *res;
};

__device__ __forceinline__ double sobel_lum(const uint8_t *src, int stride, int x, int y)   // analyze.go:186-189
{
    return lum601(*(g_u32 *)(src + static_cast<size_t>(y) * stride + 4 * static_cast<size_t>(x)));
}

__global__ __launch_bounds__(256) void analyze_sampled_kernel(SampArgs a)
{
    __shared__ double s_red[4];
    const int tid = threadIdx.x, z = blockIdx.y, b = blockIdx.x;
    const uint8_t *src = a.srcs ? a.srcs[z] : a.src;
    fnx_analysis *r = a.res + z;
    if (b < a.color_blocks) {                  // ---- sampled colour set
        const long long k = a.color_k0 + static_cast<long long>(b) * 256 + tid;
        bool fresh = false;
        // the reference stops inserting at 1024 entries (analyze.go:73) and only len() is read: the first
        // launch hashes the first 2048 samples, the second one the rest -- unless the fold in between
        // already counted 1024 (photographs: always), which spares ~48 000 device-scope CAS per image
        const bool skip = a.colors_only && r->unique_colors >= 1024;
        if (!skip && k < a.color_samples) {
            const long long idx = k * a.color_step;
            const int y = static_cast<int>(idx / a.w), x = static_cast<int>(idx - static_cast<long long>(y) * a.w);
            const uint32_t p = *(g_u32 *)(src + static_cast<size_t>(y) * a.sstride + 4 * static_cast<size_t>(x));
            const unsigned long long key = (1ull << 32) | p;     // any bijection of (r,g,b,a) counts the same
            unsigned long long *tab = a.hash + static_cast<size_t>(z) * AN_HASH_CAP;
            uint32_t slot = (p * 2654435761u) >> (32 - 17);
            for (;;) {
                // a slot only ever goes 0 -> key: a (possibly stale) plain read that already shows this
                // key settles a duplicate without an atomic; anything else is decided by the CAS
                unsigned long long old = tab[slot];
                if (old == key) break;
                if (old == 0ull) {
                    old = atomicCAS(&tab[slot], 0ull, key);
                    if (old == 0ull) { fresh = true; break; }
                    if (old == key) break;
                }
                slot = (slot + 1) & (AN_HASH_CAP - 1);
            }
        }
        // counts go to a per-workgroup slot and are folded afterwards: hundreds of device-scope atomics on
        // ONE word cost ~80 ns each when every XCD contends for it (measured: they were 2/3 of this kernel)
        const int c = __syncthreads_count(fresh);
        if (tid == 0) a.count_part[static_cast<size_t>(z) * gridDim.x + b] = c;
    } else if (b < a.color_blocks + AN_CBLOCKS) {   // ---- contrast: sum (lum - mean)^2 on the fixed grid
        const double mean = r->bright_sum / static_cast<double>(static_cast<long long>(a.w) * a.h);
        double v = 0.0;
        const int total = a.cnx * a.cny, cb = b - a.color_blocks;
        const int per = (total + AN_CBLOCKS - 1) / AN_CBLOCKS, lo = cb * per, hi = min(total, lo + per);
        for (int s0 = lo + tid; s0 < hi; s0 += 8 * 256) {        // 8 independent loads in flight per lane
            uint32_t px[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int s = s0 + e * 256;
                const int iy = s < hi ? s / a.cnx : 0, ix = s < hi ? s - iy * a.cnx : 0;
                px[e] = *(g_u32 *)(src + static_cast<size_t>(iy * a.cstep_y) * a.sstride + 4 * static_cast<size_t>(ix * a.cstep_x));
            }
#pragma unroll
            for (int e = 0; e < 8; e++) {
                if (s0 + e * 256 < hi) {
                    const double d = lum601(px[e]) - mean;
                    v += d * d;
                }
            }
        }
        const double t = block_sum_256(v, s_red);
        if (tid == 0) a.var_part[static_cast<size_t>(z) * AN_CBLOCKS + cb] = t;
    } else {                                   // ---- Sobel edge count (analyze.go:158-177)
        const int s = (b - a.color_blocks - AN_CBLOCKS) * 256 + tid;
        bool edge = false;
        if (s < a.enx * a.eny) {
            const int iy = s / a.enx, ix = s - iy * a.enx;
            const int x = 1 + ix * a.estep_x, y = 1 + iy * a.estep_y;
            const int st = a.sstride;
            const double gx = sobel_lum(src, st, x + 1, y - 1) - sobel_lum(src, st, x - 1, y - 1) +
                              2 * sobel_lum(src, st, x + 1, y) - 2 * sobel_lum(src, st, x - 1, y) +
                              sobel_lum(src, st, x + 1, y + 1) - sobel_lum(src, st, x - 1, y + 1);
            const double gy = sobel_lum(src, st, x - 1, y + 1) - sobel_lum(src, st, x - 1, y - 1) +
                              2 * sobel_lum(src, st, x, y + 1) - 2 * sobel_lum(src, st, x, y - 1) +
                              sobel_lum(src, st, x + 1, y + 1) - sobel_lum(src, st, x + 1, y - 1);
            edge = sqrt(gx * gx + gy * gy) > 30.0;
        }
        const int c = __syncthreads_count(edge);
        if (tid == 0) a.count_part[static_cast<size_t>(z) * gridDim.x + b] = c;
    }
}

// one workgroup per image: fold the per-workgroup counts of analyze_sampled_kernel
__global__ __launch_bounds__(256) void analyze_fold_kernel(const uint32_t *count_part, int blocks, int color_blocks,
                                                           const double *var_part, long long grid_samples,
                                                           long long edge_total, int colors_only, fnx_analysis *res)
{
    __shared__ int s_c[2];
    const int tid = threadIdx.x, z = blockIdx.x;
    if (tid < 2) s_c[tid] = 0;
    __syncthreads();
    int colors = 0, edges = 0;
    for (int b = tid; b < blocks; b += 256) {
        const int c = static_cast<int>(count_part[static_cast<size_t>(z) * blocks + b]);
        if (b < color_blocks) colors += c;
        else if (b >= color_blocks + AN_CBLOCKS) edges += c;
    }
    atomicAdd(&s_c[0], colors);
    atomicAdd(&s_c[1], edges);
    __syncthreads();
    if (tid == 0) {
        if (colors_only) {
            res[z].unique_colors += s_c[0];
        } else {
            res[z].unique_colors = s_c[0];
            res[z].edge_count = s_c[1];
            res[z].edge_total = edge_total;
            double v = 0.0;                                   // fixed order: bit-reproducible
            for (int k = 0; k < AN_CBLOCKS; k++) v += var_part[static_cast<size_t>(z) * AN_CBLOCKS + k];
            res[z].variance_sum = v;
            res[z].sample_count = grid_samples;
        }
    }
}

// n images (one pointer, or a device pointer array) -> d_res[n] (device).
int launch_analyze(fnx_ctx *ctx, int n, const uint8_t *src, const uint8_t *const *srcs, int sstride, int w, int h,
                   bool aligned16_ok, fnx_analysis *d_res)
{
    if (n <= 0) return FNX_OK;
    void *hp = nullptr, *bp = nullptr, *hash = nullptr;
    // workgroups per image: fill the chip ~4 deep whatever the batch size
    const int G = std::max(128, std::min(AN_WG_MAX, (4 * ctx->num_cus + n - 1) / n));
    FNX_TRY(scratch(ctx, SLOT_TMP0, sizeof(uint32_t) * 256 * G * static_cast<size_t>(n), &hp));
    FNX_TRY(scratch(ctx, SLOT_TMP1, (sizeof(double) + sizeof(uint32_t)) * G * static_cast<size_t>(n) + 16, &bp));
    FNX_TRY(scratch(ctx, SLOT_TMP3, sizeof(unsigned long long) * AN_HASH_CAP * static_cast<size_t>(n), &hash));
    FNX_HIP(hipMemsetAsync(hash, 0, sizeof(unsigned long long) * AN_HASH_CAP * static_cast<size_t>(n), ctx->stream));

    PassArgs2 pa{};
    pa.src = src; pa.srcs = srcs; pa.sstride = sstride; pa.w = w; pa.h = h;
    pa.hist_part = static_cast<uint32_t *>(hp);
    pa.bright_part = static_cast<double *>(bp);
    pa.flag_part = reinterpret_cast<uint32_t *>(pa.bright_part + static_cast<size_t>(G) * n);
    pa.G = G;
    if (sstride == 4 * w) {      // tight: one long row
        pa.rows = 1;
        pa.row_px = 0;           // set below (64-bit)
        pa.upr = 1;
        pa.units = (static_cast<long long>(w) * h + 3) / 4;
        pa.vec = aligned16_ok;
    } else {
        pa.rows = h;
        pa.upr = (w + 3) / 4;
        pa.units = static_cast<long long>(pa.upr) * h;
        pa.vec = aligned16_ok && (sstride & 15) == 0;
    }
    pa.upr_magic = static_cast<uint32_t>((0x100000000ull + pa.upr - 1) / pa.upr);
    if (pa.rows > 1 && pa.units >= (0x100000000ll / pa.upr)) {
        set_error("image too large for the strided Analyze pass");
        return FNX_ERR_INVALID;
    }
    const long long row_px = pa.rows == 1 ? static_cast<long long>(w) * h : w;
    if (row_px > 0x7fffffffll) {
        set_error("image too large for Analyze");
        return FNX_ERR_INVALID;
    }
    pa.row_px = static_cast<int>(row_px);
    FNX_TRY(prof_begin(ctx));
    hipLaunchKernelGGL(analyze_pass_kernel, dim3(G, n), dim3(256), 0, ctx->stream, pa);
    FNX_HIP(hipGetLastError());
    FNX_TRY(prof_end(ctx));
    hipLaunchKernelGGL(analyze_finish_kernel, dim3(16, n), dim3(256), 0, ctx->stream, pa.hist_part, pa.bright_part,
                       pa.flag_part, G, d_res);
    FNX_HIP(hipGetLastError());

    SampArgs sa{};
    sa.src = src; sa.srcs = srcs; sa.sstride = sstride; sa.w = w; sa.h = h;
    const long long total = static_cast<long long>(w) * h;
    sa.color_step = total > 50000 ? total / 50000 : 1;                 // analyze.go:46-50
    sa.color_samples = (total + sa.color_step - 1) / sa.color_step;    // idx % step == 0, idx < total
    sa.color_blocks = static_cast<int>((sa.color_samples + 255) / 256);
    sa.cstep_y = static_cast<int>(std::fmax(1.0, std::ceil(static_cast<double>(h) / 100)));   // analyze.go:93-94
    sa.cstep_x = static_cast<int>(std::fmax(1.0, std::ceil(static_cast<double>(w) / 100)));
    sa.cny = (h + sa.cstep_y - 1) / sa.cstep_y;
    sa.cnx = (w + sa.cstep_x - 1) / sa.cstep_x;
    int eblocks = 0;
    if (w >= 3 && h >= 3) {                                            // analyze.go:146-151
        sa.estep_x = static_cast<int>(std::fmax(1.0, static_cast<double>(w) / 200));
        sa.estep_y = static_cast<int>(std::fmax(1.0, static_cast<double>(h) / 200));
        sa.enx = (w - 2 + sa.estep_x - 1) / sa.estep_x;
        sa.eny = (h - 2 + sa.estep_y - 1) / sa.estep_y;
        eblocks = (sa.enx * sa.eny + 255) / 256;
    }
    sa.hash = static_cast<unsigned long long *>(hash);
    sa.res = d_res;
    const int all_color_blocks = sa.color_blocks;
    constexpr int FIRST_COLOR_BLOCKS = 8;      // 2048 samples
    sa.color_blocks = std::min(all_color_blocks, FIRST_COLOR_BLOCKS);
    const int sblocks = sa.color_blocks + AN_CBLOCKS + eblocks;
    void *cp = nullptr;
    const size_t counts = (sizeof(uint32_t) * std::max(sblocks, all_color_blocks) * static_cast<size_t>(n) + 15) & ~size_t(15);
    FNX_TRY(scratch(ctx, SLOT_PARTIAL, counts + sizeof(double) * AN_CBLOCKS * static_cast<size_t>(n), &cp));
    sa.count_part = static_cast<uint32_t *>(cp);
    sa.var_part = reinterpret_cast<double *>(static_cast<char *>(cp) + counts);
    hipLaunchKernelGGL(analyze_sampled_kernel, dim3(sblocks, n), dim3(256), 0, ctx->stream, sa);
    FNX_HIP(hipGetLastError());
    hipLaunchKernelGGL(analyze_fold_kernel, dim3(n), dim3(256), 0, ctx->stream, sa.count_part, sblocks, sa.color_blocks,
                       sa.var_part, static_cast<long long>(sa.cnx) * sa.cny, static_cast<long long>(sa.enx) * sa.eny, 0, d_res);
    FNX_HIP(hipGetLastError());
    if (all_color_blocks > FIRST_COLOR_BLOCKS) {
        sa.color_k0 = static_cast<long long>(FIRST_COLOR_BLOCKS) * 256;
        sa.color_blocks = all_color_blocks - FIRST_COLOR_BLOCKS;
        sa.colors_only = 1;
        hipLaunchKernelGGL(analyze_sampled_kernel, dim3(sa.color_blocks, n), dim3(256), 0, ctx->stream, sa);
        FNX_HIP(hipGetLastError());
        hipLaunchKernelGGL(analyze_fold_kernel, dim3(n), dim3(256), 0, ctx->stream, sa.count_part, sa.color_blocks,
                           sa.color_blocks, sa.var_part, 0LL, 0LL, 1, d_res);
        FNX_HIP(hipGetLastError());
    }
    return FNX_OK;
}

// ------------------------------------------------------------------------------------
// isOpaque / isGrayscale (convert.go:66-84): flat scans of Pix, row padding included
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scan_flags_kernel(const uint8_t *pix, long long npx, int vec, uint32_t *out)
{
    uint32_t flags = 0;
    const long long units = (npx + 3) / 4;
    for (long long u = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; u < units; u += static_cast<long long>(gridDim.x) * 256) {
        const long long x = 4 * u;
        const uint8_t *p = pix + 4 * x;
        uint32_t v[4];
        int cnt = npx - x >= 4 ? 4 : static_cast<int>(npx - x);
        if (vec && cnt == 4) {
            const u32x4 q = ld16_stream(p);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
            for (int e = 0; e < cnt; e++) v[e] = *(g_u32 *)(p + 4 * e);
        }
        for (int e = 0; e < cnt; e++) {
            const uint32_t r = v[e] & 0xffu, g = (v[e] >> 8) & 0xffu, b = (v[e] >> 16) & 0xffu;
            flags |= ((v[e] >> 24) != 0xffu ? 1u : 0u) | ((r != g || g != b) ? 2u : 0u);
        }
    }
    // one atomic per wave at most, and none once the word already holds the wave's bits: device-scope atomics
    // on one word serialise across XCDs (~80 ns each), and a colour photograph sets bit 1 in every wave
    const uint32_t wave_flags = (__ballot(flags & 1u) ? 1u : 0u) | (__ballot(flags & 2u) ? 2u : 0u);
    if (wave_flags && (threadIdx.x & 63) == 0) {
        const uint32_t seen = __atomic_load_n(out, __ATOMIC_RELAXED);   // stale reads only cost an extra atomic
        if ((seen & wave_flags) != wave_flags) atomicOr(out, wave_flags);
    }
}

// d_flags (device, one uint32): bit 0 = some alpha != 255, bit 1 = some pixel not grey
int launch_scan_flags(fnx_ctx *ctx, const uint8_t *pix, size_t pix_len, uint32_t *d_flags)
{
    FNX_HIP(hipMemsetAsync(d_flags, 0, sizeof(uint32_t), ctx->stream));
    const long long npx = static_cast<long long>(pix_len / 4);
    if (npx == 0) return FNX_OK;
    const long long units = (npx + 3) / 4;
    const int blocks = static_cast<int>(std::min<long long>((units + 255) / 256, 4LL * ctx->num_cus));
    hipLaunchKernelGGL(scan_flags_kernel, dim3(blocks), dim3(256), 0, ctx->stream, pix, npx,
                       (reinterpret_cast<uintptr_t>(pix) & 15u) == 0 ? 1 : 0, d_flags);
    FNX_HIP(hipGetLastError());
    return FNX_OK;
}

// analyzeFormat's samples (convert.go:105-146): the pixels whose row-major index is a multiple of `step`, in order
__global__ __launch_bounds__(256) void sample_pixels_kernel(const uint8_t *src, int sstride, int w, long long total, long long step,
                                                            uint32_t *out, int nsamples)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nsamples) return;
    const long long idx = static_cast<long long>(k) * step;
    if (idx >= total) return;
    const int y = static_cast<int>(idx / w), x = static_cast<int>(idx - static_cast<long long>(y) * w);
    out[k] = *reinterpret_cast<const uint32_t *>(src + static_cast<size_t>(y) * sstride + 4 * static_cast<size_t>(x));
}

int launch_sample_pixels(fnx_ctx *ctx, const uint8_t *src, int sstride, int w, int h, long long step, uint32_t *d_out, int nsamples)
{
    if (nsamples <= 0) return FNX_OK;
    hipLaunchKernelGGL(sample_pixels_kernel, dim3((nsamples + 255) / 256), dim3(256), 0, ctx->stream, src, sstride, w,
                       static_cast<long long>(w) * h, step, d_out, nsamples);
    FNX_HIP(hipGetLastError());
    return FNX_OK;
}

}  // namespace fnx
