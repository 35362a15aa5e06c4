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
#include <type_traits>

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
// Analyze in TWO launches (round 5).  The staged form above is seven launches for one image (table memset, pass, finish,
// samples, fold, the rest of the colour samples, fold): 83 us per 4K call, of which the kernels are ~25.  Here every
// stage that does not need another stage's result runs side by side in one grid of 1024-lane workgroups --
//   blocks [0, G)            the full pass: histogram, brightness, flags          (analyze.go:57-72)
//   blocks [G, G + CB)       the sampled colour set, all samples                   (analyze.go:46-50, 73-77)
//   blocks [G + CB, .. + EB) the Sobel edge count on its grid                      (analyze.go:146-177)
// -- and a second launch of one workgroup per image (analyze_tail_kernel) folds the partial results and then takes the
// one stage that needs the mean: the contrast grid (analyze.go:93-109, <= 100 x 100 samples).  It writes the image's
// fnx_analysis and then a ready word; both may live in pinned host memory, which the caller watches.
// The colour table must be zero on entry: a ctx keeps two, and the launch that uses one zeroes the other for the next
// call (every lane clears a few words before it starts its own work).
// ------------------------------------------------------------------------------------
constexpr int AN1_T = 1024;
constexpr int AN1_HB = 16;                     // histogram workgroups of the tail launch (16 bins each)
constexpr int AN1_CW = 8;                      // + contrast workgroups: ONE CU's miss queue made 10 000 scattered samples a 20 us stage

struct An1Args {
    PassArgs2 p;                               // G = pass workgroups per image
    long long color_samples, color_step;
    int CB, EB;
    int cstep_x, cstep_y, cnx, cny;
    int estep_x, estep_y, enx, eny;
    unsigned long long *hash;                  // [n][AN_HASH_CAP], zero
    unsigned long long *hash_next;             // next_words words to clear
    long long next_words;
    uint32_t *count_part;                      // [n][CB + EB]
    unsigned *done;                            // [n], zero between launches
    fnx_analysis *res;                         // [n]
    double *var_part;                          // [n][AN1_CW]: the contrast workgroups' sums (the caller adds them, in order)
    uint32_t *ready;                           // [n][AN1_HB + AN1_CW]: 1 once that workgroup's part of the result is complete
};

__global__ __launch_bounds__(AN1_T) void analyze_one_kernel(An1Args a)
{
    __shared__ uint32_t s_hist[16 * AN_COPIES][256];             // 32 KB
    __shared__ double s_red[16];
    __shared__ uint32_t s_flag;
    const int tid = threadIdx.x, wave = tid >> 6, z = blockIdx.y, b = blockIdx.x;
    const int G = a.p.G, NB = gridDim.x;
    const uint8_t *src = a.p.srcs ? a.p.srcs[z] : a.p.src;
    {   // the other table, for the next call
        const long long nthreads = static_cast<long long>(NB) * gridDim.y * AN1_T;
        for (long long i = (static_cast<long long>(z) * NB + b) * AN1_T + tid; i < a.next_words; i += nthreads) a.hash_next[i] = 0ull;
    }
    if (tid == 0) s_flag = 0;
    // (the sampled stages' workgroups come LAST in dispatch order: first, they took CUs from the pass -- 21.4 against 18.9 us)
    const int SB = a.CB + a.EB;
    if (b < G) {                               // ---- the full pass
        for (int i = tid; i < 16 * AN_COPIES * 256; i += AN1_T) (&s_hist[0][0])[i] = 0;
        __syncthreads();
        uint32_t *hist = s_hist[wave * AN_COPIES + (tid & (AN_COPIES - 1))];
        double bright = 0.0;
        uint32_t flags = 0;
        // two 4-px units per trip (a lane walks ~8 of them at 4K: one load at a time was a chain of eight memory round trips)
        const long long ustride = static_cast<long long>(G) * AN1_T;
        auto locate = [&](long long u, const uint8_t *&p, long long &left) {
            int y = 0;
            long long c = u;
            if (a.p.rows > 1) {
                y = __umulhi(static_cast<uint32_t>(u), a.p.upr_magic);
                c = u - static_cast<long long>(y) * a.p.upr;
            }
            const long long x = 4 * c;
            p = src + static_cast<size_t>(y) * a.p.sstride + 4 * x;
            left = a.p.row_px - x;
        };
        auto count4 = [&](const u32x4 v) {
            const int b0 = an_pixel(v.x, bright, flags), b1 = an_pixel(v.y, bright, flags);
            const int b2 = an_pixel(v.z, bright, flags), b3 = an_pixel(v.w, bright, flags);
            an_count2(hist, b0, b1);
            an_count2(hist, b2, b3);
        };
        for (long long u = static_cast<long long>(b) * AN1_T + tid; u < a.p.units; u += 2 * ustride) {
            const uint8_t *p0, *p1;
            long long left0, left1 = 0;
            locate(u, p0, left0);
            const bool two = u + ustride < a.p.units;
            locate(two ? u + ustride : u, p1, left1);
            if (a.p.vec && left0 >= 4 && two && left1 >= 4) {
                const u32x4 v0 = ld16_stream(p0), v1 = ld16_stream(p1);
                count4(v0);
                count4(v1);
            } else {
                if (a.p.vec && left0 >= 4) count4(ld16_stream(p0));
                else for (int e = 0; e < 4 && e < left0; e++) atomicAdd(&hist[an_pixel(*(g_u32 *)(p0 + 4 * e), bright, flags)], 1u);
                if (two) {
                    if (a.p.vec && left1 >= 4) count4(ld16_stream(p1));
                    else for (int e = 0; e < 4 && e < left1; e++) atomicAdd(&hist[an_pixel(*(g_u32 *)(p1 + 4 * e), bright, flags)], 1u);
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {                  // fixed tree: lanes, then waves 0..15
            bright += __shfl_down(bright, off, 64);
            flags |= __shfl_down(flags, off, 64);
        }
        if ((tid & 63) == 0) { s_red[wave] = bright; atomicOr(&s_flag, flags); }
        __syncthreads();
        const size_t part = static_cast<size_t>(z) * G + b;
        if (tid < 256) {
            uint32_t cnt = 0;
#pragma unroll 8
            for (int k = 0; k < 16 * AN_COPIES; k++) cnt += s_hist[k][tid];
            a.p.hist_part[part * 256 + tid] = cnt;
        }
        if (tid == 0) {
            double t = 0.0;
            for (int k = 0; k < 16; k++) t += s_red[k];
            a.p.bright_part[part] = t;
            a.p.flag_part[part] = s_flag;
        }
    } else if (b - G < a.CB) {                 // ---- the sampled colour set
        const long long k = static_cast<long long>(b - G) * AN1_T + tid;
        bool fresh = false;
        if (k < a.color_samples) {
            const long long idx = k * a.color_step;
            const int y = static_cast<int>(idx / a.p.w), x = static_cast<int>(idx - static_cast<long long>(y) * a.p.w);
            const uint32_t px = *(g_u32 *)(src + static_cast<size_t>(y) * a.p.sstride + 4 * static_cast<size_t>(x));
            const unsigned long long key = (1ull << 32) | px;
            unsigned long long *tab = a.hash + static_cast<size_t>(z) * AN_HASH_CAP;
            uint32_t slot = (px * 2654435761u) >> (32 - 17);
            for (;;) {                                            // (see analyze_sampled_kernel)
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
        const int c = __syncthreads_count(fresh);
        if (tid == 0) a.count_part[static_cast<size_t>(z) * SB + (b - G)] = c;
    } else {                                   // ---- Sobel edge count
        const int s = (b - G - a.CB) * AN1_T + tid;
        bool edge = false;
        if (s < a.enx * a.eny) {
            const int iy = s / a.enx, ix = s - iy * a.enx;
            const int x = 1 + ix * a.estep_x, y = 1 + iy * a.estep_y;
            const int st = a.p.sstride;
            const double gx = sobel_lum(src, st, x + 1, y - 1) - sobel_lum(src, st, x - 1, y - 1) +
                              2 * sobel_lum(src, st, x + 1, y) - 2 * sobel_lum(src, st, x - 1, y) +
                              sobel_lum(src, st, x + 1, y + 1) - sobel_lum(src, st, x - 1, y + 1);
            const double gy = sobel_lum(src, st, x - 1, y + 1) - sobel_lum(src, st, x - 1, y - 1) +
                              2 * sobel_lum(src, st, x, y + 1) - 2 * sobel_lum(src, st, x, y - 1) +
                              sobel_lum(src, st, x + 1, y + 1) - sobel_lum(src, st, x + 1, y - 1);
            edge = sqrt(gx * gx + gy * gy) > 30.0;
        }
        const int c = __syncthreads_count(edge);
        if (tid == 0) a.count_part[static_cast<size_t>(z) * SB + (b - G)] = c;
    }
}

// The second (and last) launch of a call, AN1_HB + AN1_CW workgroups per image: folds what analyze_one_kernel's workgroups left
// (a kernel boundary orders the two: no counter, no fence -- a "last workgroup" ticket on one word costs ~80 ns per
// workgroup when every XCD contends for it, 30-80 us for this grid), then takes the stage that needs the mean.
__global__ __launch_bounds__(AN1_T) void analyze_tail_kernel(An1Args a)
{
    __shared__ unsigned long long s_c[64 * 16];
    __shared__ double s_red[16];
    __shared__ uint32_t s_flag;
    __shared__ int s_cnt[2];
    const int tid = threadIdx.x, wave = tid >> 6, z = blockIdx.y;
    const int G = a.p.G;
    const uint8_t *src = a.p.srcs ? a.p.srcs[z] : a.p.src;
    fnx_analysis *r = a.res + z;
    if (blockIdx.x < AN1_HB) {
        // histogram, bins [16 hb, 16 hb + 16): lane (bin, l) sums the partial histograms g = l, l + 64, ... (independent
        // loads: one workgroup walking all G x 256 counts in a dependent loop took 45 us), then the 64 lanes of a bin
        const int hb = blockIdx.x, bin = 16 * hb + (tid & 15), l = tid >> 4;
        unsigned long long cnt = 0;
#pragma unroll 4
        for (int g = l; g < G; g += 64) cnt += a.p.hist_part[(static_cast<size_t>(z) * G + g) * 256 + bin];
        s_c[l * 16 + (tid & 15)] = cnt;
        __syncthreads();
        if (tid < 16) {
            unsigned long long t = 0;
            for (int k = 0; k < 64; k++) t += s_c[k * 16 + tid];
            // write-through (system scope), drained below: the result may be pinned host memory.  (A system-scope release
            // fence instead -- an L2 write-back per wave -- made this launch 29 us.)
            __hip_atomic_store(&r->histogram[16 * hb + tid], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&a.ready[z * (AN1_HB + AN1_CW) + hb], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
#ifdef FNX_DEVELOP
    unsigned long long st[6];
    st[0] = wall_clock64();
#define AN_STAMP(k) st[k] = wall_clock64()
#else
#define AN_STAMP(k)
#endif
    if (tid == 0) { s_flag = 0; s_cnt[0] = 0; s_cnt[1] = 0; }
    __syncthreads();
    // brightness (fixed order), flags, counts
    double bsum = 0.0;
    uint32_t f = 0;
    for (int g = tid; g < G; g += AN1_T) {
        bsum += a.p.bright_part[static_cast<size_t>(z) * G + g];
        f |= a.p.flag_part[static_cast<size_t>(z) * G + g];
    }
    int colors = 0, edges = 0;
    for (int k = tid; k < a.CB + a.EB; k += AN1_T) {
        const int c = static_cast<int>(a.count_part[static_cast<size_t>(z) * (a.CB + a.EB) + k]);
        if (k < a.CB) colors += c;
        else edges += c;
    }
    auto block_sum = [&](double v) {                               // fixed tree over 1024 lanes; valid in every lane
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        __syncthreads();
        if ((tid & 63) == 0) s_red[wave] = v;
        __syncthreads();
        double t = 0.0;
        for (int k = 0; k < 16; k++) t += s_red[k];
        return t;
    };
    if (f) atomicOr(&s_flag, f);
    if (colors) atomicAdd(&s_cnt[0], colors);
    if (edges) atomicAdd(&s_cnt[1], edges);
    AN_STAMP(1);
    const double bright_sum = block_sum(bsum);
    AN_STAMP(2);
    // contrast: sum (lum - mean)^2 on the fixed grid
    const double mean = bright_sum / static_cast<double>(static_cast<long long>(a.p.w) * a.p.h);
    double v = 0.0;
    const int total = a.cnx * a.cny;
    // this workgroup's slice of the grid (every contrast workgroup has taken the same brightness sum, in the same order)
    const int cw = blockIdx.x - AN1_HB;
    const int per = (total + AN1_CW - 1) / AN1_CW, lo = cw * per, hi = min(total, lo + per);
    for (int s0 = lo + tid; s0 < hi; s0 += 2 * AN1_T) {             // two independent loads in flight per lane
        uint32_t px[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int s = s0 + e * AN1_T;
            const int iy = s < hi ? s / a.cnx : 0, ix = s < hi ? s - iy * a.cnx : 0;
            px[e] = *(g_u32 *)(src + static_cast<size_t>(iy * a.cstep_y) * a.p.sstride + 4 * static_cast<size_t>(ix * a.cstep_x));
        }
#pragma unroll
        for (int e = 0; e < 2; e++) {
            if (s0 + e * AN1_T < hi) {
                const double d = lum601(px[e]) - mean;
                v += d * d;
            }
        }
    }
    AN_STAMP(3);
    const double var_sum = block_sum(v);
    AN_STAMP(4);
    if (tid == 0) {
        auto put = [](auto *p, auto v) { __hip_atomic_store(p, static_cast<std::remove_reference_t<decltype(*p)>>(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
        put(&a.var_part[z * AN1_CW + cw], var_sum);
        if (cw == 0) {
            put(&r->bright_sum, bright_sum);
            put(&r->variance_sum, 0.0);                             // (the caller's sum of var_part)
            put(&r->sample_count, static_cast<long long>(total));
            put(&r->edge_count, static_cast<long long>(s_cnt[1]));
            put(&r->edge_total, static_cast<long long>(a.enx) * a.eny);
            put(&r->unique_colors, s_cnt[0]);
            put(&r->has_alpha, static_cast<int>(s_flag & 1u));
            put(&r->is_grayscale, (s_flag & 2u) ? 0 : 1);
            put(&r->pad, 0);
        }
        // write-through stores, drained, then the ready word
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef FNX_DEVELOP
        if (a.done) {                                               // development: phase stamps (10 ns units) instead of the edge total
            AN_STAMP(5);
            unsigned long long pk = 0;
            for (int k = 1; k < 6; k++) pk |= (((st[k] - st[0]) & 0xfffull) << (12 * (k - 1)));
            put(&r->edge_total, static_cast<long long>(pk));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#endif
    }
    if (tid == 0) __hip_atomic_store(&a.ready[z * (AN1_HB + AN1_CW) + AN1_HB + cw], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// n images -> res[n] (device or pinned host memory), ready[n * launch_analyze_ready_words()] (likewise; the caller sets
// them to 0 and watches them all turn 1)
int launch_analyze_ready_words() { return AN1_HB + AN1_CW; }
int launch_analyze_var_parts() { return AN1_CW; }

int launch_analyze_one(fnx_ctx *ctx, int n, const uint8_t *src, const uint8_t *const *srcs, int sstride, int w, int h,
                       bool aligned16_ok, fnx_analysis *res, double *var_part, uint32_t *ready)
{
    if (n <= 0) return FNX_OK;
    An1Args a{};
    PassArgs2 &pa = a.p;
    // pass workgroups per image: 4 x 1024 lanes per CU for one image, fewer in a batch
    const int G = std::max(32, std::min(256, (ctx->num_cus + n - 1) / n));
    void *hp = nullptr, *bp = nullptr;
    FNX_TRY(scratch(ctx, SLOT_TMP0, sizeof(uint32_t) * 256 * G * static_cast<size_t>(n), &hp));
    FNX_TRY(scratch(ctx, SLOT_TMP1, (sizeof(double) + sizeof(uint32_t)) * G * static_cast<size_t>(n) + 16, &bp));
    // the two colour tables: grown together, zeroed when (re)allocated; `cur` is zero by the invariant in the header
    const size_t tbytes = sizeof(unsigned long long) * AN_HASH_CAP * static_cast<size_t>(n);
    void *tabs[2];
    for (int i = 0; i < 2; i++) {
        const Slot sl = i ? SLOT_AN_HASH1 : SLOT_AN_HASH0;
        const void *before = ctx->slot[sl].p;
        FNX_TRY(scratch(ctx, sl, tbytes, &tabs[i]));
        if (tabs[i] != before) {
            FNX_HIP(hipMemsetAsync(tabs[i], 0, ctx->slot[sl].cap, ctx->stream));
            ctx->an_dirty[i] = 0;
        }
    }
    const int cur = ctx->an_cur, nxt = cur ^ 1;
    if (ctx->an_dirty[cur] != 0) {             // (not reached: the invariant; a failed launch could leave it so)
        FNX_HIP(hipMemsetAsync(tabs[cur], 0, ctx->slot[cur ? SLOT_AN_HASH1 : SLOT_AN_HASH0].cap, ctx->stream));
        ctx->an_dirty[cur] = 0;
    }
    a.hash = static_cast<unsigned long long *>(tabs[cur]);
    a.hash_next = static_cast<unsigned long long *>(tabs[nxt]);
    a.next_words = static_cast<long long>(AN_HASH_CAP) * ctx->an_dirty[nxt];

    pa.src = src; pa.srcs = srcs; pa.sstride = sstride; pa.w = w; pa.h = h;
    pa.hist_part = static_cast<uint32_t *>(hp);
    pa.bright_part = static_cast<double *>(bp);
    pa.flag_part = reinterpret_cast<uint32_t *>(pa.bright_part + static_cast<size_t>(G) * n);
    pa.G = G;
    if (sstride == 4 * w) {      // tight: one long row
        pa.rows = 1;
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

    const long long total = static_cast<long long>(w) * h;
    a.color_step = total > 50000 ? total / 50000 : 1;                  // analyze.go:46-50
    a.color_samples = (total + a.color_step - 1) / a.color_step;
    a.CB = static_cast<int>((a.color_samples + AN1_T - 1) / AN1_T);
    a.cstep_y = static_cast<int>(std::fmax(1.0, std::ceil(static_cast<double>(h) / 100)));   // analyze.go:93-94
    a.cstep_x = static_cast<int>(std::fmax(1.0, std::ceil(static_cast<double>(w) / 100)));
    a.cny = (h + a.cstep_y - 1) / a.cstep_y;
    a.cnx = (w + a.cstep_x - 1) / a.cstep_x;
    a.EB = 0;
    if (w >= 3 && h >= 3) {                                            // analyze.go:146-151
        a.estep_x = static_cast<int>(std::fmax(1.0, static_cast<double>(w) / 200));
        a.estep_y = static_cast<int>(std::fmax(1.0, static_cast<double>(h) / 200));
        a.enx = (w - 2 + a.estep_x - 1) / a.estep_x;
        a.eny = (h - 2 + a.estep_y - 1) / a.estep_y;
        a.EB = (a.enx * a.eny + AN1_T - 1) / AN1_T;
    }
    void *cp = nullptr;
    FNX_TRY(scratch(ctx, SLOT_PARTIAL, sizeof(uint32_t) * (a.CB + a.EB) * static_cast<size_t>(n) + 16, &cp));
    a.count_part = static_cast<uint32_t *>(cp);
    a.done = nullptr;
#ifdef FNX_DEVELOP
    { static const bool stamps = dev_env("FNX_AN_STAMPS") != nullptr; if (stamps) a.done = reinterpret_cast<unsigned *>(res); }
#endif
    a.res = res;
    a.var_part = var_part;
    a.ready = ready;
    note_route(ctx, FNX_PROF_MAIN, "analyze_one_kernel");
    FNX_TRY(prof_begin(ctx));
    hipLaunchKernelGGL(analyze_one_kernel, dim3(G + a.CB + a.EB, n), dim3(AN1_T), 0, ctx->stream, a);
    FNX_HIP(hipGetLastError());
    FNX_TRY(prof_end(ctx));
    hipLaunchKernelGGL(analyze_tail_kernel, dim3(AN1_HB + AN1_CW, n), dim3(AN1_T), 0, ctx->stream, a);
    FNX_HIP(hipGetLastError());
    ctx->an_dirty[nxt] = 0;
    ctx->an_dirty[cur] = n;
    ctx->an_cur = nxt;
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

// One launch, results straight to the host: every workgroup writes ITS flag bits (| 0x100: "reported") into its own word of
// pinned host memory, which the caller set to 0xffffffff and watches; the host ORs the words.  No memset, no copy, no
// stream synchronisation, and no device-side ticket either: a "last workgroup reports" counter on one word costs ~80 ns
// per workgroup when every XCD contends for it (57 us for 1024 workgroups, measured) against a 7 us scan.  A workgroup
// stops reading once its own lanes have seen both flags; an opaque image must be read to the end.  16-byte loads, four in
// flight per lane.
constexpr int SCAN_WGS_PER_CU = 2;
__global__ __launch_bounds__(256) void scan_flags_direct_kernel(const uint8_t *pix, long long npx, int vec, uint32_t *h_slots)
{
    __shared__ uint32_t s_flags;
    if (threadIdx.x == 0) s_flags = 0;
    __syncthreads();
    uint32_t flags = 0;
    const long long units = (npx + 3) / 4;
    const long long stride = static_cast<long long>(gridDim.x) * 256;
    auto test = [&](uint32_t v) {
        // grey <=> R == G == B: bytes 0, 1 equal bytes 1, 2
        flags |= ((v >> 24) != 0xffu ? 1u : 0u) | ((((v ^ (v >> 8)) & 0xffffu) != 0u) ? 2u : 0u);
    };
    for (long long u = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; u < units; u += 4 * stride) {
        if (vec && 4 * (u + 3 * stride) + 4 <= npx) {
            u32x4 q[4];
#pragma unroll
            for (int e = 0; e < 4; e++) q[e] = ld16_stream(pix + 16 * (u + e * stride));
#pragma unroll
            for (int e = 0; e < 4; e++) { test(q[e].x); test(q[e].y); test(q[e].z); test(q[e].w); }
        } else {
            for (int e = 0; e < 4; e++) {
                const long long uu = u + e * stride;
                if (uu >= units) break;
                const long long x = 4 * uu;
                const int cnt = npx - x >= 4 ? 4 : static_cast<int>(npx - x);
                for (int k = 0; k < cnt; k++) test(*(g_u32 *)(pix + 4 * (x + k)));
            }
        }
        if (((__ballot(flags & 1u) ? 1u : 0u) | (__ballot(flags & 2u) ? 2u : 0u)) == 3u) break;   // wave-uniform
    }
    const uint32_t wave_flags = (__ballot(flags & 1u) ? 1u : 0u) | (__ballot(flags & 2u) ? 2u : 0u);
    if (wave_flags && (threadIdx.x & 63) == 0) atomicOr(&s_flags, wave_flags);
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&h_slots[blockIdx.x], s_flags | 0x100u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// h_slots: *nslots words of pinned host memory (room for launch_scan_flags_slots(ctx) of them), each set to 0xffffffff by the
// caller; the launch fills words [0, *nslots)
int launch_scan_flags_slots(const fnx_ctx *ctx) { return SCAN_WGS_PER_CU * ctx->num_cus; }

int launch_scan_flags_direct(fnx_ctx *ctx, const uint8_t *pix, size_t pix_len, uint32_t *h_slots, int *nslots)
{
    const long long npx = static_cast<long long>(pix_len / 4);
    const long long units = (npx + 3) / 4;
    const int blocks = static_cast<int>(std::max<long long>(1, std::min<long long>((units + 1023) / 1024, launch_scan_flags_slots(ctx))));
    *nslots = blocks;
    hipLaunchKernelGGL(scan_flags_direct_kernel, dim3(blocks), dim3(256), 0, ctx->stream, pix, npx,
                       (reinterpret_cast<uintptr_t>(pix) & 15u) == 0 ? 1 : 0, h_slots);
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
