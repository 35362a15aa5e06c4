// Prototype (experiments only): GaussianBlur R <= 6 with both separable passes on the i8 matrix pipe.
// Weights are 24-bit fixed point split into three signed base-256 digits; the integer sums are exact.
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include <type_traits>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// (hi * 256 + mid) * 256 + lo as two v_lshl_add_u32 (left alone the compiler builds two shifts and an add3)
__device__ __forceinline__ int comb3(int hi, int mid, int lo)
{
    int t = hi * 256 + mid;
    asm volatile("" : "+v"(t));
    return t * 256 + lo;
}

struct MArgs {
    const uint8_t *src;
    uint8_t *dst;
    size_t img_bytes;
    int sstride, dstride, w, h, tiles_x, tiles;
    int mode;
    const uint32_t *tab;   // BH[3][64][4] | BV[3][64][2] | seedH | seedV
};

__device__ __forceinline__ int xcd_tile(int bid, int total)
{
    int per = (total + 7) >> 3;
    int t = (bid & 7) * per + (bid >> 3);
    return ((bid >> 3) < per && t < total) ? t : -1;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <int TH, bool TRANS>
__global__ __launch_bounds__(256) void blur_mfma_kernel(MArgs a)
{
    constexpr int NRS = (TH + 12 + 15) / 16;             // H row sets of 16 staged rows
    constexpr int P = (16 * NRS) % 32 == 16 ? 16 * NRS : 16 * NRS + 16;   // pitch of a byte column: = 16 mod 32
    constexpr int NYS = TH / 16;
    constexpr int SP = 144;                               // pitch of a staged source row (7 chunks of 16 bytes used)
    constexpr int WT = 64 * P + 256;                      // bytes of T per wave
    constexpr int WS = 16 * SP;                           // bytes of the source stage per wave
    static_assert(TH % 16 == 0, "V row sets");
    __shared__ __attribute__((aligned(16))) uint8_t s_t[4 * WT];
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[4 * WS];
    __shared__ __attribute__((aligned(16))) uint8_t s_out[TRANS ? 2 * 16 * 272 : 16];

    const int tile = xcd_tile(blockIdx.x, a.tiles);
    if (tile < 0) return;
    const int z = blockIdx.y;
    const uint8_t *src = a.src + a.img_bytes * z;
    uint8_t *dst = a.dst + a.img_bytes * z;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int x0 = tx * 64, y0 = ty * TH;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, g = lane >> 4;
    const int xs = x0 + 16 * wave;                        // the wave's 16-px strip

    const v4i *tbh = reinterpret_cast<const v4i *>(a.tab);
    const long *tbv = reinterpret_cast<const long *>(a.tab + 3 * 64 * 4);
    const v4i bh2 = tbh[lane], bh1 = tbh[64 + lane], bh0 = tbh[128 + lane];
    const long bv2 = tbv[lane], bv1 = tbv[64 + lane], bv0 = tbv[128 + lane];
    const int seedH = a.tab[3 * 64 * 4 + 4 * 64 * 2], seedV = a.tab[3 * 64 * 4 + 4 * 64 * 2 + 1];
    const int seedHl = seedH + ((r & 3) == 3 ? 128 : 0);   // alpha lanes: a - 128 + 128
    const v4i sh = {seedHl, seedHl, seedHl, seedHl}, sv = {seedV, seedV, seedV, seedV};
    const v4i zero = {0, 0, 0, 0};
    // alpha lanes of the H sets (n % 4 == 3) keep byte 0 of their sums (the source alpha), the rest byte 3 (the rounded blur)
    const uint32_t sel01 = (r & 3) == 3 ? 0x0c0c0400u : 0x0c0c0703u, sel23 = (r & 3) == 3 ? 0x04000c0cu : 0x07030c0cu;

    // ---- H pass: set = 16 rows x 4 output px, A = source bytes (row r, 16-byte chunk of the 64-byte window) ----
    uint8_t *tw = s_t + wave * WT;
    uint8_t *st = s_stage + wave * WS;
    // no clamp anywhere in the strip's source window: rows y0-6 .. y0-6+16 NRS-1, columns xs-6 .. xs+21
    const bool inner = xs - 6 >= 0 && xs + 21 < a.w && y0 - 6 >= 0 && y0 - 6 + 16 * NRS <= a.h;
    const int lrow = lane >> 3, lch = (lane & 7) > 6 ? 6 : (lane & 7);
    auto hpass = [&](auto edge) {
        constexpr bool EDGE = decltype(edge)::value;
        const int loff0 = lrow * a.sstride + 16 * lch, loff1 = loff0 + 8 * a.sstride;   // inner: lane offsets from the row set's first byte
        auto hload = [&](int rs, u32x4 (&d)[2]) {
            if constexpr (!EDGE) {
                const uint8_t *sb = src + static_cast<ptrdiff_t>(y0 - 6 + 16 * rs) * a.sstride + 4 * static_cast<ptrdiff_t>(xs - 6);
                d[0] = *reinterpret_cast<const u32x4 *>(sb + loff0);
                d[1] = *reinterpret_cast<const u32x4 *>(sb + loff1);
            } else {
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const int y = clampi(y0 - 6 + 16 * rs + lrow + 8 * i, 0, a.h - 1);
                    const uint8_t *rowp = src + static_cast<size_t>(y) * a.sstride;
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        d[i][e] = *reinterpret_cast<const uint32_t *>(rowp + 4 * static_cast<size_t>(clampi(xs - 6 + 4 * lch + e, 0, a.w - 1)));
                }
            }
        };
        constexpr int PD = 2;                             // row sets in flight ahead of the one being filtered
        u32x4 ring[PD][2];
#pragma unroll
        for (int i = 0; i < PD; i++) if (i < NRS) hload(i, ring[i]);
#pragma unroll
        for (int rs = 0; rs < NRS; rs++) {
#pragma unroll
            for (int i = 0; i < 2; i++)
                *reinterpret_cast<u32x4 *>(st + (lrow + 8 * i) * SP + 16 * (lane & 7)) = ring[rs % PD][i] ^ 0x80808080u;
            if (rs + PD < NRS) hload(rs + PD, ring[rs % PD]);
            v4i c2[4], c1[4], c0[4];
#pragma unroll
            for (int qq = 0; qq < 4; qq++) {
                const v4i A = *reinterpret_cast<const v4i *>(st + r * SP + 16 * (qq + g));
                c2[qq] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bh2, zero, 0, 0, 0);
                c1[qq] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bh1, zero, 0, 0, 0);
                c0[qq] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bh0, sh, 0, 0, 0);
            }
#pragma unroll
            for (int qq = 0; qq < 4; qq++) {
                v4i u;
#pragma unroll
                for (int i = 0; i < 4; i++) u[i] = comb3(c2[qq][i], c1[qq][i], c0[qq][i]);
                const uint32_t t01 = __builtin_amdgcn_perm((uint32_t)u[1], (uint32_t)u[0], sel01);
                const uint32_t t23 = __builtin_amdgcn_perm((uint32_t)u[3], (uint32_t)u[2], sel23);
                *reinterpret_cast<uint32_t *>(tw + (16 * qq + r) * P + 64 * qq + 16 * rs + 4 * g) = t01 | t23;
            }
        }
    };
    if (!(a.mode & 1)) { if (inner) hpass(std::false_type{}); else hpass(std::true_type{}); }
    if (a.mode & 2) return;

    // ---- V pass: set = 16 output rows x 16 byte columns; A = staged bytes (byte column, 8 rows), B = weights ----
    const int m4 = r >> 2, mi = r & 3;
    const int x = xs + 4 * g;
    auto vload = [&](int ys, long (&A)[4], uint32_t (&al)[4]) {
#pragma unroll
        for (int j = 0; j < 4; j++) A[j] = *reinterpret_cast<const long *>(tw + (16 * m4 + 4 * j + mi) * P + 64 * m4 + 16 * ys + 8 * g);
#pragma unroll
        for (int j = 0; j < 4; j++) al[j] = *(tw + (16 * g + 4 * j + 3) * P + 64 * g + 16 * ys + r + 6);   // the centre row's alpha
    };
    long Ac[4], An[4];
    uint32_t alc[4], aln[4];
    vload(0, Ac, alc);
#pragma unroll 1
    for (int ys = 0; ys < NYS; ys++) {
        if (ys + 1 < NYS) vload(ys + 1, An, aln);
        v4i c2[4], c1[4], c0[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            c2[j] = __builtin_amdgcn_mfma_i32_16x16x32_i8(Ac[j], bv2, zero, 0, 0, 0);
            c1[j] = __builtin_amdgcn_mfma_i32_16x16x32_i8(Ac[j], bv1, zero, 0, 0, 0);
            c0[j] = __builtin_amdgcn_mfma_i32_16x16x32_i8(Ac[j], bv0, sv, 0, 0, 0);
        }
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int u0 = comb3(c2[j][0], c1[j][0], c0[j][0]);
            const int u1 = comb3(c2[j][1], c1[j][1], c0[j][1]);
            const int u2 = comb3(c2[j][2], c1[j][2], c0[j][2]);
            const uint32_t t01 = __builtin_amdgcn_perm((uint32_t)u1, (uint32_t)u0, 0x0c0c0703u);
            const uint32_t t23 = __builtin_amdgcn_perm(alc[j], (uint32_t)u2, 0x04030c0cu);
            o[j] = t01 | t23;
        }
        if constexpr (TRANS) {   // the workgroup's 16 rows x 256 bytes through LDS: every store instruction writes whole 128-byte lines
            uint8_t *so = s_out + (ys & 1) * 16 * 272;
            *reinterpret_cast<u32x4 *>(so + r * 272 + 64 * wave + 16 * g) = o;
            __syncthreads();
            const int row = tid >> 4, ch = tid & 15;
            const u32x4 ot = *reinterpret_cast<const u32x4 *>(so + row * 272 + 16 * ch);
            const int y = y0 + 16 * ys + row, xo = x0 + 4 * ch;
            if (y < a.h && (!(a.mode & 4) || ot[0] == 0x12345678u)) {
                uint8_t *dp = dst + static_cast<size_t>(y) * a.dstride + 4 * static_cast<size_t>(xo);
                if (xo + 3 < a.w) *reinterpret_cast<u32x4 *>(dp) = ot;
                else {
#pragma unroll
                    for (int e = 0; e < 4; e++) if (xo + e < a.w) reinterpret_cast<uint32_t *>(dp)[e] = ot[e];
                }
            }
        } else {
            const int y = y0 + 16 * ys + r;
            if (y < a.h && (!(a.mode & 4) || o[0] == 0x12345678u)) {
                uint8_t *dp = dst + static_cast<size_t>(y) * a.dstride + 4 * static_cast<size_t>(x);
                if (x + 3 < a.w) *reinterpret_cast<u32x4 *>(dp) = o;
                else {
#pragma unroll
                    for (int e = 0; e < 4; e++) if (x + e < a.w) reinterpret_cast<uint32_t *>(dp)[e] = o[e];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) { Ac[j] = An[j]; alc[j] = aln[j]; }
    }
}

// ------------------------------- host -------------------------------
static void digits(int64_t v, int d[3])
{
    for (int i = 0; i < 3; i++) {
        int64_t lo = ((v % 256) + 256) % 256;
        if (lo >= 128) lo -= 256;
        d[i] = (int)lo;
        v = (v - lo) / 256;
    }
    if (v != 0) { fprintf(stderr, "weight does not fit three signed digits\n"); exit(1); }
}

static std::vector<uint32_t> build_tab(const std::vector<double> &k, int R, std::vector<int64_t> &wq)
{
    const int NT = 2 * R + 1;
    wq.assign(NT, 0);
    int64_t sum = 0;
    for (int i = 0; i < NT; i++) { wq[i] = (int64_t)llround(k[i] * 16777216.0); sum += wq[i]; }
    wq[R] += 16777216 - sum;
    std::vector<uint32_t> tab(3 * 64 * 4 + 4 * 64 * 2 + 2, 0);
    int8_t *bh = reinterpret_cast<int8_t *>(tab.data());
    int8_t *bv = reinterpret_cast<int8_t *>(tab.data() + 3 * 64 * 4);
    const int off = 6 - R;
    for (int lane = 0; lane < 64; lane++) {
        const int n = lane & 15, kc = lane >> 4;
        for (int b = 0; b < 16; b++) {   // H: K index 16 kc + b = byte of the 64-byte window
            const int px = 4 * kc + b / 4, ch = b % 4, c = n % 4, pj = n / 4;
            const int t = px - pj - off;
            int d[3] = {0, 0, 0};
            if (ch == c && c < 3 && t >= 0 && t < NT) digits(wq[t], d);
            if (ch == c && c == 3 && px == pj + 6) d[0] = 1;   // alpha column: the centre pixel's alpha, as it is
            for (int l = 0; l < 3; l++) bh[((2 - l) * 64 + lane) * 16 + b] = (int8_t)d[l];   // table order: hi, mid, lo
        }
        for (int b = 0; b < 8; b++) {    // V: K index 8 kc + b = staged row relative to the set's first
            const int t = 8 * kc + b - n - off;
            int d[3] = {0, 0, 0};
            if (t >= 0 && t < NT) digits(wq[t], d);
            for (int l = 0; l < 3; l++) bv[((2 - l) * 64 + lane) * 8 + b] = (int8_t)d[l];
            bv[(3 * 64 + lane) * 8 + b] = (8 * kc + b == n + 6) ? 1 : 0;   // identity: staged row n + 6 = the output row
        }
    }
    tab[3 * 64 * 4 + 4 * 64 * 2] = 1u << 23;                    // H: staged bytes come out as (value ^ 0x80)
    tab[3 * 64 * 4 + 4 * 64 * 2 + 1] = (1u << 23) + (1u << 31); // V: plain bytes
    return tab;
}

static inline uint8_t clampF(double x)
{
    double t = std::trunc(x);
    if (std::fabs(x - t) >= 0.5) t += std::copysign(1.0, x);
    if (t < 0) t = 0;
    if (t > 255) t = 255;
    return (uint8_t)t;
}

static void ref_blur(const uint8_t *src, uint8_t *dst, int w, int h, const std::vector<double> &k, int R)
{
    std::vector<uint8_t> tmp((size_t)w * h * 4);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            double acc[3] = {0, 0, 0};
            for (int t = 0; t <= 2 * R; t++) {
                int sx = std::min(std::max(x + t - R, 0), w - 1);
                for (int c = 0; c < 3; c++) acc[c] = acc[c] + (double)src[((size_t)y * w + sx) * 4 + c] * k[t];
            }
            for (int c = 0; c < 3; c++) tmp[((size_t)y * w + x) * 4 + c] = clampF(acc[c]);
            tmp[((size_t)y * w + x) * 4 + 3] = src[((size_t)y * w + x) * 4 + 3];
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            double acc[3] = {0, 0, 0};
            for (int t = 0; t <= 2 * R; t++) {
                int sy = std::min(std::max(y + t - R, 0), h - 1);
                for (int c = 0; c < 3; c++) acc[c] = acc[c] + (double)tmp[((size_t)sy * w + x) * 4 + c] * k[t];
            }
            for (int c = 0; c < 3; c++) dst[((size_t)y * w + x) * 4 + c] = clampF(acc[c]);
            dst[((size_t)y * w + x) * 4 + 3] = src[((size_t)y * w + x) * 4 + 3];
        }
}

template <int TH, bool TRANS>
static void run(int w, int h, double sigma, int nimg, int reps)
{
    const int R = (int)std::ceil(3 * sigma);
    std::vector<double> k(2 * R + 1);
    double s = 0;
    for (int i = 0; i <= 2 * R; i++) { double x = i - R; k[i] = std::exp(-(x * x) / (2 * sigma * sigma)); s += k[i]; }
    for (auto &v : k) v /= s;
    std::vector<int64_t> wq;
    std::vector<uint32_t> tab = build_tab(k, R, wq);
    const size_t ib = (size_t)w * h * 4;
    std::vector<uint8_t> img(ib * nimg);
    uint32_t st = 12345;
    for (size_t i = 0; i < img.size(); i++) { st = st * 1664525u + 1013904223u; img[i] = (uint8_t)(st >> 24); }
    // smooth-ish content in image 0's left half so that rounding ties are not the only thing tested
    for (int y = 0; y < h; y++) for (int x = 0; x < w / 2; x++) for (int c = 0; c < 4; c++)
        img[((size_t)y * w + x) * 4 + c] = (uint8_t)((x * y + 3 * x + 7 * y * c + c * 31) % 256);
    uint8_t *dsrc, *ddst; uint32_t *dtab;
    hipMalloc(&dsrc, ib * nimg); hipMalloc(&ddst, ib * nimg); hipMalloc(&dtab, tab.size() * 4);
    hipMemcpy(dsrc, img.data(), ib * nimg, hipMemcpyHostToDevice);
    hipMemcpy(dtab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice);
    hipMemset(ddst, 0xcd, ib * nimg);
    MArgs a{};
    a.src = dsrc; a.dst = ddst; a.img_bytes = ib; a.sstride = 4 * w; a.dstride = 4 * w; a.w = w; a.h = h;
    a.tiles_x = (w + 63) / 64; a.tiles = a.tiles_x * ((h + TH - 1) / TH); a.tab = dtab;
    dim3 grid(8 * ((a.tiles + 7) / 8), nimg);
    blur_mfma_kernel<TH, TRANS><<<grid, 256>>>(a);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(e)); exit(1); }
    std::vector<uint8_t> out(ib), ref(ib);
    hipMemcpy(out.data(), ddst, ib, hipMemcpyDeviceToHost);
    if (nimg == 1 || reps >= 20) ref_blur(img.data(), ref.data(), w, h, k, R); else ref = out;
    size_t diff = 0, big = 0; int firstx = -1, firsty = -1;
    for (size_t i = 0; i < ib; i++) if (out[i] != ref[i]) {
        diff++;
        if (std::abs((int)out[i] - (int)ref[i]) > 1) { big++; if (firstx < 0) { firstx = (int)((i / 4) % w); firsty = (int)((i / 4) / w); } }
    }
    printf("TH=%d TRANS=%d %dx%d sigma=%.2f R=%d: %zu of %zu samples differ (%.5f %%), %zu by more than 1", TH, (int)TRANS, w, h, sigma, R, diff, ib, 100.0 * diff / ib, big);
    if (big) printf(" first at (%d,%d)", firstx, firsty);
    printf("\n");
    if (reps > 0) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 80; i++) blur_mfma_kernel<TH, TRANS><<<grid, 256>>>(a);
        hipEventRecord(e0);
        for (int i = 0; i < 5 * reps; i++) blur_mfma_kernel<TH, TRANS><<<grid, 256>>>(a);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1000.0 / (5 * reps) / nimg;
        printf("   %.2f us per image, %.0f MP/s, %.2f TB/s of 2S\n", us, (double)w * h / us, 2.0 * ib / us / 1e6);
        for (int mode : {1, 2, 4, 5}) {
            a.mode = mode;
            for (int i = 0; i < 30; i++) blur_mfma_kernel<TH, TRANS><<<grid, 256>>>(a);
            hipEventRecord(e0);
            for (int i = 0; i < 5 * reps; i++) blur_mfma_kernel<TH, TRANS><<<grid, 256>>>(a);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            printf("   mode %d (1: no H, 2: no V, 4: no store): %.2f us per image\n", mode, ms * 1000.0 / (5 * reps) / nimg);
        }
        a.mode = 0;
    }
    hipFree(dsrc); hipFree(ddst); hipFree(dtab);
}

int main(int argc, char **argv)
{
    if (argc > 1) { run<96, false>(3840, 2160, 2.0, 32, 5); return 0; }
    run<96, false>(640, 480, 2.0, 1, 0);
    run<112, true>(640, 480, 2.0, 1, 0);
    run<96, true>(641, 479, 1.5, 1, 0);
    run<96, false>(100, 50, 2.0, 1, 0);
    run<96, false>(3840, 2160, 2.0, 32, 20);
    run<96, true>(3840, 2160, 2.0, 32, 20);
    run<112, false>(3840, 2160, 2.0, 32, 20);
    run<112, true>(3840, 2160, 2.0, 32, 20);
    return 0;
}
