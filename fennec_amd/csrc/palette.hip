// applyPalette (targetsize.go:488-527) + palettedToNRGBA (targetsize.go:529-546) on gfx950.
// the per-pixel part of the PNG quantisation strategy (targetsize.go:180-206);
// medianCut stays with the caller (<= 100 000 samples, and its splits depend on Go's unstable
// sort.Slice -- not restated).
//
// Nearest palette entry by squared RGB distance, first minimum wins (strict `<`, :513).  The
// reference memoises per colour in a map; brute force gives the same answer.  Per entry i and
// pixel c:  dist = |c|^2 - 2 c.p_i + |p_i|^2, and -c.p_i = c.(255 - p_i) - 255*sum(c), so with the
// per-pixel constants dropped the argmin of  key_i = (|p_i|^2 + 2 c.(255 - p_i)) * 256 + i  is the
// reference's index: smaller distance first, then smaller index.  c.(255 - p_i) is one
// v_dot4_u32_u8 against the complemented entry, key_i one shift-add with the per-entry constant
// (|p_i|^2 * 256 + i, prepared on the host) and the running minimum one v_min_u32: 3 VALU
// instructions per entry per pixel, palette operands in SGPRs (uniform loads, 8 entries per
// trip; the table is padded with entries that cannot win).  Integer arithmetic: bit-exact.
//
// r5: that loop is AT the vector unit's rate -- 3 instructions x 256 entries per pixel = 150 us per 4K image, whatever the form
// (experiments/palette/pal_probe.hip: v_min3_u32, 8 px per lane, dot4 alone) -- so images of 256 k pixels and more ask fewer
// entries per pixel instead: a 32 x 32 x 32 grid over RGB (cells of 8 x 8 x 8 colours), per cell the CANDIDATES, i.e. every
// entry i whose smallest distance to the cell, mind_i, is not above U = min_j maxd_j, the smallest over the entries of the
// LARGEST distance to the cell.  For any colour x of the cell and the entry j that attains U: dist(x, p_i) >= mind_i > U >=
// dist(x, p_j) for every non-candidate i -- strictly farther than j, so it neither wins nor ties; the pixel's key over its
// cell's candidates has the same minimum as over the whole palette (same distance, same index: the key carries the original
// index).  pal_grid_kernel builds the table per call (64 cells per workgroup, four lanes per cell; ~5 us), a cell's record is
// 32 bytes -- count, then up to 31 candidates, unused slots repeating the first so that a wave runs to its longest list without
// a test per slot; a cell with more than 31 candidates (a palette crowded into one corner) is marked and its pixels walk the
// whole palette as before.  apply_palette_grid_kernel: 2-12 candidates per pixel instead of 256.
#include "common.hpp"
#include "devutil.hpp"

#include <cstdlib>
#include <vector>

namespace fnx {

struct PalArgs {
    const uint8_t *src;
    uint8_t *idx;            // w x h bytes (image.Paletted.Pix), may be null
    uint8_t *quant;          // NRGBA of palette[idx], alpha 255 (palettedToNRGBA), may be null
    int sstride, istride, qstride, w, h, n;
    int npad;                // n rounded up to a multiple of 8
    const uint32_t *rgb;     // [n] packed r | g<<8 | b<<16
    const uint32_t *comp;    // [npad] packed (255-r) | (255-g)<<8 | (255-b)<<16
    const uint32_t *konst;   // [npad] |p|^2 * 256 + i   (padding: 0xffffffff)
};

__global__ __launch_bounds__(256) void apply_palette_kernel(PalArgs a)
{
    __shared__ uint32_t s_pal[256];
    const int tid = threadIdx.x;
    if (tid < a.n) s_pal[tid] = a.rgb[tid] | 0xff000000u;
    __syncthreads();
    const int x = 4 * (blockIdx.x * 64 + (tid & 63));
    const int y = blockIdx.y * 4 + (tid >> 6);
    if (x >= a.w || y >= a.h) return;
    const uint8_t *row = a.src + static_cast<size_t>(y) * a.sstride;
    const int cnt = min(4, a.w - x);
    uint32_t c[4];
#pragma unroll
    for (int e = 0; e < 4; e++) c[e] = ld_px(row, min(x + e, a.w - 1)) & 0x00ffffffu;
    uint32_t best[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    const uint32_t *__restrict__ comp = a.comp;
    const uint32_t *__restrict__ konst = a.konst;
    for (int i0 = 0; i0 < a.npad; i0 += 8) {
        uint32_t p[8], k[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { p[j] = comp[i0 + j]; k[j] = konst[i0 + j]; }   // uniform: s_load_dwordx8
#pragma unroll
        for (int j = 0; j < 8; j++) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t dot = __builtin_amdgcn_udot4(c[e], p[j], 0u, false);
                best[e] = min(best[e], (dot << 9) + k[j]);
            }
        }
    }
    if (a.idx) {
        uint8_t *ip = a.idx + static_cast<size_t>(y) * a.istride + x;
        if (cnt == 4 && ((reinterpret_cast<uintptr_t>(ip) & 3u) == 0)) {
            *reinterpret_cast<uint32_t *>(ip) = (best[0] & 0xffu) | ((best[1] & 0xffu) << 8) | ((best[2] & 0xffu) << 16) | (best[3] << 24);
        } else {
            for (int e = 0; e < cnt; e++) ip[e] = static_cast<uint8_t>(best[e] & 0xffu);
        }
    }
    if (a.quant) {
        uint8_t *qp = a.quant + static_cast<size_t>(y) * a.qstride + 4 * static_cast<size_t>(x);
        for (int e = 0; e < cnt; e++) *reinterpret_cast<uint32_t *>(qp + 4 * e) = s_pal[best[e] & 0xffu];
    }
}


// ---- the grid form (r5): see the header ----
constexpr int PG_BITS = 5;                       // cells per axis = 32, 8 colours wide
constexpr int PG_CELLS = 1 << (3 * PG_BITS);
constexpr int PG_REC = 32;                       // bytes per cell: count (0xff: too many), 31 candidates
constexpr int PG_CAP = PG_REC - 1;

struct PalGridArgs {
    const uint32_t *rgb;     // [n] packed r | g<<8 | b<<16
    int n;
    uint8_t *grid;           // [PG_CELLS][PG_REC]
};

// smallest and largest squared distance from entry value p to the interval [lo, lo + 7] of one channel
__device__ __forceinline__ void pg_chan(int p, int lo, int &mn, int &mx)
{
    const int hi = lo + 7;
    const int below = lo - p, above = p - hi;
    const int d0 = max(max(below, above), 0);
    const int d1 = max(p - lo, hi - p);
    mn += d0 * d0;
    mx += d1 * d1;
}

__global__ __launch_bounds__(256) void pal_grid_kernel(PalGridArgs a)
{
    __shared__ uint32_t s_rgb[256];
    __shared__ int s_u[64];
    __shared__ uint32_t s_cnt[64];
    __shared__ uint8_t s_list[64][PG_REC];
    const int tid = threadIdx.x, lc = tid >> 2, q = tid & 3;
    if (tid < a.n) s_rgb[tid] = a.rgb[tid];
    if (tid < 64) { s_u[tid] = 0x7fffffff; s_cnt[tid] = 0u; }
    __syncthreads();
    const int cell = blockIdx.x * 64 + lc;
    const int r0 = (cell & 31) << 3, g0 = ((cell >> 5) & 31) << 3, b0 = ((cell >> 10) & 31) << 3;
    int u = 0x7fffffff;
    for (int i = q; i < a.n; i += 4) {
        const uint32_t p = s_rgb[i];
        int mn = 0, mx = 0;
        pg_chan(p & 0xffu, r0, mn, mx); pg_chan((p >> 8) & 0xffu, g0, mn, mx); pg_chan((p >> 16) & 0xffu, b0, mn, mx);
        u = min(u, mx);
    }
    atomicMin(&s_u[lc], u);
    __syncthreads();
    u = s_u[lc];
    for (int i = q; i < a.n; i += 4) {
        const uint32_t p = s_rgb[i];
        int mn = 0, mx = 0;
        pg_chan(p & 0xffu, r0, mn, mx); pg_chan((p >> 8) & 0xffu, g0, mn, mx); pg_chan((p >> 16) & 0xffu, b0, mn, mx);
        if (mn <= u) {
            const uint32_t k = atomicAdd(&s_cnt[lc], 1u);
            if (k < static_cast<uint32_t>(PG_CAP)) s_list[lc][1 + k] = static_cast<uint8_t>(i);
        }
    }
    __syncthreads();
    const uint32_t cnt = s_cnt[lc];                       // >= 1: the entry that attains U is its own candidate
    const uint8_t first = s_list[lc][1];
    // the record, 8 bytes per lane: slots past the list repeat its first entry
    uint32_t w[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        uint32_t v = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int slot = 8 * q + 4 * k + e;           // byte of the record
            uint32_t byte;
            if (slot == 0) byte = cnt > static_cast<uint32_t>(PG_CAP) ? 0xffu : cnt;
            else byte = static_cast<uint32_t>(slot) <= cnt ? s_list[lc][slot] : first;
            v |= byte << (8 * e);
        }
        w[k] = v;
    }
    *reinterpret_cast<u32x2 *>(a.grid + static_cast<size_t>(cell) * PG_REC + 8 * q) = (u32x2){w[0], w[1]};
}

struct PalGridApply {
    PalArgs p;
    const uint8_t *grid;
    const uint32_t *pairs;   // [256] x {complemented entry, |p|^2 * 256 + i}; past n: {0, 0xffffffff}
};

__global__ __launch_bounds__(256) void apply_palette_grid_kernel(PalGridApply ga)
{
    const PalArgs &a = ga.p;
    __shared__ uint32_t s_pal[256];
    __shared__ u32x2 s_pair[256];
    const int tid = threadIdx.x;
    if (tid < a.n) s_pal[tid] = a.rgb[tid] | 0xff000000u;
    s_pair[tid] = *reinterpret_cast<const u32x2 *>(ga.pairs + 2 * tid);
    __syncthreads();
    const int x = 4 * (blockIdx.x * 64 + (tid & 63));
    const int y = blockIdx.y * 4 + (tid >> 6);
    if (x >= a.w || y >= a.h) return;
    const uint8_t *row = a.src + static_cast<size_t>(y) * a.sstride;
    const int cnt = min(4, a.w - x);
    uint32_t c[4], best[4];
    u32x4 l0[4];
    uint32_t most = 0;
    bool over = false;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        c[e] = ld_px(row, min(x + e, a.w - 1)) & 0x00ffffffu;
        const uint32_t cell = ((c[e] >> 3) & 31u) | (((c[e] >> 11) & 31u) << 5) | (((c[e] >> 19) & 31u) << 10);
        l0[e] = *reinterpret_cast<const u32x4 *>(ga.grid + static_cast<size_t>(cell) * PG_REC);
        best[e] = 0xffffffffu;
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const uint32_t n = l0[e].x & 0xffu;
        over = over || n == 0xffu;
        most = max(most, n == 0xffu ? 0u : n);
    }
    auto visit = [&](int e, uint32_t idx) {
        const u32x2 pr = s_pair[idx];
        best[e] = min(best[e], (__builtin_amdgcn_udot4(c[e], pr.x, 0u, false) << 9) + pr.y);
    };
    // slots 1..15 live in the first 16 bytes; the wave runs to its longest list
    const bool m3 = __builtin_amdgcn_ballot_w64(most > 3u) != 0, m7 = __builtin_amdgcn_ballot_w64(most > 7u) != 0,
               m15 = __builtin_amdgcn_ballot_w64(most > 15u) != 0;   // (over the lanes that hold pixels)
#pragma unroll
    for (int e = 0; e < 4; e++) {
        visit(e, (l0[e].x >> 8) & 0xffu); visit(e, (l0[e].x >> 16) & 0xffu); visit(e, l0[e].x >> 24);
    }
    if (m3) {
#pragma unroll
        for (int e = 0; e < 4; e++) { visit(e, l0[e].y & 0xffu); visit(e, (l0[e].y >> 8) & 0xffu); visit(e, (l0[e].y >> 16) & 0xffu); visit(e, l0[e].y >> 24); }
    }
    if (m7) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
            visit(e, l0[e].z & 0xffu); visit(e, (l0[e].z >> 8) & 0xffu); visit(e, (l0[e].z >> 16) & 0xffu); visit(e, l0[e].z >> 24);
            visit(e, l0[e].w & 0xffu); visit(e, (l0[e].w >> 8) & 0xffu); visit(e, (l0[e].w >> 16) & 0xffu); visit(e, l0[e].w >> 24);
        }
    }
    if (m15) {                                      // slots 16..31: the record's second half
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t cell = ((c[e] >> 3) & 31u) | (((c[e] >> 11) & 31u) << 5) | (((c[e] >> 19) & 31u) << 10);
            const u32x4 l1 = *reinterpret_cast<const u32x4 *>(ga.grid + static_cast<size_t>(cell) * PG_REC + 16);
#pragma unroll
            for (int k = 0; k < 4; k++) { visit(e, l1[k] & 0xffu); visit(e, (l1[k] >> 8) & 0xffu); visit(e, (l1[k] >> 16) & 0xffu); visit(e, l1[k] >> 24); }
        }
    }
    if (__builtin_amdgcn_ballot_w64(over)) {               // a crowded cell: its pixels ask the whole palette
#pragma unroll
        for (int e = 0; e < 4; e++)
            if ((l0[e].x & 0xffu) == 0xffu)
                for (int i = 0; i < a.n; i++) visit(e, static_cast<uint32_t>(i));
    }
    if (a.idx) {
        uint8_t *ip = a.idx + static_cast<size_t>(y) * a.istride + x;
        if (cnt == 4 && ((reinterpret_cast<uintptr_t>(ip) & 3u) == 0)) {
            *reinterpret_cast<uint32_t *>(ip) = (best[0] & 0xffu) | ((best[1] & 0xffu) << 8) | ((best[2] & 0xffu) << 16) | (best[3] << 24);
        } else {
            for (int e = 0; e < cnt; e++) ip[e] = static_cast<uint8_t>(best[e] & 0xffu);
        }
    }
    if (a.quant) {
        uint8_t *qp = a.quant + static_cast<size_t>(y) * a.qstride + 4 * static_cast<size_t>(x);
        for (int e = 0; e < cnt; e++) *reinterpret_cast<uint32_t *>(qp + 4 * e) = s_pal[best[e] & 0xffu];
    }
}

// form "palette_grid" = "0": every image walks the whole palette (A/B, tests); "1": every image takes the grid
static int palette_grid_mode(const fnx_ctx *ctx)
{
    const char *e = form_value(ctx, FORM_PALETTE_GRID);
    return e ? atoi(e) : -1;
}

// palette: n x 4 bytes r,g,b,a on the host (a must be 255: checked by the caller)
int launch_apply_palette(fnx_ctx *ctx, const uint8_t *src, int sstride, int w, int h, const uint8_t *palette, int n,
                         uint8_t *idx, int istride, uint8_t *quant, int qstride)
{
    if (w <= 0 || h <= 0) return FNX_OK;
    const int mode = palette_grid_mode(ctx);
    if (mode == 1 || (mode != 0 && static_cast<long>(w) * h >= 262144L)) {
        std::vector<uint32_t> tab(256 + 512, 0u);                        // rgb[256] | pairs[256][2]
        for (int i = 0; i < 256; i++) {
            tab[256 + 2 * i + 1] = 0xffffffffu;
            if (i >= n) continue;
            const uint32_t r = palette[4 * i], g = palette[4 * i + 1], b = palette[4 * i + 2];
            tab[i] = r | (g << 8) | (b << 16);
            tab[256 + 2 * i] = (255u - r) | ((255u - g) << 8) | ((255u - b) << 16);
            tab[256 + 2 * i + 1] = (r * r + g * g + b * b) * 256u + static_cast<uint32_t>(i);
        }
        void *d = nullptr, *grid = nullptr;
        FNX_TRY(upload_table(ctx, SLOT_TABLE0, tab.data(), sizeof(uint32_t) * tab.size(), &d));
        FNX_TRY(scratch(ctx, SLOT_TMP1, static_cast<size_t>(PG_CELLS) * PG_REC, &grid));
        PalGridArgs ba{};
        ba.rgb = static_cast<const uint32_t *>(d); ba.n = n; ba.grid = static_cast<uint8_t *>(grid);
        hipLaunchKernelGGL(pal_grid_kernel, dim3(PG_CELLS / 64), dim3(256), 0, ctx->stream, ba);
        PalGridApply ga{};
        ga.p.src = src; ga.p.idx = idx; ga.p.quant = quant;
        ga.p.sstride = sstride; ga.p.istride = istride; ga.p.qstride = qstride; ga.p.w = w; ga.p.h = h; ga.p.n = n;
        ga.p.rgb = ba.rgb;
        ga.grid = ba.grid;
        ga.pairs = ba.rgb + 256;
        hipLaunchKernelGGL(apply_palette_grid_kernel, dim3((w + 255) / 256, (h + 3) / 4), dim3(256), 0, ctx->stream, ga);
        FNX_HIP(hipGetLastError());
        return FNX_OK;
    }
    const int npad = (n + 7) & ~7;
    std::vector<uint32_t> tab(3 * static_cast<size_t>(npad), 0u);
    for (int i = 0; i < npad; i++) {
        if (i >= n) { tab[2 * npad + i] = 0xffffffffu; continue; }       // padding never beats a real entry
        const uint32_t r = palette[4 * i], g = palette[4 * i + 1], b = palette[4 * i + 2];
        tab[i] = r | (g << 8) | (b << 16);
        tab[npad + i] = (255u - r) | ((255u - g) << 8) | ((255u - b) << 16);
        tab[2 * npad + i] = (r * r + g * g + b * b) * 256u + static_cast<uint32_t>(i);
    }
    void *d = nullptr;
    FNX_TRY(upload_table(ctx, SLOT_TABLE0, tab.data(), sizeof(uint32_t) * tab.size(), &d));
    PalArgs a{};
    a.src = src; a.idx = idx; a.quant = quant;
    a.sstride = sstride; a.istride = istride; a.qstride = qstride; a.w = w; a.h = h; a.n = n;
    a.npad = npad;
    a.rgb = static_cast<const uint32_t *>(d);
    a.comp = a.rgb + npad;
    a.konst = a.rgb + 2 * npad;
    hipLaunchKernelGGL(apply_palette_kernel, dim3((w + 255) / 256, (h + 3) / 4), dim3(256), 0, ctx->stream, a);
    FNX_HIP(hipGetLastError());
    return FNX_OK;
}

}  // namespace fnx
