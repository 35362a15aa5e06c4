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
#include "common.hpp"
#include "devutil.hpp"

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

// palette: n x 4 bytes r,g,b,a on the host (a must be 255: checked by the caller)
int launch_apply_palette(fnx_ctx *ctx, const uint8_t *src, int sstride, int w, int h, const uint8_t *palette, int n,
                         uint8_t *idx, int istride, uint8_t *quant, int qstride)
{
    if (w <= 0 || h <= 0) return FNX_OK;
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
