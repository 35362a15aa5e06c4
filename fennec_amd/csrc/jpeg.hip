// The JPEG quantisation round trip on the device -- SURVEY 8(f)2, first slice.
//
// compressJPEGOptimal (compress.go:21-87) encodes the source at a candidate quality, decodes the bytes again and
// scores the decoded image with SSIMFast, up to seven times per image: the host codec is ~99 % of CompressBatch.
// For the SEARCH only the decoded pixels matter, and they are a function of the quantised DCT coefficients alone --
// Huffman coding is lossless.  These kernels therefore do, per candidate quality, what Go's image/jpeg does to the
// pixels and nothing else: RGB -> YCbCr (color.RGBToYCbCr, 16.16 fixed point; non-opaque pixels premultiplied as
// color.NRGBA.RGBA() does), 4:2:0 with edge replication and (sum + 2) >> 2 chroma averaging (writer.go), the IJG
// integer FDCT (fdct.go), division by 8 q rounded half away from zero, multiplication by q (reader.go), the Chen-Wang
// integer IDCT (idct.go), level shift + clamp -- into the planes an *image.YCbCr would hold; convert.hip then makes
// the NRGBA image toNRGBARef would and ssim.hip scores it.  The winning quality is encoded ONCE by the real codec on
// the host.  All integer arithmetic: bit-exact against the CPU restatement the tests check it with,
// which -- like this file -- restates Go's standard library from the published algorithms it implements (the source
// is not under /root/reference): parity with Go is unpinned twice over.
#include "common.hpp"
#include "devutil.hpp"

namespace fnx {

struct YccArgs {
    const uint8_t *src;
    int sstride, w, h;
    uint8_t *yp, *cbp, *crp;     // Y: ys x 16 my; Cb, Cr: cs x 8 my
    int ys, cs, my;
};

// color.RGBToYCbCr on a (possibly premultiplied) pixel: Y | Cb << 8 | Cr << 16
__device__ __forceinline__ uint32_t rgb_to_ycc(uint32_t p)
{
    int32_t r = p & 0xffu, g = (p >> 8) & 0xffu, b = (p >> 16) & 0xffu;
    const uint32_t a = p >> 24;
    if (a != 0xffu) {                                  // color.NRGBA.RGBA(): c * 0x101 * A / 0xff, high byte kept
        r = static_cast<int32_t>((static_cast<uint32_t>(r) * 0x101u * a / 0xffu) >> 8);
        g = static_cast<int32_t>((static_cast<uint32_t>(g) * 0x101u * a / 0xffu) >> 8);
        b = static_cast<int32_t>((static_cast<uint32_t>(b) * 0x101u * a / 0xffu) >> 8);
    }
    const int32_t yy = (19595 * r + 38470 * g + 7471 * b + (1 << 15)) >> 16;
    int32_t cb = -11056 * r - 21712 * g + 32768 * b + (257 << 15);
    cb = ((static_cast<uint32_t>(cb) & 0xff000000u) == 0) ? cb >> 16 : ~(cb >> 31);
    int32_t cr = 32768 * r - 27440 * g - 5328 * b + (257 << 15);
    cr = ((static_cast<uint32_t>(cr) & 0xff000000u) == 0) ? cr >> 16 : ~(cr >> 31);
    return static_cast<uint32_t>(yy) | ((static_cast<uint32_t>(cb) & 0xffu) << 8) | ((static_cast<uint32_t>(cr) & 0xffu) << 16);
}

// a lane converts a 4 x 2 pixel patch of the MCU-padded image: 8 luma samples, 2 chroma pairs
__global__ __launch_bounds__(256) void jpeg_ycc_kernel(YccArgs a)
{
    const int px = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int py = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 2;
    if (px >= a.ys || py >= 16 * a.my) return;
    uint32_t ycc[2][4];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const uint8_t *row = a.src + static_cast<size_t>(min(py + j, a.h - 1)) * a.sstride;     // toYCbCr clamps to the last row / column
#pragma unroll
        for (int i = 0; i < 4; i++) ycc[j][i] = rgb_to_ycc(ld_px(row, min(px + i, a.w - 1)));
    }
#pragma unroll
    for (int j = 0; j < 2; j++)
        *reinterpret_cast<uint32_t *>(a.yp + static_cast<size_t>(py + j) * a.ys + px) =
            (ycc[j][0] & 0xffu) | ((ycc[j][1] & 0xffu) << 8) | ((ycc[j][2] & 0xffu) << 16) | ((ycc[j][3] & 0xffu) << 24);
    // writer.go scale(): (c00 + c01 + c10 + c11 + 2) >> 2
    uint32_t cb2 = 0, cr2 = 0;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const uint32_t sb = ((ycc[0][2 * i] >> 8) & 0xffu) + ((ycc[0][2 * i + 1] >> 8) & 0xffu) + ((ycc[1][2 * i] >> 8) & 0xffu) + ((ycc[1][2 * i + 1] >> 8) & 0xffu);
        const uint32_t sr = ((ycc[0][2 * i] >> 16) & 0xffu) + ((ycc[0][2 * i + 1] >> 16) & 0xffu) + ((ycc[1][2 * i] >> 16) & 0xffu) + ((ycc[1][2 * i + 1] >> 16) & 0xffu);
        cb2 |= ((sb + 2) >> 2) << (8 * i);
        cr2 |= ((sr + 2) >> 2) << (8 * i);
    }
    *reinterpret_cast<uint16_t *>(a.cbp + static_cast<size_t>(py / 2) * a.cs + px / 2) = static_cast<uint16_t>(cb2);
    *reinterpret_cast<uint16_t *>(a.crp + static_cast<size_t>(py / 2) * a.cs + px / 2) = static_cast<uint16_t>(cr2);
}

// ---- fdct.go (jfdctint.c's algorithm, 13-bit constants) and idct.go (Chen-Wang) on 8 values in registers ----
template <int PASS>
__device__ __forceinline__ void fdct8(int32_t &x0, int32_t &x1, int32_t &x2, int32_t &x3, int32_t &x4, int32_t &x5, int32_t &x6, int32_t &x7)
{
    constexpr int CB = 13, P1 = 2, SH = PASS == 1 ? CB - P1 : CB + P1;
    int32_t tmp0 = x0 + x7, tmp1 = x1 + x6, tmp2 = x2 + x5, tmp3 = x3 + x4;
    int32_t tmp10 = tmp0 + tmp3, tmp12 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp13 = tmp1 - tmp2;
    tmp0 = x0 - x7; tmp1 = x1 - x6; tmp2 = x2 - x5; tmp3 = x3 - x4;
    if (PASS == 1) {
        x0 = (tmp10 + tmp11 - 8 * 128) << P1;
        x4 = (tmp10 - tmp11) << P1;
    } else {
        tmp10 += 1 << (P1 - 1);
        x0 = (tmp10 + tmp11) >> P1;
        x4 = (tmp10 - tmp11) >> P1;
    }
    int32_t z1 = (tmp12 + tmp13) * 4433;
    z1 += 1 << (SH - 1);
    x2 = (z1 + tmp12 * 6270) >> SH;
    x6 = (z1 - tmp13 * 15137) >> SH;
    tmp10 = tmp0 + tmp3; tmp11 = tmp1 + tmp2; tmp12 = tmp0 + tmp2; tmp13 = tmp1 + tmp3;
    z1 = (tmp12 + tmp13) * 9633;
    z1 += 1 << (SH - 1);
    tmp0 *= 12299; tmp1 *= 25172; tmp2 *= 16819; tmp3 *= 2446;
    tmp10 *= -7373; tmp11 *= -20995; tmp12 *= -3196; tmp13 *= -16069;
    tmp12 += z1; tmp13 += z1;
    x1 = (tmp0 + tmp10 + tmp12) >> SH;
    x3 = (tmp1 + tmp11 + tmp13) >> SH;
    x5 = (tmp2 + tmp11 + tmp12) >> SH;
    x7 = (tmp3 + tmp10 + tmp13) >> SH;
}

constexpr int JW1 = 2841, JW2 = 2676, JW3 = 2408, JW5 = 1609, JW6 = 1108, JW7 = 565, JR2 = 181;

__device__ __forceinline__ void idct8_row(int32_t &s0, int32_t &s1, int32_t &s2, int32_t &s3, int32_t &s4, int32_t &s5, int32_t &s6, int32_t &s7)
{
    if ((s1 | s2 | s3 | s4 | s5 | s6 | s7) == 0) {       // all AC zero: dc << 3 everywhere (idct.go's shortcut; same bits either way is NOT guaranteed, so it is kept)
        const int32_t dc = s0 << 3;
        s0 = s1 = s2 = s3 = s4 = s5 = s6 = s7 = dc;
        return;
    }
    int32_t x0 = (s0 << 11) + 128, x1 = s4 << 11, x2 = s6, x3 = s2, x4 = s1, x5 = s7, x6 = s5, x7 = s3, x8;
    x8 = JW7 * (x4 + x5);
    x4 = x8 + (JW1 - JW7) * x4;
    x5 = x8 - (JW1 + JW7) * x5;
    x8 = JW3 * (x6 + x7);
    x6 = x8 - (JW3 - JW5) * x6;
    x7 = x8 - (JW3 + JW5) * x7;
    x8 = x0 + x1;
    x0 -= x1;
    x1 = JW6 * (x3 + x2);
    x2 = x1 - (JW2 + JW6) * x2;
    x3 = x1 + (JW2 - JW6) * x3;
    x1 = x4 + x6;
    x4 -= x6;
    x6 = x5 + x7;
    x5 -= x7;
    x7 = x8 + x3;
    x8 -= x3;
    x3 = x0 + x2;
    x0 -= x2;
    x2 = (JR2 * (x4 + x5) + 128) >> 8;
    x4 = (JR2 * (x4 - x5) + 128) >> 8;
    s0 = (x7 + x1) >> 8; s1 = (x3 + x2) >> 8; s2 = (x0 + x4) >> 8; s3 = (x8 + x6) >> 8;
    s4 = (x8 - x6) >> 8; s5 = (x0 - x4) >> 8; s6 = (x3 - x2) >> 8; s7 = (x7 - x1) >> 8;
}

__device__ __forceinline__ void idct8_col(int32_t &s0, int32_t &s1, int32_t &s2, int32_t &s3, int32_t &s4, int32_t &s5, int32_t &s6, int32_t &s7)
{
    int32_t y0 = (s0 << 8) + 8192, y1 = s4 << 8, y2 = s6, y3 = s2, y4 = s1, y5 = s7, y6 = s5, y7 = s3, y8;
    y8 = JW7 * (y4 + y5) + 4;
    y4 = (y8 + (JW1 - JW7) * y4) >> 3;
    y5 = (y8 - (JW1 + JW7) * y5) >> 3;
    y8 = JW3 * (y6 + y7) + 4;
    y6 = (y8 - (JW3 - JW5) * y6) >> 3;
    y7 = (y8 - (JW3 + JW5) * y7) >> 3;
    y8 = y0 + y1;
    y0 -= y1;
    y1 = JW6 * (y3 + y2) + 4;
    y2 = (y1 - (JW2 + JW6) * y2) >> 3;
    y3 = (y1 + (JW2 - JW6) * y3) >> 3;
    y1 = y4 + y6;
    y4 -= y6;
    y6 = y5 + y7;
    y5 -= y7;
    y7 = y8 + y3;
    y8 -= y3;
    y3 = y0 + y2;
    y0 -= y2;
    y2 = (JR2 * (y4 + y5) + 128) >> 8;
    y4 = (JR2 * (y4 - y5) + 128) >> 8;
    s0 = (y7 + y1) >> 14; s1 = (y3 + y2) >> 14; s2 = (y0 + y4) >> 14; s3 = (y8 + y6) >> 14;
    s4 = (y8 - y6) >> 14; s5 = (y0 - y4) >> 14; s6 = (y3 - y2) >> 14; s7 = (y7 - y1) >> 14;
}

struct BlockArgs {
    const uint8_t *in[3];     // Y, Cb, Cr planes before quantisation
    uint8_t *out[3];          // ... after the round trip
    int stride[3], nbx[3], nblocks[3];
    uint32_t q[2][64];        // quantiser steps, natural order: [0] luminance, [1] chrominance
    uint32_t magic[2][64];    // floor(2^32 / 8q) + 1: (|c| + 4q) / 8q by one v_mul_hi_u32 (|c| + 4q < 2^18, 8q <= 2040)
};

// one lane = one 8 x 8 block, all 64 samples in registers: no LDS, no transposes, no barriers
__global__ __launch_bounds__(256) void jpeg_block_kernel(BlockArgs a)
{
    const int plane = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.nblocks[plane]) return;
    const int t = plane ? 1 : 0;
    const int by = i / a.nbx[plane], bx = i - by * a.nbx[plane];
    const int stride = a.stride[plane];
    const uint8_t *ip = a.in[plane] + static_cast<size_t>(8 * by) * stride + 8 * bx;
    int32_t b[64];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const u32x2 v = *reinterpret_cast<const u32x2 *>(ip + static_cast<size_t>(r) * stride);
#pragma unroll
        for (int c = 0; c < 8; c++) b[8 * r + c] = static_cast<int32_t>(((c < 4 ? v.x : v.y) >> (8 * (c & 3))) & 0xffu);
    }
#pragma unroll
    for (int r = 0; r < 8; r++) fdct8<1>(b[8 * r], b[8 * r + 1], b[8 * r + 2], b[8 * r + 3], b[8 * r + 4], b[8 * r + 5], b[8 * r + 6], b[8 * r + 7]);
#pragma unroll
    for (int c = 0; c < 8; c++) fdct8<2>(b[c], b[8 + c], b[16 + c], b[24 + c], b[32 + c], b[40 + c], b[48 + c], b[56 + c]);
    // writer.go div(b, 8 * q) -- nearest, halves away from zero -- then reader.go's multiplication by q
#pragma unroll
    for (int k = 0; k < 64; k++) {
        const int32_t q = static_cast<int32_t>(a.q[t][k]);
        const uint32_t mag = static_cast<uint32_t>(b[k] < 0 ? -b[k] : b[k]) + 4u * static_cast<uint32_t>(q);
        const int32_t quot = static_cast<int32_t>(__umulhi(mag, a.magic[t][k]));
        b[k] = (b[k] < 0 ? -quot : quot) * q;
    }
#pragma unroll
    for (int r = 0; r < 8; r++) idct8_row(b[8 * r], b[8 * r + 1], b[8 * r + 2], b[8 * r + 3], b[8 * r + 4], b[8 * r + 5], b[8 * r + 6], b[8 * r + 7]);
#pragma unroll
    for (int c = 0; c < 8; c++) idct8_col(b[c], b[8 + c], b[16 + c], b[24 + c], b[32 + c], b[40 + c], b[48 + c], b[56 + c]);
    uint8_t *op = a.out[plane] + static_cast<size_t>(8 * by) * stride + 8 * bx;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        u32x2 v = {0, 0};
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const int32_t s = b[8 * r + c];
            const uint32_t u = s < -128 ? 0u : (s > 127 ? 255u : static_cast<uint32_t>(s + 128));      // reader.go: level shift, clip
            if (c < 4) v.x |= u << (8 * c); else v.y |= u << (8 * (c - 4));
        }
        *reinterpret_cast<u32x2 *>(op + static_cast<size_t>(r) * stride) = v;
    }
}

// writer.go, Encode: quality in [1, 100], scale = 5000 / q below 50, 200 - 2 q from 50, x = (x * scale + 50) / 100 in [1, 255]
static const uint8_t K1[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
                               14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
                               49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t K2[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
                               47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                               99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

void jpeg_plane_dims(int w, int h, int *ys, int *yh, int *cs, int *chh)
{
    const int mx = (w + 15) / 16, my = (h + 15) / 16;
    *ys = 16 * mx; *yh = 16 * my; *cs = 8 * mx; *chh = 8 * my;
}

// src (device NRGBA) -> the unquantised planes (quality-independent: once per source)
int launch_jpeg_ycc(fnx_ctx *ctx, const uint8_t *src, int sstride, int w, int h, uint8_t *yp, uint8_t *cbp, uint8_t *crp)
{
    YccArgs a{};
    int yh, chh;
    jpeg_plane_dims(w, h, &a.ys, &yh, &a.cs, &chh);
    a.src = src; a.sstride = sstride; a.w = w; a.h = h; a.yp = yp; a.cbp = cbp; a.crp = crp; a.my = yh / 16;
    hipLaunchKernelGGL(jpeg_ycc_kernel, dim3((a.ys / 4 + 63) / 64, (yh / 2 + 3) / 4), dim3(256), 0, ctx->stream, a);
    FNX_HIP(hipGetLastError());
    return FNX_OK;
}

// the planes at `quality`: fdct, quantise, dequantise, idct of every block (in -> out; in == out is allowed)
int launch_jpeg_blocks(fnx_ctx *ctx, int w, int h, int quality, const uint8_t *const in[3], uint8_t *const out[3])
{
    BlockArgs a{};
    int ys, yh, cs, chh;
    jpeg_plane_dims(w, h, &ys, &yh, &cs, &chh);
    if (quality < 1) quality = 1;
    if (quality > 100) quality = 100;
    const int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
    for (int t = 0; t < 2; t++)
        for (int k = 0; k < 64; k++) {
            int x = (static_cast<int>(t ? K2[k] : K1[k]) * scale + 50) / 100;
            x = x < 1 ? 1 : (x > 255 ? 255 : x);
            a.q[t][k] = static_cast<uint32_t>(x);
            a.magic[t][k] = static_cast<uint32_t>((1ull << 32) / (8ull * x)) + 1u;
        }
    for (int p = 0; p < 3; p++) {
        a.in[p] = in[p]; a.out[p] = out[p];
        a.stride[p] = p ? cs : ys;
        a.nbx[p] = (p ? cs : ys) / 8;
        a.nblocks[p] = a.nbx[p] * ((p ? chh : yh) / 8);
    }
    FNX_TRY(prof_begin(ctx, FNX_PROF_JPEG));
    hipLaunchKernelGGL(jpeg_block_kernel, dim3((a.nblocks[0] + 255) / 256, 3), dim3(256), 0, ctx->stream, a);
    FNX_HIP(hipGetLastError());
    return prof_end(ctx);
}

}  // namespace fnx
