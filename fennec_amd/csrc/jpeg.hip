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
#include <vector>
#include "common.hpp"
#include "devutil.hpp"
#include "jpeg_idct.hpp"

namespace fnx {

struct YccArgs {
    const uint8_t *src;
    int sstride, w, h;
    uint8_t *yp, *cbp, *crp;     // Y: ys x 16 my; Cb, Cr: cs x 8 my
    int ys, cs, my;
    // r3: the source as a DECODED JPEG's planes (dy != nullptr): a pixel is toNRGBARef's (devutil.hpp's ycc_nrgba_px of
    // its samples), computed here instead of read from an image that fnx_jpeg_recompress would write only for this kernel
    // and the reference plane's box sums
    const uint8_t *dy, *dcb, *dcr;
    int dys, dcs, dxs, dysh;
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
        const int yy = min(py + j, a.h - 1);                                                     // toYCbCr clamps to the last row / column
        if (a.dy) {
            const uint8_t *yr = a.dy + static_cast<size_t>(yy) * a.dys;
            const size_t co = static_cast<size_t>(yy >> a.dysh) * a.dcs;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int xx = min(px + i, a.w - 1);
                ycc[j][i] = rgb_to_ycc(ycc_nrgba_px(yr[xx], a.dcb[co + (xx >> a.dxs)], a.dcr[co + (xx >> a.dxs)]));
            }
        } else {
            const uint8_t *row = a.src + static_cast<size_t>(yy) * a.sstride;
#pragma unroll
            for (int i = 0; i < 4; i++) ycc[j][i] = rgb_to_ycc(ld_px(row, min(px + i, a.w - 1)));
        }
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

// the same from a decoded JPEG's planes (ratio: image.YCbCrSubsampleRatio), the NRGBA image in between never written
int launch_jpeg_ycc_planes(fnx_ctx *ctx, const uint8_t *dy, int dys, const uint8_t *dcb, const uint8_t *dcr, int dcs, int ratio, int w, int h,
                           uint8_t *yp, uint8_t *cbp, uint8_t *crp)
{
    static const int xs[6] = {0, 1, 1, 0, 2, 2}, ysh[6] = {0, 0, 1, 1, 0, 1};
    YccArgs a{};
    int yh, chh;
    jpeg_plane_dims(w, h, &a.ys, &yh, &a.cs, &chh);
    a.w = w; a.h = h; a.yp = yp; a.cbp = cbp; a.crp = crp; a.my = yh / 16;
    a.dy = dy; a.dcb = dcb; a.dcr = dcr; a.dys = dys; a.dcs = dcs; a.dxs = xs[ratio]; a.dysh = ysh[ratio];
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


// ------------------------------------------------------------------------------------
// Baseline entropy coding on the device -- SURVEY 8(f)2, second slice: jpeg.Encode's FILE (io.go:157-169) from a
// device-resident NRGBA image, so that CompressBatch needs the host codec only to decode its source.
//
// What writer.go does after the quantiser: zig-zag order, DC differences per component, (run, size) symbols with the
// typical Huffman tables of ITU T.81 Annex K.3.3 (writer.go's theHuffmanSpec), value bits, 0xff stuffing, 1-padding of
// the last byte, and the header segments in writer.go's order (SOI, one DQT with both tables, SOF0, one DHT with four
// tables, SOS; no APP0).  A JPEG scan is one long bit string, but a block's code depends on the rest of the image only
// through the previous block's DC and through its own position in the string:
//   1. jpeg_coef_kernel   lane = block: FDCT + quantiser, coefficients in zig-zag order, coefficient-major
//   2. jpeg_code_kernel   lane = block: the block's bits into a staging strip (<= 52 words), its bit count
//   3. prefix sum         of the bit counts (two launches) = every block's position in the string
//   4. jpeg_pack_kernel   lane = block: the strip shifted into place (atomicOr into zeroed words, MSB first)
//   5. stuffing           0xff bytes per word, prefix sum, every byte to its final place with 0x00 behind each 0xff
// The host writes the header (589 bytes) and the EOI marker.  All integer work: byte-for-byte against the CPU
// restatement the tests hold, which is itself checked against libjpeg-turbo where it can be and is
// otherwise -- like the quantiser path -- unpinned against Go.
// ------------------------------------------------------------------------------------
__device__ __constant__ uint8_t c_zigpos[64] = {          // zig-zag position of natural index k
    0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30, 41, 43, 9,  11, 18, 24, 31, 40, 44, 53,
    10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};
static const uint8_t UNZIG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                  41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                  30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct CoefArgs {
    const uint8_t *in[3];     // Y, Cb, Cr planes (jpeg_ycc_kernel)
    int stride[3], nbx[3], nblocks[3];
    uint32_t q[2][64], magic[2][64];     // as BlockArgs
    int16_t *coef;            // [64][nblk]: coefficient zig of scan-order block b at zig * nblk + b
    int mx, nblk;
};

// lane = block (plane-major, so that loads coalesce); stores go to the block's place in SCAN order:
// MCU-major, Y0 Y1 Y2 Y3 Cb Cr inside an MCU (writer.go writeSOS)
__global__ __launch_bounds__(256) void jpeg_coef_kernel(CoefArgs a)
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
    const int sb = plane == 0 ? ((by >> 1) * a.mx + (bx >> 1)) * 6 + (by & 1) * 2 + (bx & 1) : (by * a.mx + bx) * 6 + 3 + plane;
#pragma unroll
    for (int k = 0; k < 64; k++) {
        const int32_t q = static_cast<int32_t>(a.q[t][k]);
        const uint32_t mag = static_cast<uint32_t>(b[k] < 0 ? -b[k] : b[k]) + 4u * static_cast<uint32_t>(q);
        const int32_t quot = static_cast<int32_t>(__umulhi(mag, a.magic[t][k]));     // writer.go div(b, 8 * q)
        a.coef[static_cast<size_t>(c_zigpos[k]) * a.nblk + sb] = static_cast<int16_t>(b[k] < 0 ? -quot : quot);
    }
}

constexpr int JPEG_STRIP = 52;            // words per block's staging strip: 20 + 63 * 26 bits at most

struct CodeArgs {
    const int16_t *coef;
    const uint32_t *lut;      // [4][256] length << 24 | code: luminance DC, luminance AC, chrominance DC, chrominance AC
    uint32_t *strip;          // [JPEG_STRIP][nblk]
    uint32_t *nbits;          // [nblk]
    int nblk;
};

__global__ __launch_bounds__(256) void jpeg_code_kernel(CodeArgs a)
{
    __shared__ uint32_t s_lut[4 * 256];
    for (int i = threadIdx.x; i < 4 * 256; i += 256) s_lut[i] = a.lut[i];
    __syncthreads();
    const int blk = blockIdx.x * 256 + threadIdx.x;
    if (blk >= a.nblk) return;
    const int j = blk % 6;
    const uint32_t *dc_lut = s_lut + (j < 4 ? 0 : 512), *ac_lut = dc_lut + 256;
    // the component's previous block in scan order
    const int prev = j == 0 ? blk - 3 : (j < 4 ? blk - 1 : blk - 6);
    const int32_t dc = a.coef[blk];
    const int32_t pd = prev >= 0 ? static_cast<int32_t>(a.coef[prev]) : 0;
    unsigned long long acc = 0;      // MSB-first bit accumulator: the low `nacc` bits are pending
    int nacc = 0, nw = 0;
    auto emit = [&](uint32_t bits, int n) {                // n <= 27
        acc = (acc << n) | bits;
        nacc += n;
        if (nacc >= 32) {
            a.strip[static_cast<size_t>(nw) * a.nblk + blk] = static_cast<uint32_t>(acc >> (nacc - 32));
            nw++;
            nacc -= 32;
        }
    };
    // emitHuffRLE: the (run, size) code, then `size` low bits of the value (value - 1 for negatives)
    auto rle = [&](const uint32_t *lut, int run, int32_t v) {
        const int32_t av = v < 0 ? -v : v, bv = v < 0 ? v - 1 : v;
        const int nb = 32 - __clz(av);                     // 0 for 0
        const uint32_t x = lut[(run << 4) | nb];
        const int len = static_cast<int>(x >> 24);
        emit(((x & 0x00ffffffu) << nb) | (static_cast<uint32_t>(bv) & ((1u << nb) - 1u)), len + nb);
    };
    rle(dc_lut, 0, dc - pd);
    int run = 0;
    for (int zig = 1; zig < 64; zig++) {
        const int32_t ac = a.coef[static_cast<size_t>(zig) * a.nblk + blk];
        if (ac == 0) {
            run++;
        } else {
            while (run > 15) {
                const uint32_t x = ac_lut[0xf0];
                emit(x & 0x00ffffffu, static_cast<int>(x >> 24));
                run -= 16;
            }
            rle(ac_lut, run, ac);
            run = 0;
        }
    }
    if (run > 0) {
        const uint32_t x = ac_lut[0x00];
        emit(x & 0x00ffffffu, static_cast<int>(x >> 24));
    }
    if (nacc > 0) a.strip[static_cast<size_t>(nw) * a.nblk + blk] = static_cast<uint32_t>(acc << (32 - nacc));   // left-aligned, zero tail
    a.nbits[blk] = static_cast<uint32_t>(32 * nw + nacc);
}

// ---- exclusive prefix sum of n uint32 into uint64: 2048 per workgroup, then the workgroups' totals ----
constexpr int SCAN_PER_WG = 2048;

__global__ __launch_bounds__(256) void scan_local_kernel(const uint32_t *in, unsigned long long *out, unsigned long long *totals, int n)
{
    __shared__ unsigned long long s_w[4];
    const int base = blockIdx.x * SCAN_PER_WG + threadIdx.x * 8;
    uint32_t v[8];
    unsigned long long sum = 0;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        v[e] = base + e < n ? in[base + e] : 0u;
        sum += v[e];
    }
    unsigned long long inc = sum;                           // inclusive scan of the lanes' sums inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_up(inc, off, 64);
        if ((threadIdx.x & 63) >= off) inc += o;
    }
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = inc;
    __syncthreads();
    unsigned long long woff = 0;
    for (int w = 0; w < static_cast<int>(threadIdx.x >> 6); w++) woff += s_w[w];
    unsigned long long run = woff + inc - sum;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        if (base + e < n) out[base + e] = run;
        run += v[e];
    }
    if (threadIdx.x == 255) totals[blockIdx.x] = woff + inc;
}

// out[i] += sum of the totals of the workgroups before i's; grand[0] = the sum of everything
__global__ __launch_bounds__(256) void scan_add_kernel(unsigned long long *out, const unsigned long long *totals, int n, unsigned long long *grand)
{
    // the totals before this workgroup's, summed by all 256 lanes (one lane walking up to ~100 of them was a chain of
    // dependent loads: 16 us per launch, four launches per JPEG item); integer sums, any order
    __shared__ unsigned long long s_part[4];
    unsigned long long mine = 0;
    for (int w = threadIdx.x; w < static_cast<int>(blockIdx.x); w += 256) mine += totals[w];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = mine;
    __syncthreads();
    const unsigned long long o = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1 && grand) grand[0] = o + totals[blockIdx.x];
    const int base = blockIdx.x * SCAN_PER_WG;
    for (int i = threadIdx.x; i < SCAN_PER_WG && base + i < n; i += 256) out[base + i] += o;
}

int launch_scan(fnx_ctx *ctx, const uint32_t *in, unsigned long long *out, unsigned long long *totals, int n, unsigned long long *grand)
{
    const int nwg = (n + SCAN_PER_WG - 1) / SCAN_PER_WG;
    hipLaunchKernelGGL(scan_local_kernel, dim3(nwg), dim3(256), 0, ctx->stream, in, out, totals, n);
    hipLaunchKernelGGL(scan_add_kernel, dim3(nwg), dim3(256), 0, ctx->stream, out, totals, n, grand);
    FNX_HIP(hipGetLastError());
    return FNX_OK;
}

struct PackArgs {
    const uint32_t *strip, *nbits;
    const unsigned long long *pos;     // bit position of every block
    uint32_t *bits;                    // the scan's bit string, 32 bits per word, MSB first; zeroed
    int nblk;
};

__global__ __launch_bounds__(256) void jpeg_pack_kernel(PackArgs a)
{
    const int blk = blockIdx.x * 256 + threadIdx.x;
    if (blk >= a.nblk) return;
    const unsigned long long p0 = a.pos[blk];
    const int nw = static_cast<int>((a.nbits[blk] + 31u) >> 5);
    const int s = static_cast<int>(p0 & 31u);
    size_t d = static_cast<size_t>(p0 >> 5);
    for (int k = 0; k < nw; k++, d++) {
        const uint32_t w = a.strip[static_cast<size_t>(k) * a.nblk + blk];
        atomicOr(&a.bits[d], w >> s);
        if (s) atomicOr(&a.bits[d + 1], w << (32 - s));
    }
}

struct StuffArgs {
    uint32_t *bits;                       // in: the bit string; the padding 1s are added here
    const unsigned long long *total;      // its length in bits
    uint32_t *ffcount;                    // [nwords] 0xff bytes per word (bytes past the end do not count)
    const unsigned long long *ffbefore;   // exclusive prefix sum of ffcount (second pass)
    uint8_t *out;                         // the entropy-coded segment
    int nwords;
};

__device__ __forceinline__ uint32_t padded_word(const StuffArgs &a, int i, unsigned long long tbits, int *nbytes)
{
    // writer.go: emit(0x7f, 7) -- the last byte's free bits become 1s; a byte-aligned string gets nothing
    const unsigned long long tb = (tbits + 7) >> 3;                   // bytes of the string
    uint32_t w = a.bits[i];
    const unsigned long long lo = static_cast<unsigned long long>(i) * 32;
    if (tbits > lo && tbits < lo + 32 && (tbits & 7u)) {
        const int used = static_cast<int>(tbits - lo);                // bits of this word that belong to the string
        const int upto = (used + 7) & ~7;                             // ... rounded up to the byte
        w |= (0xffffffffu >> used) & ~(upto == 32 ? 0u : (0xffffffffu >> upto));
    }
    const long long left = static_cast<long long>(tb) - static_cast<long long>(i) * 4;
    *nbytes = left >= 4 ? 4 : (left > 0 ? static_cast<int>(left) : 0);
    return w;
}

__global__ __launch_bounds__(256) void jpeg_ffcount_kernel(StuffArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.nwords) return;
    int nb;
    const uint32_t w = padded_word(a, i, a.total[0], &nb);
    uint32_t c = 0;
    for (int k = 0; k < nb; k++) c += ((w >> (24 - 8 * k)) & 0xffu) == 0xffu ? 1u : 0u;
    a.ffcount[i] = c;
}

__global__ __launch_bounds__(256) void jpeg_stuff_kernel(StuffArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.nwords) return;
    int nb;
    const uint32_t w = padded_word(a, i, a.total[0], &nb);
    uint8_t *o = a.out + static_cast<size_t>(i) * 4 + a.ffbefore[i];
    for (int k = 0; k < nb; k++) {
        const uint8_t b = static_cast<uint8_t>(w >> (24 - 8 * k));
        *o++ = b;
        if (b == 0xffu) *o++ = 0x00;
    }
}

// ITU T.81 Annex K.3.3, the order of writer.go's theHuffmanSpec: luminance DC, luminance AC, chrominance DC, chrominance AC
static const uint8_t HCOUNT[4][16] = {{0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0},
                                      {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 125},
                                      {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0},
                                      {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 119}};
static const uint8_t HDC[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t HAC_LUM[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81,
    0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18,
    0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
    0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
    0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5,
    0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t HAC_CHR[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08,
    0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25,
    0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47,
    0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74,
    0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97,
    0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4,
    0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t *HVAL[4] = {HDC, HAC_LUM, HDC, HAC_CHR};
static const int HNVAL[4] = {12, 162, 12, 162};

static void quant_tables(int quality, uint32_t (&q)[2][64], uint32_t (&magic)[2][64])
{
    if (quality < 1) quality = 1;
    if (quality > 100) quality = 100;
    const int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
    for (int t = 0; t < 2; t++)
        for (int k = 0; k < 64; k++) {
            int x = (static_cast<int>(t ? K2[k] : K1[k]) * scale + 50) / 100;
            x = x < 1 ? 1 : (x > 255 ? 255 : x);
            q[t][k] = static_cast<uint32_t>(x);
            magic[t][k] = static_cast<uint32_t>((1ull << 32) / (8ull * x)) + 1u;
        }
}

// the bytes jpeg.Encode writes before the entropy-coded segment
void jpeg_header(int w, int h, int quality, std::vector<uint8_t> &o)
{
    uint32_t q[2][64], magic[2][64];
    quant_tables(quality, q, magic);
    auto marker = [&](uint8_t m, int len) { o.push_back(0xff); o.push_back(m); o.push_back(uint8_t(len >> 8)); o.push_back(uint8_t(len & 0xff)); };
    o.push_back(0xff); o.push_back(0xd8);
    marker(0xdb, 2 + 2 * (1 + 64));
    for (int t = 0; t < 2; t++) {
        o.push_back(uint8_t(t));
        for (int zig = 0; zig < 64; zig++) o.push_back(uint8_t(q[t][UNZIG[zig]]));
    }
    marker(0xc0, 8 + 3 * 3);
    o.push_back(8);
    o.push_back(uint8_t(h >> 8)); o.push_back(uint8_t(h & 0xff));
    o.push_back(uint8_t(w >> 8)); o.push_back(uint8_t(w & 0xff));
    o.push_back(3);
    const uint8_t comp[9] = {1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1};
    o.insert(o.end(), comp, comp + 9);
    int dht = 2;
    for (int t = 0; t < 4; t++) dht += 1 + 16 + HNVAL[t];
    marker(0xc4, dht);
    const uint8_t tcth[4] = {0x00, 0x10, 0x01, 0x11};
    for (int t = 0; t < 4; t++) {
        o.push_back(tcth[t]);
        o.insert(o.end(), HCOUNT[t], HCOUNT[t] + 16);
        o.insert(o.end(), HVAL[t], HVAL[t] + HNVAL[t]);
    }
    const uint8_t sos[14] = {0xff, 0xda, 0x00, 0x0c, 0x03, 0x01, 0x00, 0x02, 0x11, 0x03, 0x11, 0x00, 0x3f, 0x00};
    o.insert(o.end(), sos, sos + 14);
}

static void huffman_luts(uint32_t *luts)        // [4][256]: length << 24 | code (canonical codes, T.81 Annex C)
{
    for (int t = 0; t < 4; t++) {
        uint32_t *lut = luts + 256 * t;
        for (int i = 0; i < 256; i++) lut[i] = 0;
        uint32_t code = 0;
        int k = 0;
        for (int len = 1; len <= 16; len++) {
            for (int j = 0; j < HCOUNT[t][len - 1]; j++) lut[HVAL[t][k++]] = (static_cast<uint32_t>(len) << 24) | code++;
            code <<= 1;
        }
    }
}

// The entropy-coded segment of jpeg.Encode(src, quality), in two steps with ONE small read-back between them (the
// length of the bit string decides how much the second step has to touch; sizing it for the worst case would scan
// 52 words per block instead of the ~3 a photograph needs):
//   jpeg_entropy_code   coefficients, per-block codes, their prefix sum; totals[0] (device) = bits of the string
//   jpeg_entropy_pack   (total_bits known on the host) the string packed, stuffed into `ecs`; totals[1] = 0xff bytes in it
// planes: jpeg_ycc_kernel's output.
int jpeg_entropy_code(fnx_ctx *ctx, int w, int h, int quality, const uint8_t *const planes[3], unsigned long long *totals)
{
    int ys, yh, cs, chh;
    jpeg_plane_dims(w, h, &ys, &yh, &cs, &chh);
    const int mx = ys / 16, my = yh / 16, nblk = mx * my * 6;
    const size_t strip_words = static_cast<size_t>(nblk) * JPEG_STRIP;
    auto al = [](size_t n) { return (n + 255) & ~size_t(255); };
    const size_t b_coef = al(sizeof(int16_t) * 64 * nblk), b_strip = al(4 * strip_words), b_nbits = al(4 * size_t(nblk)),
                 b_pos = al(8 * size_t(nblk)), b_tot = al(8 * (size_t(nblk) / SCAN_PER_WG + 2));
    void *sc = nullptr;
    FNX_TRY(scratch(ctx, SLOT_JPEG_ENC, b_coef + b_strip + b_nbits + b_pos + b_tot, &sc));
    unsigned char *p = static_cast<unsigned char *>(sc);
    int16_t *coef = reinterpret_cast<int16_t *>(p); p += b_coef;
    uint32_t *strip = reinterpret_cast<uint32_t *>(p); p += b_strip;
    uint32_t *nbits = reinterpret_cast<uint32_t *>(p); p += b_nbits;
    unsigned long long *pos = reinterpret_cast<unsigned long long *>(p); p += b_pos;
    unsigned long long *wgt = reinterpret_cast<unsigned long long *>(p);
    uint32_t hlut[4 * 256];
    huffman_luts(hlut);
    void *dlut = nullptr;
    FNX_TRY(upload_table(ctx, SLOT_JPEG_LUT, hlut, sizeof(hlut), &dlut));

    CoefArgs ca{};
    quant_tables(quality, ca.q, ca.magic);
    for (int pl = 0; pl < 3; pl++) {
        ca.in[pl] = planes[pl];
        ca.stride[pl] = pl ? cs : ys;
        ca.nbx[pl] = (pl ? cs : ys) / 8;
        ca.nblocks[pl] = ca.nbx[pl] * ((pl ? chh : yh) / 8);
    }
    ca.coef = coef; ca.mx = mx; ca.nblk = nblk;
    FNX_TRY(prof_begin(ctx, FNX_PROF_JPEG));
    hipLaunchKernelGGL(jpeg_coef_kernel, dim3((ca.nblocks[0] + 255) / 256, 3), dim3(256), 0, ctx->stream, ca);
    FNX_HIP(hipGetLastError());
    FNX_TRY(prof_end(ctx));
    CodeArgs co{coef, static_cast<const uint32_t *>(dlut), strip, nbits, nblk};
    hipLaunchKernelGGL(jpeg_code_kernel, dim3((nblk + 255) / 256), dim3(256), 0, ctx->stream, co);
    FNX_HIP(hipGetLastError());
    return launch_scan(ctx, nbits, pos, wgt, nblk, totals);
}

size_t jpeg_ecs_capacity(unsigned long long total_bits)
{
    return static_cast<size_t>((total_bits + 7) / 8) * 2 + 64;        // every byte could be 0xff
}

int jpeg_entropy_pack(fnx_ctx *ctx, int w, int h, unsigned long long total_bits, uint8_t *ecs, unsigned long long *totals)
{
    int ys, yh, cs, chh;
    jpeg_plane_dims(w, h, &ys, &yh, &cs, &chh);
    const int nblk = (ys / 16) * (yh / 16) * 6;
    const size_t strip_words = static_cast<size_t>(nblk) * JPEG_STRIP;
    auto al = [](size_t n) { return (n + 255) & ~size_t(255); };
    // phase-1 layout again (same slot, same sizes: nothing moves)
    const size_t b_coef = al(sizeof(int16_t) * 64 * nblk), b_strip = al(4 * strip_words), b_nbits = al(4 * size_t(nblk));
    void *sc = nullptr;
    FNX_TRY(scratch(ctx, SLOT_JPEG_ENC, b_coef + b_strip + b_nbits + al(8 * size_t(nblk)) + al(8 * (size_t(nblk) / SCAN_PER_WG + 2)), &sc));
    unsigned char *p = static_cast<unsigned char *>(sc) + b_coef;
    const uint32_t *strip = reinterpret_cast<const uint32_t *>(p); p += b_strip;
    const uint32_t *nbits = reinterpret_cast<const uint32_t *>(p); p += b_nbits;
    const unsigned long long *pos = reinterpret_cast<const unsigned long long *>(p);
    const size_t nw = static_cast<size_t>((total_bits + 31) / 32);
    if (nw > strip_words || nw >= (size_t(1) << 31)) {
        set_error("jpeg: bit string of %llu bits does not fit its blocks' strips", total_bits);
        return FNX_ERR_INVALID;
    }
    const int nwords = static_cast<int>(nw);
    const size_t b_bits = al(4 * (nw + 2)), b_ffc = al(4 * nw + 4), b_ffb = al(8 * nw + 8), b_tot = al(8 * (nw / SCAN_PER_WG + 2));
    void *s2 = nullptr;
    FNX_TRY(scratch(ctx, SLOT_JPEG_ENC2, b_bits + b_ffc + b_ffb + b_tot, &s2));
    p = static_cast<unsigned char *>(s2);
    uint32_t *bits = reinterpret_cast<uint32_t *>(p); p += b_bits;
    uint32_t *ffc = reinterpret_cast<uint32_t *>(p); p += b_ffc;
    unsigned long long *ffb = reinterpret_cast<unsigned long long *>(p); p += b_ffb;
    unsigned long long *wgt = reinterpret_cast<unsigned long long *>(p);
    FNX_HIP(hipMemsetAsync(bits, 0, 4 * (nw + 2), ctx->stream));
    PackArgs pa{strip, nbits, pos, bits, nblk};
    hipLaunchKernelGGL(jpeg_pack_kernel, dim3((nblk + 255) / 256), dim3(256), 0, ctx->stream, pa);
    if (nwords > 0) {
        StuffArgs sa{bits, totals, ffc, ffb, ecs, nwords};
        hipLaunchKernelGGL(jpeg_ffcount_kernel, dim3((nwords + 255) / 256), dim3(256), 0, ctx->stream, sa);
        FNX_TRY(launch_scan(ctx, ffc, ffb, wgt, nwords, totals + 1));
        hipLaunchKernelGGL(jpeg_stuff_kernel, dim3((nwords + 255) / 256), dim3(256), 0, ctx->stream, sa);
    }
    FNX_HIP(hipGetLastError());
    return FNX_OK;
}

}  // namespace fnx
