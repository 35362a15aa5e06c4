// Device-side helpers shared by the kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace fnx {

// explicit global-address-space views: pointers fetched from pointer tables are generic to
// the compiler and would otherwise be accessed with flat_load/flat_store
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// 16-byte accesses at PIXEL alignment (4 bytes): image views may be pitched and windows start at any pixel, so the
// global views say aligned(4) -- gfx950's global_load / store_dwordx4 need dword alignment only, and without the
// attribute the compiler would be entitled to assume 16
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));
typedef __attribute__((address_space(1))) const u32x4_a4 g_u32x4;
typedef __attribute__((address_space(1))) u32x4_a4 g_u32x4w;
typedef __attribute__((address_space(1))) const uint32_t g_u32;
typedef __attribute__((address_space(1))) uint32_t g_u32w;

// clampF (convert.go:149-158): math.Round (ties away from zero) then clamp to [0,255].
__device__ __forceinline__ uint32_t clampF_dev(double x)
{
    // Same value as int64(math.Round(x)) clamped, without the f64->i64 conversion and libm round():
    // t = trunc(x); x - t is exact; ties (|x - t| == 0.5) and beyond go away from zero.
    double t = trunc(x);
    if (fabs(x - t) >= 0.5) t += copysign(1.0, x);
    // int64(NaN) and int64(|x| >= 2^63) are 0x8000000000000000 on amd64 -> negative -> 0
    if (!(fabs(t) < 9223372036854775808.0)) return 0u;
    t = fmin(fmax(t, 0.0), 255.0);
    return static_cast<uint32_t>(static_cast<int>(t));
}

// clampF in three instructions for the exact (fp64) resize loops: min(u32(trunc(x + pred(0.5))), 255).
// Why it equals clampF (convert.go:149-158) for every finite x below 2^63 (the CPU suite replays it in exact rationals):
//  * x < 0 or NaN: math.Round gives -0 / a negative / int64(NaN) < 0 -> 0; here x + c < 0.5 truncates to 0 or is
//    negative / NaN, and v_cvt_u32_f64 saturates both to 0.
//  * 0 <= x, n = floor(x), u = ulp(x), c = 0.5 - 2^-54.  x >= n + 0.5: x + c >= n + 1 - 2^-54, which for n >= 1
//    lies closer to n + 1 than to the double below it (spacing >= 2^-52), and for n = 0 is the tie 1 - 2^-54 between
//    1 - 2^-53 (odd) and 1.0 (even) or above it: the sum rounds to >= n + 1.  x < n + 0.5: x <= n + 0.5 - u (both are
//    multiples of u when u <= 0.5), so x + c <= (n + 1 - u) - 2^-54 rounds to at most the double n + 1 - u.  (For
//    x < 0.5 that reads x <= 0.5 - 2^-54, whose sum 1 - 2^-53 is exact.)  u >= 1: x is an integer and c < u / 2.
//    [floor(fl(x + 0.5)) is wrong for the one double 0.5 - 2^-54, which this form gets right.]
//  * the convert truncates and saturates at 2^32 - 1; the min makes that 255 = the reference's v > 255 branch.
__device__ __forceinline__ uint32_t clampF_fast64(double x)
{
    const double t = x + 0.49999999999999994;
    uint32_t i;
    asm("v_cvt_u32_f64 %0, %1" : "=v"(i) : "v"(t));
    return min(i, 255u);
}

// fast-mode rounding of an fp32 accumulator into byte `sel` of `old`:
// floor(x+0.5) then saturating u8 convert + pack (v_cvt_pk_u8_f32).
__device__ __forceinline__ uint32_t pack_u8(float x, uint32_t sel, uint32_t old)
{
    return __builtin_amdgcn_cvt_pk_u8_f32(floorf(x + 0.5f), sel, old);
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

// color.YCbCr.RGBA() (Go image/color, the published integer form) then convertToNRGBA's uint8(c >> 8), convert.go:22-64:
// one NRGBA pixel of a decoded JPEG sample (convert.hip's kernel and ssim.hip's box sums straight from the planes)
__device__ __forceinline__ uint32_t ycc_u8(int v)
{
    // v >> 16 inside [0, 2^24), else 0 / 255: one v_med3_i32 and a shift
    const int c = v < 0 ? 0 : (v > 0xffffff ? 0xffffff : v);
    return static_cast<uint32_t>(c) >> 16;
}
__device__ __forceinline__ uint32_t ycc_nrgba_px(uint32_t yv, uint32_t cbv, uint32_t crv)
{
    const int yy1 = static_cast<int>(yv) * 0x10101;
    const int cb1 = static_cast<int>(cbv) - 128, cr1 = static_cast<int>(crv) - 128;
    const uint32_t r = ycc_u8(yy1 + 91881 * cr1);
    const uint32_t g = ycc_u8(yy1 - 22554 * cb1 - 46802 * cr1);
    const uint32_t b = ycc_u8(yy1 + 116130 * cb1);
    return r | (g << 8) | (b << 16) | 0xff000000u;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ uint32_t ld_px(const uint8_t *row, int x)
{
    return *(g_u32 *)(row + 4 * static_cast<size_t>(x));
}

// 16-byte streaming load: data that is read exactly once (box-downsample sources) is fetched
// with the non-temporal hint -- measured 7.0 TB/s against 6.2 TB/s for plain loads on MI355X
// (tools/membw.cpp), because the lines do not displace anything useful in L2 / Infinity Cache.
__device__ __forceinline__ u32x4 ld16_stream(const uint8_t *p)
{
    return __builtin_nontemporal_load((g_u32x4 *)p);
}

// BT.601 luminance exactly as the reference writes it (ssim.go:216, effects.go:96):
// (0.299*R + 0.587*G) + 0.114*B in fp64, no contraction (TU built with -ffp-contract=off).
__device__ __forceinline__ double u8_to_f64(uint32_t v);
__device__ __forceinline__ double lum601(uint32_t p)
{
    double r = u8_to_f64(p & 0xffu);
    double g = u8_to_f64((p >> 8) & 0xffu);
    double b = u8_to_f64((p >> 16) & 0xffu);
    return 0.299 * r + 0.587 * g + 0.114 * b;
}

// u8 (any v < 2^32) -> f64 without v_cvt_f64_u32: 2^52 + v is exact, so (2^52 | v) - 2^52 == v
__device__ __forceinline__ double u8_to_f64(uint32_t v)
{
    return __hiloint2double(0x43300000, static_cast<int>(v)) - 4503599627370496.0;
}

// sum over a 256-lane workgroup, result valid in thread 0 (s_red: 4 doubles of LDS)
__device__ __forceinline__ double block_sum_256(double v, double *s_red)
{
    // fixed-shape tree => bit-reproducible from run to run
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    double t = 0;
    if (threadIdx.x == 0) t = ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
    return t;
}

// boxDownsample's source range of output index d (ssim.go:262-275)
__host__ __device__ __forceinline__ void box_edge(int d, double ratio, int srcN, int &s0, int &s1)
{
    s0 = static_cast<int>(static_cast<double>(d) * ratio);
    s1 = static_cast<int>(static_cast<double>(d + 1) * ratio);
    if (s1 > srcN) s1 = srcN;
    if (s0 >= s1) s0 = s1 - 1;
    if (s0 < 0) s0 = 0;
}

__device__ __forceinline__ uint32_t box_finish(uint32_t r, uint32_t g, uint32_t b, uint32_t al, int count)
{
    // sums are exact integers; inv := 1.0/count; clampF(sum*inv)  (ssim.go:301-308)
    const double inv = 1.0 / static_cast<double>(count);
    return clampF_dev(u8_to_f64(r) * inv) | (clampF_dev(u8_to_f64(g) * inv) << 8) |
           (clampF_dev(u8_to_f64(b) * inv) << 16) | (clampF_dev(u8_to_f64(al) * inv) << 24);
}

// XCD-aware work-item remap: consecutive workgroup ids land on different XCDs
// (b % 8, MI355X_MICROARCH "Workgroup dispatch"), so hand each XCD a CONTIGUOUS range of
// tiles -- neighbouring tiles share halo rows/columns and then hit the same 4 MiB L2.
// grid must be launched with 8*ceil(total/8) blocks; returns -1 for the padding blocks.
__device__ __forceinline__ int xcd_tile(int bid, int total)
{
    int per = (total + 7) >> 3;
    int t = (bid & 7) * per + (bid >> 3);
    return ((bid >> 3) < per && t < total) ? t : -1;
}

}  // namespace fnx
