// idct.go's 8-point passes (Chen-Wang, 11-bit constants) on values in registers: shared by the quantisation round trip
// (jpeg.hip) and the decoder (jpeg_dec.hip).  Restated from the published algorithm, as the rest of the JPEG path.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace fnx {

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

}  // namespace fnx
