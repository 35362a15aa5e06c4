// EXIF orientation permutations (exif.go:178-203 over convert.go:186-256) on gfx950.
// Pure byte moves, bit-exact.  src(x,y) lands at dst(row,col):
//   2 FlipH      (y, w-1-x)        3 Rotate180 (h-1-y, w-1-x)     4 FlipV (h-1-y, x)
//   5 "Transpose"  = rot270 then flipH -> (w-1-x, h-1-y)
//   6 Rotate90CW   (x, h-1-y)      7 "Transverse" = rot90 then flipH -> (x, y)
//   8 Rotate270CW  (w-1-x, y)
// 2,3,4 keep w x h and are row-coalesced on both sides; 5..8 swap the axes and go through
// a padded 32x32 LDS tile so that both the reads and the writes are row-coalesced.
#include "common.hpp"
#include "devutil.hpp"

namespace fnx {

struct OrientArgs {
    const uint8_t *src;
    uint8_t *dst;
    int sstride, dstride, w, h;   // source dims
    int flip_r, flip_c;           // reverse the dst row / column index
};

// dst(y' , x') with y' = flip_r ? h-1-y : y, x' = flip_c ? w-1-x : x
__global__ __launch_bounds__(256) void orient_keep_kernel(OrientArgs a)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.w || y >= a.h) return;
    const int sx = a.flip_c ? a.w - 1 - x : x;
    const int sy = a.flip_r ? a.h - 1 - y : y;
    const uint32_t p = ld_px(a.src + static_cast<size_t>(sy) * a.sstride, sx);
    *reinterpret_cast<uint32_t *>(a.dst + static_cast<size_t>(y) * a.dstride + 4 * static_cast<size_t>(x)) = p;
}

// dst is h wide, w high: dst(r, c) = src(x, y) with r = flip_r ? w-1-x : x, c = flip_c ? h-1-y : y
__global__ __launch_bounds__(256) void orient_swap_kernel(OrientArgs a)
{
    __shared__ uint32_t tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;   // source tile origin
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5; // 32 x 8 threads
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int x = bx + lx, y = by + ly + j;
        if (x < a.w && y < a.h) tile[ly + j][lx] = ld_px(a.src + static_cast<size_t>(y) * a.sstride, x);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        // this thread writes source pixel (x = bx + ly + j, y = by + lx): lanes run along y,
        // which is the dst COLUMN axis -> coalesced row writes
        const int x = bx + ly + j, y = by + lx;
        if (x < a.w && y < a.h) {
            const int r = a.flip_r ? a.w - 1 - x : x;
            const int c = a.flip_c ? a.h - 1 - y : y;
            *reinterpret_cast<uint32_t *>(a.dst + static_cast<size_t>(r) * a.dstride + 4 * static_cast<size_t>(c)) =
                tile[lx][ly + j];
        }
    }
}

int launch_orient(fnx_ctx *ctx, const uint8_t *src, int sstride, int w, int h, int orient,
                  uint8_t *dst, int dstride)
{
    if (w <= 0 || h <= 0) return FNX_OK;
    OrientArgs a{src, dst, sstride, dstride, w, h, 0, 0};
    bool swap = false;
    switch (orient) {
    case 2: a.flip_c = 1; break;
    case 3: a.flip_r = 1; a.flip_c = 1; break;
    case 4: a.flip_r = 1; break;
    case 5: swap = true; a.flip_r = 1; a.flip_c = 1; break;
    case 6: swap = true; a.flip_c = 1; break;
    case 7: swap = true; break;
    case 8: swap = true; a.flip_r = 1; break;
    default: set_error("orientation %d is a no-op in the reference", orient); return FNX_ERR_INVALID;
    }
    if (swap) {
        dim3 grid((w + 31) / 32, (h + 31) / 32);
        hipLaunchKernelGGL(orient_swap_kernel, grid, dim3(256), 0, ctx->stream, a);
    } else {
        dim3 grid((w + 63) / 64, (h + 3) / 4);
        hipLaunchKernelGGL(orient_keep_kernel, grid, dim3(256), 0, ctx->stream, a);
    }
    FNX_HIP(hipGetLastError());
    return FNX_OK;
}

}  // namespace fnx
