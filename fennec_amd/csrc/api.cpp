// fnx_* entry points: argument checks, host<->device staging, table upload, kernel launches.
#include <atomic>
#include <limits>
#include <cmath>

#include "common.hpp"

using namespace fnx;

namespace {

// Pix length of a w x h image with the given stride (Go: (h-1)*Stride + 4*w)
inline size_t pix_len(int w, int h, int stride)
{
    return (w > 0 && h > 0) ? static_cast<size_t>(h - 1) * stride + static_cast<size_t>(w) * 4 : 0;
}

int check_img(const void *p, int stride, int w, int h, const char *what)
{
    if (w <= 0 || h <= 0) return FNX_OK;
    if (!p) {
        set_error("invalid argument: %s pixel pointer is null", what);
        return FNX_ERR_INVALID;
    }
    if (stride < w * 4 || (stride & 3)) {
        set_error("invalid argument: %s stride %d for width %d", what, stride, w);
        return FNX_ERR_INVALID;
    }
    return FNX_OK;
}

int check_space(int space)
{
    if (space != FNX_HOST && space != FNX_DEVICE) {
        set_error("invalid argument: space must be FNX_HOST or FNX_DEVICE");
        return FNX_ERR_INVALID;
    }
    return FNX_OK;
}

// image -> image ops also take FNX_DEVICE_SRC: device-resident source, host destination
int check_space_io(int space)
{
    if (space != FNX_HOST && space != FNX_DEVICE && space != FNX_DEVICE_SRC) {
        set_error("invalid argument: space must be FNX_HOST, FNX_DEVICE or FNX_DEVICE_SRC");
        return FNX_ERR_INVALID;
    }
    return FNX_OK;
}

// SSIMFast's dims (ssim.go:52-56)
bool ssim_fast_dims(int w, int h, int *nw, int *nh)
{
    *nw = w;
    *nh = h;
    const int maxDim = 512;
    if (w > maxDim || h > maxDim) {
        double scale = double(maxDim) / std::fmax(double(w), double(h));
        *nw = int(std::fmax(8, std::round(double(w) * scale)));
        *nh = int(std::fmax(8, std::round(double(h) * scale)));
        return true;
    }
    return false;
}

// SSIMFast of n device image pairs (single pointers or device pointer arrays) -> d_out[n].
// Image i of the single-pointer form lives at a + i*a_img (used by MSSSIM with n == 1).
// defer (n == 1 only): windowed paths leave their final mean to launch_ssim_finish_deferred, which writes
// d_out_base[defer_index]; d_out is then d_out_base + defer_index
int ssim_fast_device(fnx_ctx *ctx, int n, const uint8_t *a, const uint8_t *const *as, int astride,
                     const uint8_t *b, const uint8_t *const *bs, int bstride, int w, int h,
                     const double *h_window, const double *d_window, double *d_out,
                     SsimDeferred *defer = nullptr, int defer_index = 0)
{
    int nw, nh;
    if (ssim_fast_dims(w, h, &nw, &nh)) {
        // boxDownsample both sides (ssim.go:57-58) into tight planes: [a0..an-1][b0..bn-1]
        const size_t plane = static_cast<size_t>(nw) * nh * 4;
        void *t = nullptr;
        FNX_TRY(scratch(ctx, SLOT_TMP2, plane * 2 * n + 16, &t));
        uint8_t *da = static_cast<uint8_t *>(t), *db = da + plane * n;
        FNX_TRY(launch_box_downsample_pair(ctx, n, a, as, astride, b, bs, bstride, w, h, da, nw * 4, plane, nw, nh));
        if (nw < 8 || nh < 8) {
            for (int i = 0; i < n; i++)
                FNX_TRY(launch_pixel_ssim(ctx, da + plane * i, db + plane * i, nw, nh, plane, d_out + i));
            return FNX_OK;
        }
        return launch_windowed_ssim(ctx, n, da, nw * 4, plane, db, nw * 4, plane, nw, nh, h_window, d_window,
                                    defer ? d_out - defer_index : d_out, defer, defer_index);
    }
    if (as || bs) {
        set_error("batched SSIMFast needs images larger than 512 px (the downsample path)");
        return FNX_ERR_INVALID;
    }
    if (w < 8 || h < 8) return launch_pixel_ssim(ctx, a, b, w, h, pix_len(w, h, astride), d_out);
    return launch_windowed_ssim(ctx, 1, a, astride, 0, b, bstride, 0, w, h, h_window, d_window,
                                defer ? d_out - defer_index : d_out, defer, defer_index);
}

// n doubles the result kernels write into: pinned host memory mapped into the device's address
// space, so that a blocking entry point only has to wait for the stream (result_wait) -- no D2H copy
// The slots start out as NaN (no SSIM value is one: the denominators are >= C1*C2 > 0): the host can then
// watch them fill instead of waiting for the runtime's completion signal (poll_results).
int result_slot(fnx_ctx *ctx, int n, double **d)
{
    void *p = nullptr;
    FNX_TRY(pinned_alloc(ctx, sizeof(double) * static_cast<size_t>(n > 16 ? n : 16), &p));
    *d = static_cast<double *>(p);
    for (int i = 0; i < n; i++) (*d)[i] = std::numeric_limits<double>::quiet_NaN();
    return FNX_OK;
}

// Result slots of an *_enqueue call: the batch's OWN pinned buffer (one per FIFO position), so that nothing
// a later call does to the pinned ring (growth frees it, wrap-around reuses it) can touch results that were
// enqueued and not fetched yet.  The position's buffer is free by construction: can_enqueue() has checked
// res_count < RES_DEPTH, and a fetched batch has been copied out.
int result_slot_queued(fnx_ctx *ctx, int n, double **d)
{
    fnx_ctx::ResBuf &rb = ctx->res_buf[(ctx->res_head + ctx->res_count) % fnx_ctx::RES_DEPTH];
    const size_t need = sizeof(double) * static_cast<size_t>(n > 16 ? n : 16);
    if (need > rb.cap) {
        if (rb.p) FNX_HIP(hipHostFree(rb.p));
        rb.p = nullptr;
        rb.cap = 0;
        FNX_HIP(hipHostMalloc(reinterpret_cast<void **>(&rb.p), need * 2, hipHostMallocDefault));
        rb.cap = need * 2;
    }
    *d = rb.p;
    for (int i = 0; i < n; i++) (*d)[i] = std::numeric_limits<double>::quiet_NaN();
    return FNX_OK;
}

// Wait until the n result slots hold values.  The result kernels are the last work of a call and write
// straight into (uncached) pinned host memory, so the values arrive a PCIe write after the kernel stores
// them, while hipStreamSynchronize / hipEventSynchronize return 10-20 us later; the stream (or the event)
// is still queried now and then, which also ends the wait if a value really is NaN or the GPU faulted.
template <typename Done>
int poll_results(const double *pinned, int n, Done done)
{
    const volatile double *v = pinned;
    for (unsigned spin = 1;; spin++) {
        bool all = true;
        for (int i = 0; i < n; i++) {
            const double x = v[i];
            if (x != x) { all = false; break; }
        }
        if (all) {
            std::atomic_thread_fence(std::memory_order_acquire);
            return FNX_OK;
        }
        if ((spin & 127u) == 0) {
            const hipError_t q = done();
            if (q == hipSuccess) return FNX_OK;
            if (q != hipErrorNotReady) FNX_HIP(q);
        }
        __builtin_ia32_pause();
    }
}

int result_wait(fnx_ctx *ctx, const double *pinned, double *out, int n)
{
    FNX_TRY(poll_results(pinned, n, [&] { return hipStreamQuery(ctx->stream); }));
    std::memcpy(out, pinned, sizeof(double) * size_t(n));
    return FNX_OK;
}

// ssim.go:344-352: exp(sum_i weights[i] * log(max(level_i, 1e-10)))
double msssim_combine(const double *lv, const double *weights, int nlev)
{
    double result = 0;
    for (int i = 0; i < nlev; i++) result += weights[i] * std::log(std::fmax(lv[i], 1e-10));   // ssim.go:351
    return std::exp(result);
}

// An event right behind the result kernels, and the batch joins the ctx's FIFO of unfetched results:
// fnx_results_fetch then never waits for work that was queued on this stream after the batch.
int publish_results(fnx_ctx *ctx, const double *pinned, int n)
{
    if (ctx->res_count == fnx_ctx::RES_DEPTH) {
        set_error("invalid argument: %d enqueued batches are waiting for fnx_results_fetch on this ctx", ctx->res_count);
        return FNX_ERR_INVALID;
    }
    fnx_ctx::Pending &q = ctx->res_q[(ctx->res_head + ctx->res_count) % fnx_ctx::RES_DEPTH];
    if (!q.ev) FNX_HIP(hipEventCreateWithFlags(&q.ev, hipEventDisableTiming));
    FNX_HIP(hipEventRecord(q.ev, ctx->stream));
    q.pinned = pinned;
    q.n = n;
    q.nraw = 0;
    q.nimg = 1;
    q.tail_parity = -1;
    ctx->res_count++;
    return FNX_OK;
}

int can_enqueue(fnx_ctx *ctx)
{
    if (ctx->res_count == fnx_ctx::RES_DEPTH) {
        set_error("invalid argument: %d enqueued batches are waiting for fnx_results_fetch on this ctx", ctx->res_count);
        return FNX_ERR_INVALID;
    }
    return FNX_OK;
}

}  // namespace

namespace fnx {

int lanczos_resize_tables(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int srcW, int srcH,
                          const TapTable &th, const TapTable &tv, uint8_t *dst, int dstride, int dstW, int dstH)
{
    FNX_ENTER(ctx);
    FNX_TRY(check_space_io(space));
    if (srcW <= 0 || srcH <= 0 || dstW <= 0 || dstH <= 0) return FNX_EMPTY;   // resize.go:41-43
    FNX_TRY(check_img(src, sstride, srcW, srcH, "src"));
    FNX_TRY(check_img(dst, dstride, dstW, dstH, "dst"));
    if (srcW == dstW && srcH == dstH) {   // flat copy of Pix (resize.go:45-49): no tables
        const size_t sl = pix_len(srcW, srcH, sstride), dl = pix_len(dstW, dstH, dstride);
        const size_t nbytes = sl < dl ? sl : dl;
        if (space == FNX_HOST) {
            std::memcpy(dst, src, nbytes);
        } else if (space == FNX_DEVICE_SRC) {
            FNX_HIP(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToHost, ctx->stream));
            FNX_HIP(hipStreamSynchronize(ctx->stream));
        } else {
            FNX_HIP(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, ctx->stream));
        }
        return FNX_OK;
    }
    FNX_REQUIRE(th.off && th.idx && th.wt && tv.off && tv.idx && tv.wt, "tap table is null");
    DevImg s;
    DevOut d;
    FNX_TRY(stage_in(ctx, space, src, sstride, srcW, srcH, SLOT_IN_A, &s));
    FNX_TRY(stage_out(ctx, space, dst, dstride, dstW, dstH, SLOT_OUT, &d));
    // both passes in one launch where the tables allow it (the uint8 intermediate stays in LDS)
    {
        const int rc = resize_fused(ctx, th, tv, s.p, s.stride, srcW, srcH, d.p, d.stride);
        if (rc < 0) return rc;
        if (rc != FNX_NOOP) return finish(ctx, space, &d);
    }
    // uint8 intermediate dstW x srcH (resize.go:51)
    const int tp = pitch16(dstW);
    void *tmp = nullptr;
    // ... followed by the H pass's verdict cells for the V pass (ResizeHint): at most one per 128 x 8 tmp pixels
    const size_t tmp_bytes = (static_cast<size_t>(tp) * srcH + 16 + 15) & ~size_t(15);
    ResizeHint hint;
    hint.cap = (static_cast<size_t>(dstW) / 128 + 2) * (static_cast<size_t>(srcH) / 8 + 2);
    FNX_TRY(scratch(ctx, SLOT_TMP0, tmp_bytes + sizeof(uint32_t) * hint.cap, &tmp));
    hint.cells = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(tmp) + tmp_bytes);
    FNX_TRY(resize_pass(ctx, false, th, s.p, s.stride, srcW, srcH, static_cast<uint8_t *>(tmp), tp, &hint));
    FNX_TRY(resize_pass(ctx, true, tv, static_cast<const uint8_t *>(tmp), tp, dstW, srcH, d.p, d.stride, &hint));
    return finish(ctx, space, &d);
}

// n same-geometry device images through ONE set of launches where resize_fused applies (every launch then holds n images'
// workgroups: a 4K call alone is 700 workgroups for 768-1024 slots -- one under-filled round), else image by image
int lanczos_resize_tables_batch(fnx_ctx *ctx, int n, const uint8_t *const *srcs, int sstride, int srcW, int srcH,
                                const TapTable &th, const TapTable &tv, uint8_t *const *dsts, int dstride, int dstW, int dstH)
{
    FNX_ENTER(ctx);
    FNX_REQUIRE(n >= 0 && (n == 0 || (srcs && dsts)), "batch arguments");
    FNX_REQUIRE(n <= FNX_BATCH_MAX, "more than FNX_BATCH_MAX (65535) images in one batch call: the image is a grid dimension");
    if (n == 0) return FNX_OK;
    if (srcW <= 0 || srcH <= 0 || dstW <= 0 || dstH <= 0) return FNX_EMPTY;   // resize.go:41-43
    FNX_REQUIRE((srcW == dstW && srcH == dstH) || (th.off && th.idx && th.wt && tv.off && tv.idx && tv.wt), "tap table is null");
    for (int i = 0; i < n; i++) {
        FNX_REQUIRE(srcs[i] && dsts[i], "null image in batch");
        FNX_TRY(check_img(srcs[i], sstride, srcW, srcH, "src"));
        FNX_TRY(check_img(dsts[i], dstride, dstW, dstH, "dst"));
    }
    if (n > 1 && !(srcW == dstW && srcH == dstH)) {
        const void *hosts[2] = {srcs, dsts};
        const size_t sizes[2] = {sizeof(void *) * size_t(n), sizeof(void *) * size_t(n)};
        void *dp[2];
        FNX_TRY(upload_tables(ctx, SLOT_PTRS, hosts, sizes, 2, dp));
        const int rc = resize_fused(ctx, th, tv, srcs[0], sstride, srcW, srcH, dsts[0], dstride, n,
                                    static_cast<const uint8_t *const *>(dp[0]), static_cast<uint8_t *const *>(dp[1]));
        if (rc < 0) return rc;
        if (rc != FNX_NOOP) return FNX_OK;
    }
    for (int i = 0; i < n; i++) {
        const int rc = lanczos_resize_tables(ctx, FNX_DEVICE, srcs[i], sstride, srcW, srcH, th, tv, dsts[i], dstride, dstW, dstH);
        if (rc < 0) return rc;
    }
    return FNX_OK;
}

}  // namespace fnx

extern "C" {

static int blur_batch_body(fnx_ctx *ctx, int n, const uint8_t *const *srcs, int sstride, int w, int h, const double *kernel, int radius,
                           int flags, uint8_t *const *dsts, int dstride);

int fnx_gaussian_blur(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h,
                      const double *kernel, int radius, int flags, uint8_t *dst, int dstride)
{
    FNX_ENTER(ctx);
    FNX_TRY(check_space_io(space));
    FNX_REQUIRE(kernel != nullptr && radius >= 0, "blur kernel");
    FNX_TRY(check_img(src, sstride, w, h, "src"));
    FNX_TRY(check_img(dst, dstride, w, h, "dst"));
    if (w <= 0 || h <= 0) return FNX_OK;
    FNX_REQUIRE(space != FNX_DEVICE || src != dst, "dst aliases src (the blur is not in-place)");
    ctx->kept.valid = false;
    if ((flags & FNX_BLUR_KEEP_BOX_SUMS) && space == FNX_DEVICE && !(sstride & 3) && !(dstride & 3))   // a batch of one (fnx_ssim_fast consumes it)
        return blur_batch_body(ctx, 1, &src, sstride, w, h, kernel, radius, flags, &dst, dstride);
    flags &= ~FNX_BLUR_KEEP_BOX_SUMS;
    DevImg s;
    DevOut d;
    FNX_TRY(stage_in(ctx, space, src, sstride, w, h, SLOT_IN_A, &s));
    FNX_TRY(stage_out(ctx, space, dst, dstride, w, h, SLOT_OUT, &d));
    FNX_TRY(launch_blur(ctx, 1, s.p, nullptr, s.stride, w, h, kernel, radius, flags, d.p, nullptr, d.stride));
    return finish(ctx, space, &d);
}

int fnx_gaussian_blur_batch(fnx_ctx *ctx, int n, const uint8_t *const *srcs, int sstride, int w,
                            int h, const double *kernel, int radius, int flags,
                            uint8_t *const *dsts, int dstride)
{
    FNX_ENTER(ctx);
    return blur_batch_body(ctx, n, srcs, sstride, w, h, kernel, radius, flags, dsts, dstride);
}

static int blur_batch_body(fnx_ctx *ctx, int n, const uint8_t *const *srcs, int sstride, int w, int h, const double *kernel, int radius,
                           int flags, uint8_t *const *dsts, int dstride)
{
    FNX_REQUIRE(n >= 0 && srcs && dsts && kernel && radius >= 0, "batch arguments");
    if (n == 0 || w <= 0 || h <= 0) return FNX_OK;
    FNX_REQUIRE(sstride >= 4 * w && dstride >= 4 * w && !(sstride & 3) && !(dstride & 3), "stride");
    for (int i = 0; i < n; i++) FNX_REQUIRE(srcs[i] && dsts[i] && srcs[i] != dsts[i], "null image in batch, or dst aliases src (the blur is not in-place)");
    const void *hosts[2] = {srcs, dsts};
    const size_t sizes[2] = {sizeof(void *) * size_t(n), sizeof(void *) * size_t(n)};
    void *dp[2];
    FNX_TRY(upload_tables(ctx, SLOT_PTRS, hosts, sizes, 2, dp));
    const bool keep = (flags & FNX_BLUR_KEEP_BOX_SUMS) != 0;
    flags &= ~FNX_BLUR_KEEP_BOX_SUMS;
    ctx->kept.valid = false;
    int nw, nh;
    if (keep && ssim_fast_dims(w, h, &nw, &nh) && nw >= 8 && nh >= 8) {
        // the one-pass kernel (fnx_gaussian_blur_ssim_fast_batch's first half): the same blurred bytes, and the box planes of both
        // sides into buffer set p.  The set stays this batch's until the scoring call -- or, if none comes, until the next
        // one-pass launch on it (ordered behind this one on `stream`).
        const int p = ctx->parity;
        const size_t plane = static_cast<size_t>(nw) * nh * 4;
        void *t = nullptr;
        FNX_TRY(scratch(ctx, p ? SLOT_PLANES1 : SLOT_PLANES0, plane * 2 * n + 16, &t));
        if (ctx->tail_pending[p]) FNX_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_tail[p], 0));
        ctx->boxes_on_main = true;               // blur, box planes and (next call) the windowed SSIM back to back on `stream`
        const int st = launch_blur_scored(ctx, n, static_cast<const uint8_t *const *>(dp[0]), sstride, w, h, kernel, radius, flags,
                                          static_cast<uint8_t *const *>(dp[1]), dstride, static_cast<uint8_t *>(t), plane, nw, nh);
        ctx->boxes_on_main = false;
        if (st < 0) return st;
        if (st == FNX_OK) {
            ctx->tail_pending[p] = false;        // whatever read set p before is in front of this blur on `stream`, and so is all that follows
            fnx_ctx::KeptBoxes &k = ctx->kept;
            k.srcs.assign(srcs, srcs + n);
            k.dsts.assign(dsts, dsts + n);
            k.n = n; k.sstride = sstride; k.dstride = dstride; k.w = w; k.h = h; k.nw = nw; k.nh = nh; k.parity = p;
            k.planes = static_cast<uint8_t *>(t); k.plane = plane;
            k.seq = ctx->op_seq;
            k.valid = true;
            return FNX_OK;
        }
        // FNX_NOOP: a shape the one-pass kernel is not built for -- the plain blur, nothing kept
    }
    return launch_blur(ctx, n, nullptr, static_cast<const uint8_t *const *>(dp[0]), sstride, w, h, kernel,
                       radius, flags, nullptr, static_cast<uint8_t *const *>(dp[1]), dstride);
}

// The effects read a SubImage (sstride != 4w) two ways: by rows, and as the flat front of its Pix slice
// (copy(dst.Pix, img.Pix), effects.go:68,120 -- see fx_flat_kernel): a host source of that kind goes up as the
// slice it is, stride kept, instead of being packed row by row.
static int stage_fx_src(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h, DevImg *s)
{
    return sstride != w * 4 ? stage_in_flat(ctx, space, src, sstride, w, h, SLOT_IN_A, s)
                            : stage_in(ctx, space, src, sstride, w, h, SLOT_IN_A, s);
}

int fnx_blur3x3(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h,
                uint8_t *dst, int dstride)
{
    FNX_ENTER(ctx);
    FNX_TRY(check_space_io(space));
    FNX_TRY(check_img(src, sstride, w, h, "src"));
    FNX_TRY(check_img(dst, dstride, w, h, "dst"));
    if (w <= 0 || h <= 0) return FNX_OK;
    DevImg s;
    DevOut d;
    FNX_TRY(stage_fx_src(ctx, space, src, sstride, w, h, &s));
    FNX_TRY(stage_out(ctx, space, dst, dstride, w, h, SLOT_OUT, &d));
    FNX_TRY(launch_blur3x3(ctx, s.p, s.stride, w, h, d.p, d.stride));
    return finish(ctx, space, &d);
}

static int sharpen_common(fnx_ctx *ctx, bool adaptive, int space, const uint8_t *src, int sstride,
                          int w, int h, double amount, uint8_t *dst, int dstride)
{
    FNX_ENTER(ctx);
    FNX_TRY(check_space_io(space));
    FNX_REQUIRE(w >= 3 && h >= 3, "sharpen needs w,h >= 3 (the reference returns the input below that)");
    FNX_TRY(check_img(src, sstride, w, h, "src"));
    FNX_TRY(check_img(dst, dstride, w, h, "dst"));
    DevImg s;
    DevOut d;
    FNX_TRY(stage_fx_src(ctx, space, src, sstride, w, h, &s));
    FNX_TRY(stage_out(ctx, space, dst, dstride, w, h, SLOT_OUT, &d));
    FNX_TRY(launch_sharpen(ctx, adaptive, s.p, s.stride, w, h, amount, d.p, d.stride));
    return finish(ctx, space, &d);
}

int fnx_sharpen(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h,
                double amount, uint8_t *dst, int dstride)
{
    return sharpen_common(ctx, false, space, src, sstride, w, h, amount, dst, dstride);
}

int fnx_adaptive_sharpen(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h,
                         double amount, uint8_t *dst, int dstride)
{
    return sharpen_common(ctx, true, space, src, sstride, w, h, amount, dst, dstride);
}

// n same-geometry device images in ONE launch of the streaming kernel (tight images); anything else image by image
static int sharpen_batch_common(fnx_ctx *ctx, bool adaptive, int n, const uint8_t *const *srcs, int sstride, int w, int h, double amount,
                                uint8_t *const *dsts, int dstride)
{
    FNX_ENTER(ctx);
    FNX_REQUIRE(n >= 0 && (n == 0 || (srcs && dsts)), "batch arguments");
    FNX_REQUIRE(n <= FNX_BATCH_MAX, "more than FNX_BATCH_MAX (65535) images in one batch call: the image is a grid dimension");
    if (n == 0) return FNX_OK;
    FNX_REQUIRE(w >= 3 && h >= 3, "sharpen needs w,h >= 3 (the reference returns the input below that)");
    for (int i = 0; i < n; i++) {
        FNX_REQUIRE(srcs[i] && dsts[i] && srcs[i] != dsts[i], "null image in batch, or dst aliases src");
        FNX_TRY(check_img(srcs[i], sstride, w, h, "src"));
        FNX_TRY(check_img(dsts[i], dstride, w, h, "dst"));
    }
    if (n > 1) {
        const void *hosts[2] = {srcs, dsts};
        const size_t sizes[2] = {sizeof(void *) * size_t(n), sizeof(void *) * size_t(n)};
        void *dp[2];
        FNX_TRY(upload_tables(ctx, SLOT_PTRS, hosts, sizes, 2, dp));
        const int rc = launch_sharpen_batch(ctx, adaptive, n, srcs[0], static_cast<const uint8_t *const *>(dp[0]), sstride, w, h, amount,
                                            dsts[0], static_cast<uint8_t *const *>(dp[1]), dstride);
        if (rc < 0) return rc;
        if (rc != FNX_NOOP) return FNX_OK;
    }
    for (int i = 0; i < n; i++) FNX_TRY(launch_sharpen(ctx, adaptive, srcs[i], sstride, w, h, amount, dsts[i], dstride));
    return FNX_OK;
}

int fnx_sharpen_batch(fnx_ctx *ctx, int n, const uint8_t *const *srcs, int sstride, int w, int h, double amount,
                      uint8_t *const *dsts, int dstride)
{
    return sharpen_batch_common(ctx, false, n, srcs, sstride, w, h, amount, dsts, dstride);
}

int fnx_adaptive_sharpen_batch(fnx_ctx *ctx, int n, const uint8_t *const *srcs, int sstride, int w, int h, double amount,
                               uint8_t *const *dsts, int dstride)
{
    return sharpen_batch_common(ctx, true, n, srcs, sstride, w, h, amount, dsts, dstride);
}

// ---- resize ------------------------------------------------------------------------
int fnx_resize_h(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int srcW, int srcH,
                 const int32_t *offset, const int32_t *index, const double *weight,
                 uint8_t *dst, int dstride, int dstW)
{
    FNX_ENTER(ctx);
    FNX_TRY(check_space_io(space));
    FNX_REQUIRE(srcW > 0 && srcH > 0 && dstW > 0, "dims");
    FNX_REQUIRE(offset && index && weight, "tap table is null");
    FNX_TRY(check_img(src, sstride, srcW, srcH, "src"));
    FNX_TRY(check_img(dst, dstride, dstW, srcH, "dst"));
    DevImg s;
    DevOut d;
    FNX_TRY(stage_in(ctx, space, src, sstride, srcW, srcH, SLOT_IN_A, &s));
    FNX_TRY(stage_out(ctx, space, dst, dstride, dstW, srcH, SLOT_OUT, &d));
    const TapTable t{offset, index, weight, dstW, 0};
    FNX_TRY(resize_pass(ctx, false, t, s.p, s.stride, srcW, srcH, d.p, d.stride));
    return finish(ctx, space, &d);
}

int fnx_resize_v(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int srcW, int srcH,
                 const int32_t *offset, const int32_t *index, const double *weight,
                 uint8_t *dst, int dstride, int dstH)
{
    FNX_ENTER(ctx);
    FNX_TRY(check_space_io(space));
    FNX_REQUIRE(srcW > 0 && srcH > 0 && dstH > 0, "dims");
    FNX_REQUIRE(offset && index && weight, "tap table is null");
    FNX_TRY(check_img(src, sstride, srcW, srcH, "src"));
    FNX_TRY(check_img(dst, dstride, srcW, dstH, "dst"));
    DevImg s;
    DevOut d;
    FNX_TRY(stage_in(ctx, space, src, sstride, srcW, srcH, SLOT_IN_A, &s));
    FNX_TRY(stage_out(ctx, space, dst, dstride, srcW, dstH, SLOT_OUT, &d));
    const TapTable t{offset, index, weight, dstH, 0};
    FNX_TRY(resize_pass(ctx, true, t, s.p, s.stride, srcW, srcH, d.p, d.stride));
    return finish(ctx, space, &d);
}

int fnx_lanczos_resize(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int srcW,
                       int srcH, const int32_t *offH, const int32_t *idxH, const double *wH,
                       const int32_t *offV, const int32_t *idxV, const double *wV,
                       uint8_t *dst, int dstride, int dstW, int dstH)
{
    const TapTable th{offH, idxH, wH, dstW, 0}, tv{offV, idxV, wV, dstH, 0};
    return fnx::lanczos_resize_tables(ctx, space, src, sstride, srcW, srcH, th, tv, dst, dstride, dstW, dstH);
}

int fnx_lanczos_resize_batch(fnx_ctx *ctx, int n, const uint8_t *const *srcs, int sstride, int srcW, int srcH,
                             const int32_t *offH, const int32_t *idxH, const double *wH,
                             const int32_t *offV, const int32_t *idxV, const double *wV,
                             uint8_t *const *dsts, int dstride, int dstW, int dstH)
{
    const TapTable th{offH, idxH, wH, dstW, 0}, tv{offV, idxV, wV, dstH, 0};
    return fnx::lanczos_resize_tables_batch(ctx, n, srcs, sstride, srcW, srcH, th, tv, dsts, dstride, dstW, dstH);
}

// ---- ssim.go -----------------------------------------------------------------------
int fnx_box_downsample(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int srcW,
                       int srcH, uint8_t *dst, int dstride, int dstW, int dstH)
{
    FNX_ENTER(ctx);
    FNX_TRY(check_space_io(space));
    if (srcW <= 0 || srcH <= 0 || dstW <= 0 || dstH <= 0) return FNX_EMPTY;   // ssim.go:246-248
    FNX_TRY(check_img(src, sstride, srcW, srcH, "src"));
    FNX_TRY(check_img(dst, dstride, dstW, dstH, "dst"));
    DevImg s;
    DevOut d;
    FNX_TRY(stage_in(ctx, space, src, sstride, srcW, srcH, SLOT_IN_A, &s));
    FNX_TRY(stage_out(ctx, space, dst, dstride, dstW, dstH, SLOT_OUT, &d));
    FNX_TRY(launch_box_downsample(ctx, 1, s.p, nullptr, s.stride, srcW, srcH, d.p, d.stride, 0, dstW, dstH));
    return finish(ctx, space, &d);
}

// the box planes a blur with FNX_BLUR_KEEP_BOX_SUMS left for exactly this scoring call?  (Either way they are gone afterwards.)
static bool kept_matches(fnx_ctx *ctx, int n, const uint8_t *const *as, int astride, const uint8_t *const *bs, int bstride, int w, int h)
{
    fnx_ctx::KeptBoxes &k = ctx->kept;
    bool use = k.valid && k.seq + 1 == ctx->op_seq && k.n == n && k.sstride == astride && k.dstride == bstride && k.w == w && k.h == h;
    for (int i = 0; use && i < n; i++) use = as[i] == k.srcs[i] && bs[i] == k.dsts[i];
    k.valid = false;
    return use;
}

// fnx_gaussian_blur_ssim_fast_batch's second half, on the main stream right behind the planes (the kept form's box_from_slabs_kernel
// ran there too): neither full-size image is read again
static int kept_score(fnx_ctx *ctx, int n, const double *window, const double *dwin, double *dres, bool batch_form)
{
    fnx_ctx::KeptBoxes &k = ctx->kept;
    const int p = k.parity;
    if (batch_form) ctx->partial_slot = p ? SLOT_PART1 : SLOT_PART0;    // (the one-pass entry's arithmetic; a single call keeps its own)
    const int rc = launch_windowed_ssim(ctx, n, k.planes, k.nw * 4, k.plane, k.planes + k.plane * n, k.nw * 4, k.plane, k.nw, k.nh, window, dwin, dres);
    ctx->partial_slot = -1;
    if (rc < 0) return rc;
    note_route(ctx, FNX_PROF_SSIM, "kept box planes + windowed SSIM");
    ctx->parity ^= 1;
    return FNX_OK;
}

int fnx_ssim_fast(fnx_ctx *ctx, int space, const uint8_t *a, int astride, const uint8_t *b,
                  int bstride, int w, int h, const double *window, double *out)
{
    FNX_ENTER(ctx);
    FNX_TRY(check_space(space));
    FNX_REQUIRE(window && out, "window/out is null");
    FNX_TRY(check_img(a, astride, w, h, "a"));
    FNX_TRY(check_img(b, bstride, w, h, "b"));
    if (w <= 0 || h <= 0) {   // pixelSSIM: n == 0 -> 1.0 (ssim.go:172-175)
        *out = 1.0;
        return FNX_OK;
    }
    void *dwin = nullptr;
    FNX_TRY(upload_table(ctx, SLOT_TABLE0, window, sizeof(double) * 64, &dwin));
    if (kept_matches(ctx, 1, &a, astride, &b, bstride, w, h) && space == FNX_DEVICE) {
        double *dres;
        FNX_TRY(result_slot(ctx, 1, &dres));
        FNX_TRY(kept_score(ctx, 1, window, static_cast<const double *>(dwin), dres, false));
        return result_wait(ctx, dres, out, 1);
    }
    DevImg da, db;
    int nw, nh;
    if (!ssim_fast_dims(w, h, &nw, &nh) && (w < 8 || h < 8)) {   // pixelSSIM on the inputs themselves
        FNX_REQUIRE(pix_len(w, h, bstride) >= pix_len(w, h, astride), "b.Pix shorter than a.Pix (the reference would panic)");
        FNX_TRY(stage_in_flat(ctx, space, a, astride, w, h, SLOT_IN_A, &da));
        FNX_TRY(stage_in_flat(ctx, space, b, bstride, w, h, SLOT_IN_B, &db));
    } else {
        FNX_TRY(stage_in(ctx, space, a, astride, w, h, SLOT_IN_A, &da));
        FNX_TRY(stage_in(ctx, space, b, bstride, w, h, SLOT_IN_B, &db));
    }
    double *dres;
    FNX_TRY(result_slot(ctx, 1, &dres));
    FNX_TRY(ssim_fast_device(ctx, 1, da.p, nullptr, da.stride, db.p, nullptr, db.stride, w, h, window,
                             static_cast<const double *>(dwin), dres));
    return result_wait(ctx, dres, out, 1);
}

int fnx_ssim_fast_batch(fnx_ctx *ctx, int n, const uint8_t *const *as, int astride,
                        const uint8_t *const *bs, int bstride, int w, int h,
                        const double *window, double *out)
{
    FNX_REQUIRE(out != nullptr, "out is null");
    FNX_REQUIRE(ctx && ctx->res_count == 0, "enqueued batches are waiting for fnx_results_fetch: fetch them before a blocking batch call");
    FNX_TRY(fnx_ssim_fast_batch_enqueue(ctx, n, as, astride, bs, bstride, w, h, window));
    return fnx_results_fetch(ctx, n, out);
}

int fnx_results_fetch(fnx_ctx *ctx, int n, double *out)
{
    FNX_ENTER(ctx);
    FNX_REQUIRE(n >= 0 && out, "fetch arguments");
    if (n == 0) return FNX_OK;
    FNX_REQUIRE(ctx->res_count > 0 && n <= ctx->res_q[ctx->res_head].n, "no enqueued results of that size on this ctx");
    fnx_ctx::Pending &q = ctx->res_q[ctx->res_head];            // oldest unfetched batch
    if (q.nraw > 0) {                                            // enqueued MSSSIMs: levels -> the weighted product, image by image
        for (int i = 0; i < n; i++) {
            FNX_TRY(poll_results(q.pinned + 5 * i, q.nraw, [&] { return hipEventQuery(q.ev); }));
            out[i] = msssim_combine(q.pinned + 5 * i, q.weights, q.nraw);
        }
    } else {
        FNX_TRY(poll_results(q.pinned, n, [&] { return hipEventQuery(q.ev); }));
        std::memcpy(out, q.pinned, sizeof(double) * size_t(n));
    }
    // a one-pass batch whose results have arrived has read its slabs / planes / partial sums for the last time: the
    // step that reuses the buffer set needs no stream-side wait for this tail (one barrier packet less per step)
    // -- only when ALL of the batch's results were polled: a partial fetch (n < q.n) has seen images 0..n-1 done while the
    // tail may still be reading the slabs and planes of the others (each image's last workgroup publishes its own mean)
    if (q.tail_parity >= 0 && q.tail_gen == ctx->tail_gen[q.tail_parity] && (q.nraw > 0 || n == q.n)) ctx->tail_pending[q.tail_parity] = false;
    ctx->res_head = (ctx->res_head + 1) % fnx_ctx::RES_DEPTH;
    ctx->res_count--;
    return FNX_OK;
}

int fnx_ssim_fast_batch_enqueue(fnx_ctx *ctx, int n, const uint8_t *const *as, int astride,
                                const uint8_t *const *bs, int bstride, int w, int h,
                                const double *window)
{
    FNX_ENTER(ctx);
    FNX_REQUIRE(n >= 0 && as && bs && window, "batch arguments");
    if (n == 0) return FNX_OK;
    FNX_REQUIRE(w > 0 && h > 0 && astride >= 4 * w && bstride >= 4 * w, "dims");
    FNX_TRY(can_enqueue(ctx));
    void *dwin = nullptr;
    FNX_TRY(upload_table(ctx, SLOT_TABLE0, window, sizeof(double) * 64, &dwin));
    double *dres;
    FNX_TRY(result_slot_queued(ctx, n, &dres));
    if (kept_matches(ctx, n, as, astride, bs, bstride, w, h)) {
        FNX_TRY(kept_score(ctx, n, window, static_cast<const double *>(dwin), dres, true));
        return publish_results(ctx, dres, n);
    }
    int nw, nh;
    bool al = !(astride & 15) && !(bstride & 15);
    for (int i = 0; i < n; i++) {
        FNX_REQUIRE(as[i] && bs[i], "null image in batch");
        al = al && !(reinterpret_cast<uintptr_t>(as[i]) & 15) && !(reinterpret_cast<uintptr_t>(bs[i]) & 15);
    }
    if (ssim_fast_dims(w, h, &nw, &nh) && al) {
        const void *hosts[2] = {as, bs};
        const size_t sizes[2] = {sizeof(void *) * size_t(n), sizeof(void *) * size_t(n)};
        void *dp[2];
        FNX_TRY(upload_tables(ctx, SLOT_PTRS, hosts, sizes, 2, dp));
        FNX_TRY(ssim_fast_device(ctx, n, nullptr, static_cast<const uint8_t *const *>(dp[0]), astride, nullptr,
                                 static_cast<const uint8_t *const *>(dp[1]), bstride, w, h, window,
                                 static_cast<const double *>(dwin), dres));
    } else {
        for (int i = 0; i < n; i++)
            FNX_TRY(ssim_fast_device(ctx, 1, as[i], nullptr, astride, bs[i], nullptr, bstride, w, h, window,
                                     static_cast<const double *>(dwin), dres + i));
    }
    return publish_results(ctx, dres, n);
}

int fnx_gaussian_blur_ssim_fast_batch_enqueue(fnx_ctx *ctx, int n, const uint8_t *const *srcs, int sstride,
                                              int w, int h, const double *kernel, int radius, int flags,
                                              uint8_t *const *dsts, int dstride, const double *window)
{
    FNX_ENTER(ctx);
    FNX_REQUIRE(n >= 0 && srcs && dsts && kernel && radius >= 0 && window, "batch arguments");
    if (n == 0) return FNX_OK;
    FNX_REQUIRE(w > 0 && h > 0, "dims");
    FNX_REQUIRE(sstride >= 4 * w && dstride >= 4 * w && !(sstride & 3) && !(dstride & 3), "stride");
    for (int i = 0; i < n; i++) FNX_REQUIRE(srcs[i] && dsts[i] && srcs[i] != dsts[i], "null image in batch, or dst aliases src (the blur is not in-place)");
    FNX_TRY(can_enqueue(ctx));
    int nw, nh;
    const bool down = ssim_fast_dims(w, h, &nw, &nh);
    if (down && nw >= 8 && nh >= 8) {
        // one pass: the blur kernel also accumulates both boxDownsample planes.  Buffer set p (slabs, planes,
        // partial sums) belongs to this step; the blur waits for the tail that used it two steps ago, and the
        // step's own tail -- box_from_slabs, windowed SSIM, finish -- runs on the second stream, i.e. under the
        // blur of whatever step the caller enqueues next.
        const int p = ctx->parity;
        const void *hosts[2] = {srcs, dsts};
        const size_t sizes[2] = {sizeof(void *) * size_t(n), sizeof(void *) * size_t(n)};
        void *dp[2];
        FNX_TRY(upload_tables(ctx, SLOT_PTRS, hosts, sizes, 2, dp));
        void *dwin = nullptr;
        FNX_TRY(upload_table(ctx, SLOT_TABLE0, window, sizeof(double) * 64, &dwin));   // before the blur: uploads ride on `stream`
        const size_t plane = static_cast<size_t>(nw) * nh * 4;
        void *t = nullptr;
        FNX_TRY(scratch(ctx, p ? SLOT_PLANES1 : SLOT_PLANES0, plane * 2 * n + 16, &t));
        uint8_t *planes = static_cast<uint8_t *>(t);
        if (ctx->tail_pending[p]) FNX_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_tail[p], 0));
        const int st = launch_blur_scored(ctx, n, static_cast<const uint8_t *const *>(dp[0]), sstride, w, h, kernel,
                                          radius, flags, static_cast<uint8_t *const *>(dp[1]), dstride, planes, plane, nw, nh);
        if (st < 0) return st;
        if (st == FNX_OK) {
            double *dres;
            FNX_TRY(result_slot_queued(ctx, n, &dres));
            // the windowed-SSIM launcher works on ctx->stream: point it at the tail stream for this call
            hipStream_t main_stream = ctx->stream;
            ctx->stream = ctx->stream2;
            ctx->partial_slot = p ? SLOT_PART1 : SLOT_PART0;
            int rc = launch_windowed_ssim(ctx, n, planes, nw * 4, plane, planes + plane * n, nw * 4, plane, nw, nh,
                                          window, static_cast<const double *>(dwin), dres);
            if (rc >= 0) {
                rc = publish_results(ctx, dres, n);      // the batch's event: behind the tail
                if (rc >= 0) {
                    fnx_ctx::Pending &q = ctx->res_q[(ctx->res_head + ctx->res_count - 1) % fnx_ctx::RES_DEPTH];
                    q.tail_parity = p;
                    q.tail_gen = ++ctx->tail_gen[p];
                }
            }
            if (rc >= 0 && hipEventRecord(ctx->ev_tail[p], ctx->stream2) != hipSuccess) {
                set_error("hipEventRecord (tail) failed");
                rc = FNX_ERR_HIP;
            }
            ctx->partial_slot = -1;
            ctx->stream = main_stream;
            if (rc < 0) return rc;
            ctx->tail_pending[p] = true;
            ctx->parity ^= 1;
            return FNX_OK;
        }
    }
    // shapes the one-pass kernel is not built for: the two ops back to back
    FNX_TRY(fnx_gaussian_blur_batch(ctx, n, srcs, sstride, w, h, kernel, radius, flags, dsts, dstride));
    return fnx_ssim_fast_batch_enqueue(ctx, n, srcs, sstride, const_cast<const uint8_t *const *>(dsts), dstride, w, h, window);
}

int fnx_gaussian_blur_ssim_fast_batch(fnx_ctx *ctx, int n, const uint8_t *const *srcs, int sstride, int w,
                                      int h, const double *kernel, int radius, int flags,
                                      uint8_t *const *dsts, int dstride, const double *window, double *out)
{
    FNX_REQUIRE(out != nullptr, "out is null");
    FNX_REQUIRE(ctx && ctx->res_count == 0, "enqueued batches are waiting for fnx_results_fetch: fetch them before a blocking batch call");
    FNX_TRY(fnx_gaussian_blur_ssim_fast_batch_enqueue(ctx, n, srcs, sstride, w, h, kernel, radius, flags, dsts,
                                                      dstride, window));
    return fnx_results_fetch(ctx, n, out);
}

// GaussianBlur + SSIMFast(src, blurred) of ONE image: what fnx_gaussian_blur followed by fnx_ssim_fast compute, with the image
// crossing PCIe once each way when it lives in host memory (the two calls upload the source twice and the blurred image once).
// The two kernels' launches back to back on the ctx's stream, on the staged copies: for one image the time is the link's (host
// space) or two launch latencies (device space), not HBM traffic -- the one-pass kernel and its second-stream tail are the
// BATCH entry's (a first version went through them: 82 us per device-space call against 53 for this form).
int fnx_gaussian_blur_ssim_fast(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h, const double *kernel,
                                int radius, int flags, uint8_t *dst, int dstride, const double *window, double *ssim)
{
    FNX_ENTER(ctx);
    FNX_TRY(check_space_io(space));
    FNX_REQUIRE(kernel != nullptr && radius >= 0 && window != nullptr && ssim != nullptr, "blur kernel / window / ssim");
    FNX_TRY(check_img(src, sstride, w, h, "src"));
    FNX_TRY(check_img(dst, dstride, w, h, "dst"));
    FNX_REQUIRE(w > 0 && h > 0, "dims");
    flags &= ~FNX_BLUR_KEEP_BOX_SUMS;
    ctx->kept.valid = false;
    int nw, nh;
    if (!ssim_fast_dims(w, h, &nw, &nh) && (w < 8 || h < 8)) {
        // pixelSSIM's sizes (ssim.go:61-63, the FLAT Pix slices): the two calls as they are
        FNX_TRY(fnx_gaussian_blur(ctx, space, src, sstride, w, h, kernel, radius, flags, dst, dstride));
        return fnx_ssim_fast(ctx, space, src, sstride, dst, dstride, w, h, window, ssim);
    }
    FNX_REQUIRE(space != FNX_DEVICE || src != dst, "dst aliases src (the blur is not in-place)");
    void *dwin = nullptr;
    FNX_TRY(upload_table(ctx, SLOT_TABLE0, window, sizeof(double) * 64, &dwin));
    DevImg s;
    DevOut d;
    FNX_TRY(stage_in(ctx, space, src, sstride, w, h, SLOT_IN_A, &s));
    FNX_TRY(stage_out(ctx, space, dst, dstride, w, h, SLOT_OUT, &d));
    FNX_TRY(launch_blur(ctx, 1, s.p, nullptr, s.stride, w, h, kernel, radius, flags, d.p, nullptr, d.stride));
    double *dres;
    FNX_TRY(result_slot(ctx, 1, &dres));
    FNX_TRY(ssim_fast_device(ctx, 1, s.p, nullptr, s.stride, d.p, nullptr, d.stride, w, h, window, static_cast<const double *>(dwin), dres));
    FNX_TRY(finish_enqueue(ctx, space, &d));                     // the blurred image starts back behind the score's kernels
    FNX_TRY(result_wait(ctx, dres, ssim, 1));
    if (space != FNX_DEVICE) FNX_HIP(hipStreamSynchronize(ctx->stream));
    return FNX_OK;
}

int fnx_ssim(fnx_ctx *ctx, int space, const uint8_t *a, int astride, const uint8_t *b,
             int bstride, int w, int h, const double *window, double *out)
{
    FNX_ENTER(ctx);
    FNX_TRY(check_space(space));
    FNX_REQUIRE(window && out, "window/out is null");
    FNX_TRY(check_img(a, astride, w, h, "a"));
    FNX_TRY(check_img(b, bstride, w, h, "b"));
    if (w <= 0 || h <= 0) {
        *out = 1.0;
        return FNX_OK;
    }
    void *dwin = nullptr;
    FNX_TRY(upload_table(ctx, SLOT_TABLE0, window, sizeof(double) * 64, &dwin));
    DevImg da, db;
    if (w < 8 || h < 8) {
        FNX_REQUIRE(pix_len(w, h, bstride) >= pix_len(w, h, astride), "b.Pix shorter than a.Pix (the reference would panic)");
        FNX_TRY(stage_in_flat(ctx, space, a, astride, w, h, SLOT_IN_A, &da));
        FNX_TRY(stage_in_flat(ctx, space, b, bstride, w, h, SLOT_IN_B, &db));
    } else {
        FNX_TRY(stage_in(ctx, space, a, astride, w, h, SLOT_IN_A, &da));
        FNX_TRY(stage_in(ctx, space, b, bstride, w, h, SLOT_IN_B, &db));
    }
    double *dres;
    FNX_TRY(result_slot(ctx, 1, &dres));
    if (w < 8 || h < 8) {   // ssim.go:35-37
        FNX_TRY(launch_pixel_ssim(ctx, da.p, db.p, w, h, pix_len(w, h, da.stride), dres));
    } else {                // toLuminance x2 + windowedSSIM at full resolution (ssim.go:39-42)
        FNX_TRY(launch_windowed_ssim(ctx, 1, da.p, da.stride, 0, db.p, db.stride, 0, w, h, window,
                                     static_cast<const double *>(dwin), dres));
    }
    return result_wait(ctx, dres, out, 1);
}

int fnx_ssim_enqueue(fnx_ctx *ctx, const uint8_t *a, int astride, const uint8_t *b, int bstride, int w, int h,
                     const double *window)
{
    FNX_ENTER(ctx);
    FNX_REQUIRE(window != nullptr && w > 0 && h > 0, "enqueue arguments");
    FNX_TRY(check_img(a, astride, w, h, "a"));
    FNX_TRY(check_img(b, bstride, w, h, "b"));
    FNX_TRY(can_enqueue(ctx));
    void *dwin = nullptr;
    FNX_TRY(upload_table(ctx, SLOT_TABLE0, window, sizeof(double) * 64, &dwin));
    double *dres;
    FNX_TRY(result_slot_queued(ctx, 1, &dres));
    if (w < 8 || h < 8) {   // ssim.go:35-37
        FNX_REQUIRE(pix_len(w, h, bstride) >= pix_len(w, h, astride), "b.Pix shorter than a.Pix (the reference would panic)");
        FNX_TRY(launch_pixel_ssim(ctx, a, b, w, h, pix_len(w, h, astride), dres));
        return publish_results(ctx, dres, 1);
    }
    // The score's kernels run on the ctx's second stream, behind everything enqueued so far (the images were
    // produced on `stream`): what the caller enqueues next -- the next image's AdaptiveSharpen in config 4 -- does
    // not wait for them, so one kernel's last workgroups and the next one's first share the chip instead of each
    // launch draining it (three launch boundaries per image, ~5 us each, at 8K).  The partial sums use the tail's
    // own slot: other calls on `stream` may use SLOT_PARTIAL meanwhile.  a and b stay the caller's until the fetch.
    static const bool same_stream = [] { const char *e = dev_env("FNX_SSIM_ENQUEUE_INLINE"); return e && e[0] == '1'; }();
    if (same_stream) {
        FNX_TRY(launch_windowed_ssim(ctx, 1, a, astride, 0, b, bstride, 0, w, h, window, static_cast<const double *>(dwin), dres));
        return publish_results(ctx, dres, 1);
    }
    hipEvent_t ev = ctx->ev_blur[ctx->ev_toggle];                 // the one-pass batches' hand-over events, same use
    FNX_HIP(hipEventRecord(ev, ctx->stream));
    FNX_HIP(hipStreamWaitEvent(ctx->stream2, ev, 0));
    ctx->stream2_used = true;
    hipStream_t main_stream = ctx->stream;
    ctx->stream = ctx->stream2;
    ctx->partial_slot = SLOT_PART0;
    int rc = launch_windowed_ssim(ctx, 1, a, astride, 0, b, bstride, 0, w, h, window, static_cast<const double *>(dwin), dres);
    if (rc >= 0) rc = publish_results(ctx, dres, 1);              // the result's event: behind the score, on the second stream
    ctx->partial_slot = -1;
    ctx->stream = main_stream;
    ctx->ev_toggle ^= 1;
    return rc;
}

// MSSSIM's weights, trimmed while a level's min dim < 8 (ssim.go:324-342)
static int msssim_weights(int w, int h, double (&weights)[5])
{
    const double full[5] = {0.0448, 0.2856, 0.3001, 0.2363, 0.1333};
    for (int i = 0; i < 5; i++) weights[i] = full[i];
    int nweights = 5;
    int tw = w, th = h;
    for (int i = 0; i < 4; i++) {
        int minDim = int(std::fmin(double(tw), double(th)));
        if (minDim < 8) {
            nweights = i + 1;
            double sum = 0;
            for (int j = 0; j < nweights; j++) sum += weights[j];
            for (int j = 0; j < nweights; j++) weights[j] /= sum;
            break;
        }
        tw /= 2;
        th /= 2;
    }
    return nweights;
}

// Every level's SSIMFast of a device-resident pair into dres[0 .. *nlev) (ssim.go:344-362); nothing waits.
static int msssim_levels_device(fnx_ctx *ctx, const uint8_t *ap, int astride, const uint8_t *bp, int bstride, int w, int h,
                                int nweights, const double *window, double *dres, int *nlev_out)
{
    void *dwin = nullptr;
    FNX_TRY(upload_table(ctx, SLOT_TABLE0, window, sizeof(double) * 64, &dwin));
    int nlev = 0;
    const int fused = launch_msssim_fused(ctx, ap, astride, bp, bstride, w, h, nweights, window, dres, &nlev);
    if (fused < 0) return fused;
    if (fused != FNX_OK) {
        // pyramid storage: levels 1.. of both images, ping-ponged in two slots per side
        const uint8_t *ca = ap, *cb = bp;
        int cas = astride, cbs = bstride, cw = w, ch = h;
        size_t lvl_bytes = static_cast<size_t>(w / 2) * (h / 2) * 4 + 16;
        void *pyr = nullptr;   // level k lives at (k&1)*2*lvl_bytes: [a][b]
        FNX_TRY(scratch(ctx, SLOT_TMP0, lvl_bytes * 4, &pyr));
        // the five levels' final means are taken by one launch at the end
        SsimDeferred defer;
        void *reserve = nullptr;
        FNX_TRY(scratch(ctx, SLOT_PARTIAL, sizeof(double) * SSIM_DEFER_DOUBLES, &reserve));
        for (int i = 0; i < nweights; i++) {
            FNX_TRY(ssim_fast_device(ctx, 1, ca, nullptr, cas, cb, nullptr, cbs, cw, ch, window,
                                     static_cast<const double *>(dwin), dres + i, &defer, i));
            nlev = i + 1;
            if (i < nweights - 1) {
                const int nw = cw / 2, nh = ch / 2;
                if (nw < 8 || nh < 8) break;              // ssim.go:354-358
                uint8_t *na = static_cast<uint8_t *>(pyr) + (i & 1) * 2 * lvl_bytes;
                uint8_t *nb = na + lvl_bytes;
                FNX_TRY(launch_box_downsample_pair(ctx, 1, ca, nullptr, cas, cb, nullptr, cbs, cw, ch, na, nw * 4,
                                                   lvl_bytes, nw, nh));
                ca = na; cb = nb; cas = cbs = nw * 4; cw = nw; ch = nh;
            }
        }
        FNX_TRY(launch_ssim_finish_deferred(ctx, defer, dres));
    }
    *nlev_out = nlev;
    return FNX_OK;
}

int fnx_pixel_ssim(fnx_ctx *ctx, int space, const uint8_t *a_pix, size_t a_pix_len, const uint8_t *b_pix,
                   size_t b_pix_len, int w, int h, double *out)
{
    FNX_ENTER(ctx);
    FNX_TRY(check_space(space));
    FNX_REQUIRE(out != nullptr && w >= 0 && h >= 0, "pixel_ssim arguments");
    if (static_cast<long long>(w) * h == 0) {      // ssim.go:172-175
        *out = 1.0;
        return FNX_OK;
    }
    const size_t walk = (a_pix_len + 3) & ~size_t(3);         // i < len(a.Pix), i += 4, reads [i, i+2]
    FNX_REQUIRE(a_pix_len == 0 || (a_pix && b_pix), "null Pix");
    FNX_REQUIRE(a_pix_len == 0 || (walk - 1 <= a_pix_len && walk - 1 <= b_pix_len),
                "a Pix slice ends inside the last pixel the loop reads, or b.Pix is shorter than a.Pix (the reference panics)");
    const uint8_t *da = a_pix, *db = b_pix;
    if (space == FNX_HOST && a_pix_len) {
        void *ta = nullptr, *tb = nullptr;
        FNX_TRY(scratch(ctx, SLOT_IN_A, walk + 16, &ta));
        FNX_TRY(scratch(ctx, SLOT_IN_B, walk + 16, &tb));
        FNX_HIP(hipMemsetAsync(static_cast<uint8_t *>(ta) + (walk - 4), 0, 4, ctx->stream));
        FNX_HIP(hipMemsetAsync(static_cast<uint8_t *>(tb) + (walk - 4), 0, 4, ctx->stream));
        FNX_HIP(hipMemcpyAsync(ta, a_pix, a_pix_len < walk ? a_pix_len : walk, hipMemcpyHostToDevice, ctx->stream));
        FNX_HIP(hipMemcpyAsync(tb, b_pix, b_pix_len < walk ? b_pix_len : walk, hipMemcpyHostToDevice, ctx->stream));
        da = static_cast<const uint8_t *>(ta);
        db = static_cast<const uint8_t *>(tb);
    }
    double *dres;
    FNX_TRY(result_slot(ctx, 1, &dres));
    FNX_TRY(launch_pixel_ssim(ctx, da, db, w, h, walk, dres));
    return result_wait(ctx, dres, out, 1);
}

int fnx_msssim(fnx_ctx *ctx, int space, const uint8_t *a, int astride, const uint8_t *b,
               int bstride, int w, int h, const double *window, double *out, double *per_level)
{
    FNX_ENTER(ctx);
    FNX_TRY(check_space(space));
    FNX_REQUIRE(window && out, "window/out is null");
    FNX_TRY(check_img(a, astride, w, h, "a"));
    FNX_TRY(check_img(b, bstride, w, h, "b"));
    double weights[5];
    const int nweights = msssim_weights(w, h, weights);
    double lv[5];
    for (double &v : lv) v = NAN;
    int nlev = 0;
    if (w <= 0 || h <= 0) {
        // SSIMFast of empty images: pixelSSIM n==0 -> 1.0; the halving loop then breaks (nw < 8)
        lv[0] = 1.0;
        nlev = 1;
    } else {
        DevImg da, db;
        // aCopy := toNRGBA(a), bCopy := toNRGBA(b) (ssim.go:345-346): flat copies -- the strides only vouch for the
        // slices' lengths (check_img above), the pyramid reads the first 4wh bytes of each as a tight image
        FNX_TRY(stage_in_front(ctx, space, a, w, h, SLOT_IN_A, &da));
        FNX_TRY(stage_in_front(ctx, space, b, w, h, SLOT_IN_B, &db));
        double *dres;
        FNX_TRY(result_slot(ctx, 5, &dres));
        FNX_TRY(msssim_levels_device(ctx, da.p, da.stride, db.p, db.stride, w, h, nweights, window, dres, &nlev));
        FNX_TRY(result_wait(ctx, dres, lv, nlev));
    }
    *out = msssim_combine(lv, weights, nlev);
    if (per_level)
        for (int i = 0; i < 5; i++) per_level[i] = i < nlev ? lv[i] : NAN;
    return FNX_OK;
}

int fnx_msssim_enqueue(fnx_ctx *ctx, const uint8_t *a, int astride, const uint8_t *b, int bstride, int w, int h,
                       const double *window)
{
    FNX_ENTER(ctx);
    FNX_REQUIRE(window != nullptr && w > 0 && h > 0, "enqueue arguments");
    FNX_TRY(check_img(a, astride, w, h, "a"));
    FNX_TRY(check_img(b, bstride, w, h, "b"));
    FNX_TRY(can_enqueue(ctx));
    double weights[5];
    const int nweights = msssim_weights(w, h, weights);
    double *dres;
    FNX_TRY(result_slot_queued(ctx, 5, &dres));
    int nlev = 0;
    FNX_TRY(msssim_levels_device(ctx, a, w * 4, b, w * 4, w, h, nweights, window, dres, &nlev));   // toNRGBA: flat (see fnx_msssim)
    fnx_ctx::Pending &q = ctx->res_q[(ctx->res_head + ctx->res_count) % fnx_ctx::RES_DEPTH];
    FNX_TRY(publish_results(ctx, dres, 1));
    q.nraw = nlev;
    for (int i = 0; i < 5; i++) q.weights[i] = weights[i];
    return FNX_OK;
}

// n same-geometry device pairs scored by ONE launch of the window kernel (the image is its second grid dimension), on the
// ctx's second stream like fnx_ssim_enqueue; one FIFO entry of n values
int fnx_ssim_batch_enqueue(fnx_ctx *ctx, int n, const uint8_t *const *as, int astride, const uint8_t *const *bs, int bstride,
                           int w, int h, const double *window)
{
    FNX_ENTER(ctx);
    FNX_REQUIRE(n >= 0 && (n == 0 || (as && bs)) && window != nullptr && w > 0 && h > 0, "batch arguments");
    FNX_REQUIRE(n <= FNX_BATCH_MAX, "more than FNX_BATCH_MAX (65535) images in one batch call: the image is a grid dimension");
    if (n == 0) return FNX_OK;
    for (int i = 0; i < n; i++) {
        FNX_REQUIRE(as[i] && bs[i], "null image in batch");
        FNX_TRY(check_img(as[i], astride, w, h, "a"));
        FNX_TRY(check_img(bs[i], bstride, w, h, "b"));
    }
    FNX_TRY(can_enqueue(ctx));
    void *dwin = nullptr;
    FNX_TRY(upload_table(ctx, SLOT_TABLE0, window, sizeof(double) * 64, &dwin));
    double *dres;
    FNX_TRY(result_slot_queued(ctx, n, &dres));
    if (w < 8 || h < 8) {   // ssim.go:35-37: pixelSSIM, pair by pair
        FNX_REQUIRE(pix_len(w, h, bstride) >= pix_len(w, h, astride), "b.Pix shorter than a.Pix (the reference would panic)");
        for (int i = 0; i < n; i++) FNX_TRY(launch_pixel_ssim(ctx, as[i], bs[i], w, h, pix_len(w, h, astride), dres + i));
        return publish_results(ctx, dres, n);
    }
    const void *hosts[2] = {as, bs};
    const size_t sizes[2] = {sizeof(void *) * size_t(n), sizeof(void *) * size_t(n)};
    void *dp[2];
    FNX_TRY(upload_tables(ctx, SLOT_PTRS, hosts, sizes, 2, dp));
    hipEvent_t ev = ctx->ev_blur[ctx->ev_toggle];                 // (see fnx_ssim_enqueue)
    FNX_HIP(hipEventRecord(ev, ctx->stream));
    FNX_HIP(hipStreamWaitEvent(ctx->stream2, ev, 0));
    ctx->stream2_used = true;
    hipStream_t main_stream = ctx->stream;
    ctx->stream = ctx->stream2;
    ctx->partial_slot = SLOT_PART0;
    int rc = launch_windowed_ssim(ctx, n, as[0], astride, 0, bs[0], bstride, 0, w, h, window, static_cast<const double *>(dwin), dres, nullptr, 0,
                                  static_cast<const uint8_t *const *>(dp[0]), static_cast<const uint8_t *const *>(dp[1]));
    if (rc >= 0) rc = publish_results(ctx, dres, n);
    ctx->partial_slot = -1;
    ctx->stream = main_stream;
    ctx->ev_toggle ^= 1;
    return rc;
}

int fnx_msssim_batch_enqueue(fnx_ctx *ctx, int n, const uint8_t *const *as, int astride, const uint8_t *const *bs, int bstride,
                             int w, int h, const double *window)
{
    FNX_ENTER(ctx);
    FNX_REQUIRE(n >= 0 && (n == 0 || (as && bs)) && window != nullptr && w > 0 && h > 0, "batch arguments");
    FNX_REQUIRE(n <= FNX_BATCH_MAX, "more than FNX_BATCH_MAX (65535) images in one batch call: the image is a grid dimension");
    if (n == 0) return FNX_OK;
    for (int i = 0; i < n; i++) {
        FNX_REQUIRE(as[i] && bs[i], "null image in batch");
        FNX_TRY(check_img(as[i], astride, w, h, "a"));
        FNX_TRY(check_img(bs[i], bstride, w, h, "b"));
    }
    FNX_TRY(can_enqueue(ctx));
    double weights[5];
    const int nweights = msssim_weights(w, h, weights);
    double *dres;
    FNX_TRY(result_slot_queued(ctx, 5 * n, &dres));
    int nlev = 0;
    bool batched = false;
    if (n > 1) {
        // the five launches of the fused form with the image as a grid dimension of each (ssim.hip); shapes it does not cover
        // (odd dims, a pair that is not 16-byte aligned, the non-default forms) take the loop below
        bool al = true;
        for (int i = 0; i < n; i++) al = al && !(reinterpret_cast<uintptr_t>(as[i]) & 15) && !(reinterpret_cast<uintptr_t>(bs[i]) & 15);
        if (al) {
            const void *hosts[2] = {as, bs};
            const size_t sizes[2] = {sizeof(void *) * size_t(n), sizeof(void *) * size_t(n)};
            void *dp[2];
            FNX_TRY(upload_tables(ctx, SLOT_PTRS, hosts, sizes, 2, dp));
            void *dwin = nullptr;
            FNX_TRY(upload_table(ctx, SLOT_TABLE0, window, sizeof(double) * 64, &dwin));
            const int rc = launch_msssim_fused(ctx, as[0], w * 4, bs[0], w * 4, w, h, nweights, window, dres, &nlev, n,
                                               static_cast<const uint8_t *const *>(dp[0]), static_cast<const uint8_t *const *>(dp[1]));
            if (rc < 0) return rc;
            batched = rc == FNX_OK;
        }
    }
    for (int i = 0; i < n && !batched; i++)      // (every image has the same levels: the dims decide)
        FNX_TRY(msssim_levels_device(ctx, as[i], w * 4, bs[i], w * 4, w, h, nweights, window, dres + 5 * i, &nlev));   // toNRGBA: flat (see fnx_msssim)
    fnx_ctx::Pending &q = ctx->res_q[(ctx->res_head + ctx->res_count) % fnx_ctx::RES_DEPTH];
    FNX_TRY(publish_results(ctx, dres, n));
    q.nraw = nlev;
    q.nimg = n;
    for (int i = 0; i < 5; i++) q.weights[i] = weights[i];
    return FNX_OK;
}

// ---- prepared reference ------------------------------------------------------------------
int fnx_ssim_fast_prepare(fnx_ctx *ctx, int space, const uint8_t *a, int astride, int w, int h,
                          fnx_prepared **out)
{
    FNX_ENTER(ctx);
    FNX_TRY(check_space(space));
    FNX_REQUIRE(out != nullptr && w > 0 && h > 0, "prepare arguments");
    FNX_TRY(check_img(a, astride, w, h, "a"));
    *out = nullptr;
    fnx_prepared *p = new fnx_prepared();
    p->w = w;
    p->h = h;
    const bool ds = ssim_fast_dims(w, h, &p->pw, &p->ph);
    void *d = nullptr;
    hipError_t e = hipMalloc(&d, static_cast<size_t>(p->pw) * p->ph * 4 + 16);
    if (e != hipSuccess) {
        delete p;
        set_error("hipMalloc failed: %s", hipGetErrorString(e));
        return FNX_ERR_OOM;
    }
    p->pix = static_cast<uint8_t *>(d);
    DevImg da;
    int rc = stage_in(ctx, space, a, astride, w, h, SLOT_IN_A, &da);
    if (rc >= 0) {
        if (ds) {
            rc = launch_box_downsample(ctx, 1, da.p, nullptr, da.stride, w, h, p->pix, p->pw * 4, 0, p->pw, p->ph);
        } else if (hipMemcpy2DAsync(p->pix, size_t(w) * 4, da.p, da.stride, size_t(w) * 4, h,
                                    hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
            set_error("hipMemcpy2DAsync failed");
            rc = FNX_ERR_HIP;
        }
    }
    if (rc >= 0 && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = FNX_ERR_HIP;
    if (rc < 0) {
        (void)hipFree(p->pix);
        delete p;
        return rc;
    }
    *out = p;
    return FNX_OK;
}

// SSIMFast(prepared reference, device-resident candidate)
// b_is_plane: b is already the candidate's pw x ph plane (launch_box_downsample_ycc made it from the JPEG planes)
static int against_device(fnx_ctx *ctx, const fnx_prepared *ref, const uint8_t *b, int bstride, const double *window,
                          double *out, bool b_is_plane = false)
{
    const int w = ref->w, h = ref->h, pw = ref->pw, ph = ref->ph;
    void *dwin = nullptr;
    FNX_TRY(upload_table(ctx, SLOT_TABLE0, window, sizeof(double) * 64, &dwin));
    double *dres;
    FNX_TRY(result_slot(ctx, 1, &dres));
    const uint8_t *cb = b;
    int cbs = bstride;
    if (!b_is_plane && (pw != w || ph != h)) {
        void *t = nullptr;
        FNX_TRY(scratch(ctx, SLOT_TMP2, static_cast<size_t>(pw) * ph * 4 + 16, &t));
        FNX_TRY(launch_box_downsample(ctx, 1, b, nullptr, bstride, w, h, static_cast<uint8_t *>(t), pw * 4, 0, pw, ph));
        cb = static_cast<const uint8_t *>(t);
        cbs = pw * 4;
    }
    if (pw < 8 || ph < 8) {
        // pixelSSIM walks both flat Pix slices; the prepared side is tight, so b must be too
        if (cbs != pw * 4) {
            void *t = nullptr;
            FNX_TRY(scratch(ctx, SLOT_TMP3, static_cast<size_t>(pw) * ph * 4 + 16, &t));
            FNX_HIP(hipMemcpy2DAsync(t, size_t(pw) * 4, cb, cbs, size_t(pw) * 4, ph, hipMemcpyDeviceToDevice, ctx->stream));
            cb = static_cast<const uint8_t *>(t);
        }
        FNX_TRY(launch_pixel_ssim(ctx, ref->pix, cb, pw, ph, static_cast<size_t>(pw) * ph * 4, dres));
    } else {
        FNX_TRY(launch_windowed_ssim(ctx, 1, ref->pix, pw * 4, 0, cb, cbs, 0, pw, ph, window,
                                     static_cast<const double *>(dwin), dres));
    }
    return result_wait(ctx, dres, out, 1);
}

int fnx_ssim_fast_against(fnx_ctx *ctx, const fnx_prepared *ref, int space, const uint8_t *b,
                          int bstride, const double *window, double *out)
{
    FNX_ENTER(ctx);
    FNX_TRY(check_space(space));
    FNX_REQUIRE(ref && window && out, "against arguments");
    FNX_TRY(check_img(b, bstride, ref->w, ref->h, "b"));
    DevImg db;
    FNX_TRY(stage_in(ctx, space, b, bstride, ref->w, ref->h, SLOT_IN_B, &db));
    return against_device(ctx, ref, db.p, db.stride, window, out);
}

// ---- decoded JPEG planes (image.YCbCr / image.Gray) -> NRGBA: convert.go:22-64 ----------------
static int chroma_dims(int ratio, int w, int h, int *cw, int *ch)
{
    switch (ratio) {   // image.NewYCbCr's plane sizes for Rect.Min == (0,0)
    case 0: *cw = w; *ch = h; break;
    case 1: *cw = (w + 1) / 2; *ch = h; break;
    case 2: *cw = (w + 1) / 2; *ch = (h + 1) / 2; break;
    case 3: *cw = w; *ch = (h + 1) / 2; break;
    case 4: *cw = (w + 3) / 4; *ch = h; break;
    case 5: *cw = (w + 3) / 4; *ch = (h + 1) / 2; break;
    default: set_error("invalid argument: subsample ratio"); return FNX_ERR_INVALID;
    }
    return FNX_OK;
}

// planes -> tight device NRGBA in `slot`
static int ycbcr_stage_convert(fnx_ctx *ctx, int space, const uint8_t *y, int ystride, const uint8_t *cb,
                               const uint8_t *cr, int cstride, int ratio, int w, int h, uint8_t *dst, int dstride)
{
    const bool gray = !cb && !cr;
    FNX_REQUIRE(y != nullptr && ystride >= w, "Y plane");
    FNX_REQUIRE(gray || (cb && cr), "Cb and Cr must both be given (or both NULL for image.Gray)");
    int cw = 0, ch = 0;
    if (!gray) {
        FNX_TRY(chroma_dims(ratio, w, h, &cw, &ch));
        FNX_REQUIRE(cstride >= cw, "chroma stride");
    }
    const uint8_t *dy = y, *dcb = cb, *dcr = cr;
    int dys = ystride, dcs = cstride;
    if (space == FNX_HOST) {
        dys = (w + 15) & ~15;
        dcs = (cw + 15) & ~15;
        void *t = nullptr;
        const size_t ybytes = static_cast<size_t>(dys) * h, cbytes = gray ? 0 : static_cast<size_t>(dcs) * ch;
        FNX_TRY(scratch(ctx, SLOT_TMP1, ybytes + 2 * cbytes + 16, &t));
        uint8_t *base = static_cast<uint8_t *>(t);
        FNX_HIP(hipMemcpy2DAsync(base, dys, y, ystride, w, h, hipMemcpyHostToDevice, ctx->stream));
        dy = base;
        if (!gray) {
            FNX_HIP(hipMemcpy2DAsync(base + ybytes, dcs, cb, cstride, cw, ch, hipMemcpyHostToDevice, ctx->stream));
            FNX_HIP(hipMemcpy2DAsync(base + ybytes + cbytes, dcs, cr, cstride, cw, ch, hipMemcpyHostToDevice, ctx->stream));
            dcb = base + ybytes;
            dcr = base + ybytes + cbytes;
        }
    }
    return launch_ycbcr_to_nrgba(ctx, dy, dys, gray ? nullptr : dcb, gray ? nullptr : dcr, dcs, gray ? 0 : ratio, w, h, dst,
                                 dstride);
}

int fnx_ycbcr_to_nrgba(fnx_ctx *ctx, int space, const uint8_t *y, int ystride, const uint8_t *cb,
                       const uint8_t *cr, int cstride, int ratio, int w, int h, uint8_t *dst, int dstride)
{
    FNX_ENTER(ctx);
    FNX_TRY(check_space(space));
    FNX_TRY(check_img(dst, dstride, w, h, "dst"));
    if (w <= 0 || h <= 0) return FNX_OK;
    DevOut d;
    FNX_TRY(stage_out(ctx, space, dst, dstride, w, h, SLOT_OUT, &d));
    FNX_TRY(ycbcr_stage_convert(ctx, space, y, ystride, cb, cr, cstride, ratio, w, h, d.p, d.stride));
    return finish(ctx, space, &d);
}

int fnx_ssim_fast_against_ycbcr(fnx_ctx *ctx, const fnx_prepared *ref, int space, const uint8_t *y, int ystride,
                                const uint8_t *cb, const uint8_t *cr, int cstride, int ratio,
                                const double *window, double *out)
{
    FNX_ENTER(ctx);
    FNX_TRY(check_space(space));
    FNX_REQUIRE(ref && window && out, "against arguments");
    const int w = ref->w, h = ref->h;
    FNX_REQUIRE(w > 0 && h > 0, "empty reference");
    void *t = nullptr;
    if (space == FNX_DEVICE && cb && cr && (ref->pw != w || ref->ph != h)) {
        // the candidate's plane straight from its planes: no NRGBA image (launch_box_downsample_ycc)
        FNX_REQUIRE(y != nullptr && ystride >= w, "Y plane");
        int cw = 0, ch = 0;
        FNX_TRY(chroma_dims(ratio, w, h, &cw, &ch));
        FNX_REQUIRE(cstride >= cw, "chroma stride");
        bool fused = false;
        FNX_TRY(scratch(ctx, SLOT_TMP2, static_cast<size_t>(ref->pw) * ref->ph * 4 + 16, &t));
        FNX_TRY(launch_box_downsample_ycc(ctx, y, ystride, cb, cr, cstride, ratio, w, h, static_cast<uint8_t *>(t), ref->pw * 4, ref->pw,
                                          ref->ph, &fused));
        if (fused) return against_device(ctx, ref, static_cast<const uint8_t *>(t), ref->pw * 4, window, out, true);
    }
    FNX_TRY(scratch(ctx, SLOT_IN_B, static_cast<size_t>(w) * h * 4 + 16, &t));
    FNX_TRY(ycbcr_stage_convert(ctx, space, y, ystride, cb, cr, cstride, ratio, w, h, static_cast<uint8_t *>(t), w * 4));
    return against_device(ctx, ref, static_cast<const uint8_t *>(t), w * 4, window, out);
}

// ---- the JPEG quantisation round trip (compress.go:45-74; SURVEY 8(f)2, first slice: jpeg.hip) -------------
namespace {

struct JpegPlanes {
    uint8_t *p[3];
    int ys, yh, cs, ch;
};

int jpeg_planes(fnx_ctx *ctx, Slot slot, int w, int h, JpegPlanes *jp)
{
    jpeg_plane_dims(w, h, &jp->ys, &jp->yh, &jp->cs, &jp->ch);
    const size_t yb = static_cast<size_t>(jp->ys) * jp->yh, cb = static_cast<size_t>(jp->cs) * jp->ch;
    void *t = nullptr;
    FNX_TRY(scratch(ctx, slot, yb + 2 * cb + 16, &t));
    jp->p[0] = static_cast<uint8_t *>(t);
    jp->p[1] = jp->p[0] + yb;
    jp->p[2] = jp->p[1] + cb;
    return FNX_OK;
}

// planes at `quality` from the unquantised planes in SLOT_JPEG0, then the NRGBA image toNRGBARef makes of them
int jpeg_decode_at(fnx_ctx *ctx, const JpegPlanes &orig, int w, int h, int quality, uint8_t *dst, int dstride)
{
    JpegPlanes work;
    FNX_TRY(jpeg_planes(ctx, SLOT_JPEG1, w, h, &work));
    const uint8_t *in[3] = {orig.p[0], orig.p[1], orig.p[2]};
    FNX_TRY(launch_jpeg_blocks(ctx, w, h, quality, in, work.p));
    return launch_ycbcr_to_nrgba(ctx, work.p[0], work.ys, work.p[1], work.p[2], work.cs, 2, w, h, dst, dstride);
}

}  // namespace

int fnx_jpeg_roundtrip(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h, int quality,
                       uint8_t *dst, int dstride)
{
    FNX_ENTER(ctx);
    FNX_TRY(check_space_io(space));
    FNX_TRY(check_img(src, sstride, w, h, "src"));
    FNX_TRY(check_img(dst, dstride, w, h, "dst"));
    if (w <= 0 || h <= 0) return FNX_OK;
    DevImg s;
    DevOut d;
    FNX_TRY(stage_in(ctx, space, src, sstride, w, h, SLOT_IN_A, &s));
    FNX_TRY(stage_out(ctx, space, dst, dstride, w, h, SLOT_OUT, &d));
    JpegPlanes orig;
    FNX_TRY(jpeg_planes(ctx, SLOT_JPEG0, w, h, &orig));
    FNX_TRY(launch_jpeg_ycc(ctx, s.p, s.stride, w, h, orig.p[0], orig.p[1], orig.p[2]));
    FNX_TRY(jpeg_decode_at(ctx, orig, w, h, quality, d.p, d.stride));
    return finish(ctx, space, &d);
}

// compressJPEGOptimal's search (compress.go:24-74) on a device-resident source whose unquantised planes are `orig`
// src_planes != nullptr (fnx_jpeg_recompress, r3): the source is a decoded JPEG still in its planes {Y, Cb, Cr, ystride, cstride,
// ratio} and s.p may be null -- the reference plane comes straight from them (launch_box_downsample_ycc); *fell_back = true
// says the planes' layout was not that kernel's case and nothing was done (the caller makes the image and calls again)
struct SrcPlanes {
    const uint8_t *y, *cb, *cr;
    int ys, cs, ratio;
};
static int jpeg_search_device(fnx_ctx *ctx, const DevImg &s, const JpegPlanes &orig, int w, int h, double target_ssim,
                              const double *window, int *quality, double *ssim, int *steps, bool *found_out,
                              const SrcPlanes *src_planes = nullptr, bool *fell_back = nullptr)
{
    // the source side of every SSIMFast of the search: prepared once (ssim.go:57 on the reference side)
    fnx_prepared ref;
    ref.w = w; ref.h = h;
    const bool ds = ssim_fast_dims(w, h, &ref.pw, &ref.ph);
    void *rp = nullptr;
    FNX_TRY(scratch(ctx, SLOT_JPEG3, static_cast<size_t>(ref.pw) * ref.ph * 4 + 16, &rp));
    ref.pix = static_cast<uint8_t *>(rp);
    if (src_planes) {
        bool done = false;
        if (ds) FNX_TRY(launch_box_downsample_ycc(ctx, src_planes->y, src_planes->ys, src_planes->cb, src_planes->cr, src_planes->cs,
                                                  src_planes->ratio, w, h, ref.pix, ref.pw * 4, ref.pw, ref.ph, &done));
        *fell_back = !done;
        if (!done) {
            ref.pix = nullptr;
            return FNX_OK;
        }
    } else if (ds) {
        FNX_TRY(launch_box_downsample(ctx, 1, s.p, nullptr, s.stride, w, h, ref.pix, ref.pw * 4, 0, ref.pw, ref.ph));
    } else {
        FNX_HIP(hipMemcpy2DAsync(ref.pix, size_t(w) * 4, s.p, s.stride, size_t(w) * 4, h, hipMemcpyDeviceToDevice, ctx->stream));
    }
    void *dec = nullptr;
    FNX_TRY(scratch(ctx, SLOT_JPEG2, static_cast<size_t>(w) * h * 4 + 16, &dec));
    // compress.go:24-74
    if (target_ssim >= 1.0) target_ssim = 0.999;
    int lo = 1, hi = 100, best_q = hi, n = 0;
    double best_ssim = 1.0;
    bool found = false;
    if (target_ssim >= 0.99) lo = 75;
    else if (target_ssim >= 0.97) lo = 50;
    else if (target_ssim >= 0.94) lo = 30;
    else if (target_ssim >= 0.90) lo = 15;
    void *cand = nullptr;                                         // the candidate's SSIMFast plane
    if (ds) FNX_TRY(scratch(ctx, SLOT_TMP2, static_cast<size_t>(ref.pw) * ref.ph * 4 + 16, &cand));
    while (lo <= hi) {
        const int mid = (lo + hi) / 2;
        double v = 0;
        // the candidate: planes at quality `mid`; its <= 256 px plane straight from them where the image would only be
        // written to be box-summed (r3), else toNRGBARef's image
        JpegPlanes work;
        FNX_TRY(jpeg_planes(ctx, SLOT_JPEG1, w, h, &work));
        const uint8_t *in[3] = {orig.p[0], orig.p[1], orig.p[2]};
        FNX_TRY(launch_jpeg_blocks(ctx, w, h, mid, in, work.p));
        bool fused = false;
        if (ds) FNX_TRY(launch_box_downsample_ycc(ctx, work.p[0], work.ys, work.p[1], work.p[2], work.cs, 2, w, h, static_cast<uint8_t *>(cand),
                                                  ref.pw * 4, ref.pw, ref.ph, &fused));
        if (fused) {
            FNX_TRY(against_device(ctx, &ref, static_cast<const uint8_t *>(cand), ref.pw * 4, window, &v, true));
        } else {
            FNX_TRY(launch_ycbcr_to_nrgba(ctx, work.p[0], work.ys, work.p[1], work.p[2], work.cs, 2, w, h, static_cast<uint8_t *>(dec), w * 4));
            FNX_TRY(against_device(ctx, &ref, static_cast<const uint8_t *>(dec), w * 4, window, &v));
        }
        n++;
        if (v >= target_ssim) {
            best_q = mid; best_ssim = v; found = true;
            hi = mid - 1;
        } else {
            lo = mid + 1;
        }
    }
    *quality = best_q;
    *ssim = best_ssim;
    if (steps) *steps = n;
    *found_out = found;
    ref.pix = nullptr;
    return FNX_OK;
}

int fnx_jpeg_quality_search(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h, double target_ssim,
                            const double *window, int *quality, double *ssim, int *steps)
{
    FNX_ENTER(ctx);
    FNX_TRY(check_space(space));
    FNX_REQUIRE(window && quality && ssim && w > 0 && h > 0, "search arguments");
    FNX_TRY(check_img(src, sstride, w, h, "src"));
    DevImg s;
    FNX_TRY(stage_in(ctx, space, src, sstride, w, h, SLOT_IN_A, &s));
    JpegPlanes orig;
    FNX_TRY(jpeg_planes(ctx, SLOT_JPEG0, w, h, &orig));
    FNX_TRY(launch_jpeg_ycc(ctx, s.p, s.stride, w, h, orig.p[0], orig.p[1], orig.p[2]));
    bool found = false;
    FNX_TRY(jpeg_search_device(ctx, s, orig, w, h, target_ssim, window, quality, ssim, steps, &found));
    return found ? FNX_OK : FNX_NOOP;      // FNX_NOOP: no quality reached the target (compress.go:82-86: encode at 100)
}

// the file of `orig`'s image at `quality` into host memory (see fnx_jpeg_encode)
static int jpeg_file_from_planes(fnx_ctx *ctx, const JpegPlanes &orig, int w, int h, int quality, uint8_t *out, size_t cap, size_t *nbytes)
{
    // two u64 the kernels write and the host reads: bits of the scan's string, 0xff bytes in it
    double *slot;
    FNX_TRY(result_slot(ctx, 2, &slot));
    unsigned long long *totals = reinterpret_cast<unsigned long long *>(slot);
    totals[0] = totals[1] = ~0ull;
    const uint8_t *planes[3] = {orig.p[0], orig.p[1], orig.p[2]};
    FNX_TRY(jpeg_entropy_code(ctx, w, h, quality, planes, totals));
    FNX_HIP(hipStreamSynchronize(ctx->stream));
    const unsigned long long tbits = totals[0];
    void *ecs = nullptr;
    FNX_TRY(scratch(ctx, SLOT_JPEG_ECS, jpeg_ecs_capacity(tbits), &ecs));
    FNX_TRY(jpeg_entropy_pack(ctx, w, h, tbits, static_cast<uint8_t *>(ecs), totals));
    FNX_HIP(hipStreamSynchronize(ctx->stream));
    const size_t ecs_bytes = static_cast<size_t>((tbits + 7) / 8) + static_cast<size_t>(tbits ? totals[1] : 0);
    std::vector<uint8_t> hdr;
    jpeg_header(w, h, quality, hdr);
    const size_t total = hdr.size() + ecs_bytes + 2;
    *nbytes = total;
    if (out == nullptr && cap == 0) return FNX_OK;       // size query: targetsize.go's searches need len(encoded) only
    if (out == nullptr || cap < total) {
        set_error("invalid argument: the file needs %zu bytes, the buffer holds %zu", total, cap);
        return FNX_ERR_INVALID;
    }
    std::memcpy(out, hdr.data(), hdr.size());
    if (ecs_bytes) FNX_HIP(hipMemcpy(out + hdr.size(), ecs, ecs_bytes, hipMemcpyDeviceToHost));
    out[total - 2] = 0xff;
    out[total - 1] = 0xd9;          // EOI
    return FNX_OK;
}

int fnx_jpeg_encode(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h, int quality, uint8_t *out, size_t cap,
                    size_t *nbytes)
{
    FNX_ENTER(ctx);
    FNX_TRY(check_space(space));
    FNX_REQUIRE(nbytes != nullptr && w > 0 && h > 0 && w <= 65535 && h <= 65535, "encode arguments (JPEG dims are 16-bit)");
    FNX_TRY(check_img(src, sstride, w, h, "src"));
    *nbytes = 0;
    DevImg s;
    FNX_TRY(stage_in(ctx, space, src, sstride, w, h, SLOT_IN_A, &s));
    JpegPlanes orig;
    FNX_TRY(jpeg_planes(ctx, SLOT_JPEG0, w, h, &orig));
    FNX_TRY(launch_jpeg_ycc(ctx, s.p, s.stride, w, h, orig.p[0], orig.p[1], orig.p[2]));
    return jpeg_file_from_planes(ctx, orig, w, h, quality, out, cap, nbytes);
}

int fnx_jpeg_size_search(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h, long long target_bytes, int skip_ssim,
                         const double *window, uint8_t *out, size_t cap, size_t *nbytes, int *quality, double *ssim, int *steps)
{
    FNX_ENTER(ctx);
    FNX_TRY(check_space(space));
    FNX_REQUIRE(nbytes && quality && ssim && (skip_ssim || window) && w > 0 && h > 0 && w <= 65535 && h <= 65535, "size search arguments");
    FNX_TRY(check_img(src, sstride, w, h, "src"));
    *nbytes = 0; *quality = 0; *ssim = 0.0;
    DevImg s;
    FNX_TRY(stage_in(ctx, space, src, sstride, w, h, SLOT_IN_A, &s));
    JpegPlanes orig;
    FNX_TRY(jpeg_planes(ctx, SLOT_JPEG0, w, h, &orig));
    FNX_TRY(launch_jpeg_ycc(ctx, s.p, s.stride, w, h, orig.p[0], orig.p[1], orig.p[2]));
    // targetsize.go:125-176
    const double pixels = static_cast<double>(static_cast<long long>(w) * h);
    const double bpp = static_cast<double>(target_bytes * 8) / pixels;
    int lo = 1, hi = 100;
    if (bpp < 0.5) hi = 40;
    else if (bpp < 1.0) { lo = 10; hi = 70; }
    else if (bpp < 2.0) { lo = 30; hi = 90; }
    else if (bpp > 4.0) lo = 60;
    int best_q = 0, n = 0;
    while (lo <= hi) {
        const int mid = (lo + hi) / 2;
        size_t sz = 0;
        FNX_TRY(jpeg_file_from_planes(ctx, orig, w, h, mid, nullptr, 0, &sz));       // len(encoded) only
        n++;
        if (static_cast<long long>(sz) <= target_bytes) {
            best_q = mid;
            lo = mid + 1;
        } else {
            hi = mid - 1;
        }
    }
    if (steps) *steps = n;
    if (best_q == 0) return FNX_NOOP;                        // bestBuf == nil: nothing fits (the caller tries its next strategy)
    *quality = best_q;
    if (!skip_ssim) {
        // the SSIMFast the reference takes of every fitting candidate; the one that survives is the best quality's
        fnx_prepared ref;
        ref.w = w; ref.h = h;
        const bool ds = ssim_fast_dims(w, h, &ref.pw, &ref.ph);
        void *rp = nullptr;
        FNX_TRY(scratch(ctx, SLOT_JPEG3, static_cast<size_t>(ref.pw) * ref.ph * 4 + 16, &rp));
        ref.pix = static_cast<uint8_t *>(rp);
        if (ds) FNX_TRY(launch_box_downsample(ctx, 1, s.p, nullptr, s.stride, w, h, ref.pix, ref.pw * 4, 0, ref.pw, ref.ph));
        else FNX_HIP(hipMemcpy2DAsync(ref.pix, size_t(w) * 4, s.p, s.stride, size_t(w) * 4, h, hipMemcpyDeviceToDevice, ctx->stream));
        void *dec = nullptr;
        FNX_TRY(scratch(ctx, SLOT_JPEG2, static_cast<size_t>(w) * h * 4 + 16, &dec));
        FNX_TRY(jpeg_decode_at(ctx, orig, w, h, best_q, static_cast<uint8_t *>(dec), w * 4));
        FNX_TRY(against_device(ctx, &ref, static_cast<const uint8_t *>(dec), w * 4, window, ssim));
        ref.pix = nullptr;
    }
    return jpeg_file_from_planes(ctx, orig, w, h, best_q, out, cap, nbytes);
}

int fnx_jpeg_compress(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h, double target_ssim, const double *window,
                      uint8_t *out, size_t cap, size_t *nbytes, int *quality, double *ssim, int *steps)
{
    FNX_ENTER(ctx);
    FNX_TRY(check_space(space));
    FNX_REQUIRE(window && nbytes && quality && ssim && w > 0 && h > 0 && w <= 65535 && h <= 65535, "compress arguments");
    FNX_TRY(check_img(src, sstride, w, h, "src"));
    *nbytes = 0;
    DevImg s;
    FNX_TRY(stage_in(ctx, space, src, sstride, w, h, SLOT_IN_A, &s));
    JpegPlanes orig;
    FNX_TRY(jpeg_planes(ctx, SLOT_JPEG0, w, h, &orig));
    FNX_TRY(launch_jpeg_ycc(ctx, s.p, s.stride, w, h, orig.p[0], orig.p[1], orig.p[2]));
    bool found = false;
    FNX_TRY(jpeg_search_device(ctx, s, orig, w, h, target_ssim, window, quality, ssim, steps, &found));
    // compress.go:76-86: the best candidate's bytes, or -- nothing reached the target -- an encode at bestQuality (100)
    return jpeg_file_from_planes(ctx, orig, w, h, *quality, out, cap, nbytes);
}

// ---- image.Decode of a baseline JPEG on the device (third slice: jpeg_dec.hip) ----------------
// toNRGBARef(jpeg.Decode(data)) into SLOT_JPEG_DEC_IMG (tight rows); *f describes the file
int fnx_jpeg_decode(fnx_ctx *ctx, const uint8_t *data, size_t n, int space, uint8_t *dst, int dstride, int *w, int *h)
{
    FNX_REQUIRE(data != nullptr && w != nullptr && h != nullptr, "decode arguments");
    JpegFile f;
    if (dst == nullptr) {                        // jpeg.DecodeConfig: the dimensions only (and whether the device handles the file);
        FNX_TRY(jpeg_parse(data, n, &f));        // host work, no ctx needed
        *w = f.w; *h = f.h;
        return FNX_OK;
    }
    FNX_ENTER(ctx);
    FNX_TRY(check_space_io(space));
    FNX_TRY(jpeg_parse(data, n, &f));
    *w = f.w; *h = f.h;
    FNX_TRY(check_img(dst, dstride, f.w, f.h, "dst"));
    DevOut d;
    FNX_TRY(stage_out(ctx, space, dst, dstride, f.w, f.h, SLOT_OUT, &d));
    uint8_t *pl[4] = {nullptr, nullptr, nullptr, nullptr};
    int ys = 0, cs = 0;
    FNX_TRY(jpeg_decode_planes(ctx, data, n, &f, pl, &ys, &cs));
    const bool grey = f.ncomp == 1;
    if (f.ncomp == 4) FNX_TRY(launch_cmyk_to_nrgba(ctx, pl, ys, f.adobe, f.w, f.h, d.p, d.stride));
    else FNX_TRY(launch_ycbcr_to_nrgba(ctx, pl[0], ys, grey ? nullptr : pl[1], grey ? nullptr : pl[2], cs, grey ? 0 : f.ratio, f.w, f.h, d.p, d.stride));
    return finish(ctx, space, &d);
}

// host only: jpeg_prog.cpp's output as the tests and the sanitizer runs read it
int fnx_jpeg_progressive_coefficients(const uint8_t *data, size_t n, int16_t *coef, size_t cap_blocks, size_t *blocks, int *w, int *h, int *ratio)
{
    FNX_REQUIRE(data != nullptr && blocks != nullptr && w != nullptr && h != nullptr && ratio != nullptr, "coefficient arguments");
    JpegFile f;
    FNX_TRY(jpeg_parse(data, n, &f));
    if (!f.progressive) return jpeg_unsupported("a baseline file here (its scan is decoded on the device)");
    const unsigned long long nblk = static_cast<unsigned long long>(f.mx) * f.my * f.nslots;
    *blocks = static_cast<size_t>(nblk);
    *w = f.w; *h = f.h; *ratio = f.ratio;
    if (nblk > static_cast<unsigned long long>(JPEG_HOST_MAX_BLOCKS)) return jpeg_unsupported("a host-decoded file of more than 4 M blocks (FNX_JPEG_HOST_MAX_BLOCKS)");
    if (coef == nullptr) return FNX_OK;
    FNX_REQUIRE(cap_blocks >= nblk, "coefficient capacity");
    if (8ull * n < nblk) return jpeg_corrupt("the file is too short for the image's blocks");       // (as fnx_jpeg_decode: before anything is sized by the header)
    std::memset(coef, 0, sizeof(int16_t) * 64 * static_cast<size_t>(nblk));
    return jpeg_progressive_coefficients(data, n, &f, coef);
}

int fnx_jpeg_recompress(fnx_ctx *ctx, const uint8_t *data, size_t n, double target_ssim, const double *window, uint8_t *out, size_t cap,
                        size_t *nbytes, int *quality, double *ssim, int *steps, int *w, int *h)
{
    FNX_ENTER(ctx);
    FNX_REQUIRE(data && window && nbytes && quality && ssim && w && h, "recompress arguments");
    *nbytes = 0;
    JpegFile f;
    uint8_t *pl[4] = {nullptr, nullptr, nullptr, nullptr};
    int ys = 0, cs = 0;
    FNX_TRY(jpeg_decode_planes(ctx, data, n, &f, pl, &ys, &cs));
    *w = f.w; *h = f.h;
    JpegPlanes orig;
    FNX_TRY(jpeg_planes(ctx, SLOT_JPEG0, f.w, f.h, &orig));
    bool found = false;
    DevImg s;
    s.p = nullptr; s.stride = f.w * 4;
    if (f.ncomp == 3) {
        // r3: toNRGBARef's image of the decoded planes (33 MB at 4K) was written for two readers only -- the encoder's colour
        // conversion and the reference plane's box sums; both take the planes themselves now (same per-pixel arithmetic)
        const SrcPlanes sp{pl[0], pl[1], pl[2], ys, cs, f.ratio};
        bool fell_back = false;
        FNX_TRY(launch_jpeg_ycc_planes(ctx, pl[0], ys, pl[1], pl[2], cs, f.ratio, f.w, f.h, orig.p[0], orig.p[1], orig.p[2]));
        FNX_TRY(jpeg_search_device(ctx, s, orig, f.w, f.h, target_ssim, window, quality, ssim, steps, &found, &sp, &fell_back));
        if (!fell_back) return jpeg_file_from_planes(ctx, orig, f.w, f.h, *quality, out, cap, nbytes);
    }
    void *t = nullptr;
    FNX_TRY(scratch(ctx, SLOT_JPEG_DEC_IMG, static_cast<size_t>(f.w) * f.h * 4 + 16, &t));
    uint8_t *img = static_cast<uint8_t *>(t);
    s.p = img;
    const bool grey = f.ncomp == 1;
    if (f.ncomp == 4) FNX_TRY(launch_cmyk_to_nrgba(ctx, pl, ys, f.adobe, f.w, f.h, img, s.stride));
    else FNX_TRY(launch_ycbcr_to_nrgba(ctx, pl[0], ys, grey ? nullptr : pl[1], grey ? nullptr : pl[2], cs, grey ? 0 : f.ratio, f.w, f.h, img, s.stride));
    FNX_TRY(launch_jpeg_ycc(ctx, s.p, s.stride, f.w, f.h, orig.p[0], orig.p[1], orig.p[2]));
    FNX_TRY(jpeg_search_device(ctx, s, orig, f.w, f.h, target_ssim, window, quality, ssim, steps, &found));
    return jpeg_file_from_planes(ctx, orig, f.w, f.h, *quality, out, cap, nbytes);
}

void fnx_prepared_free(fnx_ctx *ctx, fnx_prepared *p)
{
    if (!p) return;
    if (ctx && bind(ctx) == FNX_OK) {
        // hipFree waits for the device's outstanding work itself; a lent stream (fnx_ctx_use_stream) is its owner's to drain
        if (ctx->stream == ctx->own_stream) (void)hipStreamSynchronize(ctx->stream);
        if (p->pix) (void)hipFree(p->pix);
    }
    delete p;
}

// ---- Analyze (analyze.go:26-124) and the flat scans (convert.go:66-84) --------------------
// One launch per call (analyze.hip: analyze_one_kernel): the results land in pinned host memory, each image's ready word
// after them; this thread watches the words (FNX_ANALYZE_STAGED=1: the staged launches of rounds 1-4, A/B and tests).
static bool analyze_staged()
{
    static const bool v = [] { const char *e = dev_env("FNX_ANALYZE_STAGED"); return e && e[0] == '1'; }();
    return v;
}

static int analyze_direct(fnx_ctx *ctx, int n, const uint8_t *src, const uint8_t *const *srcs, int sstride, int w, int h, bool al,
                          fnx_analysis *out)
{
    void *pin = nullptr;
    const size_t rbytes = (sizeof(fnx_analysis) * static_cast<size_t>(n) + 63) & ~size_t(63);
    const int nr = n * launch_analyze_ready_words(), nv = launch_analyze_var_parts();
    const size_t vbytes = (sizeof(double) * static_cast<size_t>(n) * nv + 63) & ~size_t(63);
    FNX_TRY(pinned_alloc(ctx, rbytes + vbytes + sizeof(uint32_t) * static_cast<size_t>(nr), &pin));
    fnx_analysis *hres = static_cast<fnx_analysis *>(pin);
    double *hvar = reinterpret_cast<double *>(static_cast<char *>(pin) + rbytes);
    volatile uint32_t *ready = reinterpret_cast<volatile uint32_t *>(static_cast<char *>(pin) + rbytes + vbytes);
    for (int i = 0; i < nr; i++) ready[i] = 0u;
    std::atomic_thread_fence(std::memory_order_release);
    FNX_TRY(launch_analyze_one(ctx, n, src, srcs, sstride, w, h, al, hres, hvar, const_cast<uint32_t *>(ready)));
    for (unsigned spin = 1;; spin++) {
        bool all = true;
        for (int i = 0; i < nr; i++)
            if (ready[i] != 1u) { all = false; break; }
        if (all) break;
        if ((spin & 127u) == 0) {
            const hipError_t q = hipStreamQuery(ctx->stream);
            if (q == hipSuccess) break;                         // the stream is empty: the words are there
            if (q != hipErrorNotReady) FNX_HIP(q);
        }
        __builtin_ia32_pause();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    for (int i = 0; i < nr; i++) FNX_REQUIRE(ready[i] == 1u, "Analyze: the launch finished without a result");
    std::memcpy(out, hres, sizeof(fnx_analysis) * static_cast<size_t>(n));
    for (int i = 0; i < n; i++) {                                   // the contrast workgroups' sums, in order: bit-reproducible
        double v = 0.0;
        for (int k = 0; k < nv; k++) v += hvar[static_cast<size_t>(i) * nv + k];
        out[i].variance_sum = v;
    }
    return FNX_OK;
}

static void clamp_unique(fnx_analysis *a, int n)
{
    for (int i = 0; i < n; i++)
        if (a[i].unique_colors > 1024) a[i].unique_colors = 1024;   // len(colorSet) < 1024 gate, analyze.go:73
}

int fnx_analyze(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h, fnx_analysis *out)
{
    FNX_ENTER(ctx);
    FNX_TRY(check_space(space));
    FNX_REQUIRE(out != nullptr, "out is null");
    FNX_TRY(check_img(src, sstride, w, h, "src"));
    std::memset(out, 0, sizeof(*out));
    if (w <= 0 || h <= 0) return FNX_EMPTY;      // Analyze returns the zero ImageStats (analyze.go:37-39)
    DevImg s;
    FNX_TRY(stage_in(ctx, space, src, sstride, w, h, SLOT_IN_A, &s));
    const bool al = (reinterpret_cast<uintptr_t>(s.p) & 15u) == 0;
    if (!analyze_staged()) {
        FNX_TRY(analyze_direct(ctx, 1, s.p, nullptr, s.stride, w, h, al, out));
    } else {
        void *dres = nullptr;
        FNX_TRY(scratch(ctx, SLOT_RESULT, sizeof(fnx_analysis), &dres));
        FNX_TRY(launch_analyze(ctx, 1, s.p, nullptr, s.stride, w, h, al, static_cast<fnx_analysis *>(dres)));
        FNX_TRY(fetch_bytes(ctx, dres, out, sizeof(fnx_analysis)));
    }
    clamp_unique(out, 1);
    return FNX_OK;
}

int fnx_analyze_batch(fnx_ctx *ctx, int n, const uint8_t *const *srcs, int sstride, int w, int h,
                      fnx_analysis *out)
{
    FNX_ENTER(ctx);
    FNX_REQUIRE(n >= 0 && srcs && out, "batch arguments");
    if (n == 0) return FNX_OK;
    FNX_REQUIRE(w > 0 && h > 0 && sstride >= 4 * w && !(sstride & 3), "dims");
    bool al = true;
    for (int i = 0; i < n; i++) {
        FNX_REQUIRE(srcs[i], "null image in batch");
        al = al && !(reinterpret_cast<uintptr_t>(srcs[i]) & 15);
    }
    void *dp = nullptr;
    FNX_TRY(upload_table(ctx, SLOT_PTRS, srcs, sizeof(void *) * size_t(n), &dp));
    if (!analyze_staged() && n <= 4096) {
        FNX_TRY(analyze_direct(ctx, n, nullptr, static_cast<const uint8_t *const *>(dp), sstride, w, h, al, out));
    } else {
        void *dres = nullptr;
        FNX_TRY(scratch(ctx, SLOT_RESULT, sizeof(fnx_analysis) * size_t(n), &dres));
        FNX_TRY(launch_analyze(ctx, n, nullptr, static_cast<const uint8_t *const *>(dp), sstride, w, h, al,
                               static_cast<fnx_analysis *>(dres)));
        FNX_TRY(fetch_bytes(ctx, dres, out, sizeof(fnx_analysis) * size_t(n)));
    }
    clamp_unique(out, n);
    return FNX_OK;
}

int fnx_scan_flags(fnx_ctx *ctx, int space, const uint8_t *pix, size_t pix_len, int *is_opaque, int *is_grayscale)
{
    FNX_ENTER(ctx);
    FNX_TRY(check_space(space));
    FNX_REQUIRE(pix != nullptr || pix_len == 0, "pix is null");
    uint32_t flags = 0;
    if (pix_len >= 4) {
        const uint8_t *d = pix;
        if (space == FNX_HOST) {
            void *t = nullptr;
            FNX_TRY(scratch(ctx, SLOT_IN_A, pix_len + 16, &t));
            FNX_HIP(hipMemcpyAsync(t, pix, pix_len, hipMemcpyHostToDevice, ctx->stream));
            d = static_cast<const uint8_t *>(t);
        }
        // one launch; its last workgroup writes the flags into pinned host memory, which this thread watches (the
        // memset + kernel + copy + stream synchronisation this replaces took 39 us for a 6 us scan of a 4K image)
        void *pin = nullptr;
        const int cap = launch_scan_flags_slots(ctx);
        FNX_TRY(pinned_alloc(ctx, sizeof(uint32_t) * static_cast<size_t>(cap), &pin));
        volatile uint32_t *hf = static_cast<volatile uint32_t *>(pin);
        for (int i = 0; i < cap; i++) hf[i] = 0xffffffffu;
        std::atomic_thread_fence(std::memory_order_release);
        int nslots = 0;
        FNX_TRY(launch_scan_flags_direct(ctx, d, pix_len, static_cast<uint32_t *>(pin), &nslots));
        int first = 0;                                             // slots below it have reported
        for (unsigned spin = 1;; spin++) {
            while (first < nslots && hf[first] != 0xffffffffu) { flags |= hf[first]; first++; }
            if (first == nslots) break;
            if ((spin & 127u) == 0) {
                const hipError_t q = hipStreamQuery(ctx->stream);
                if (q != hipSuccess && q != hipErrorNotReady) FNX_HIP(q);
                if (q == hipSuccess) {                             // the stream is empty: every word is there
                    while (first < nslots && hf[first] != 0xffffffffu) { flags |= hf[first]; first++; }
                    FNX_REQUIRE(first == nslots, "scan_flags: the kernel finished without all of its results");
                    break;
                }
            }
            __builtin_ia32_pause();
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        flags &= 3u;
    }
    if (is_opaque) *is_opaque = (flags & 1u) ? 0 : 1;
    if (is_grayscale) *is_grayscale = (flags & 2u) ? 0 : 1;
    return FNX_OK;
}

// ---- applyPalette + palettedToNRGBA (targetsize.go:488-546) ---------------------------------
int fnx_apply_palette(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h,
                      const uint8_t *palette, int ncolors, uint8_t *indices, int istride,
                      uint8_t *quantized, int qstride)
{
    FNX_ENTER(ctx);
    FNX_TRY(check_space(space));
    FNX_REQUIRE(palette != nullptr && ncolors >= 1 && ncolors <= 256, "palette: 1..256 colours (image.Paletted indices are uint8)");
    for (int i = 0; i < ncolors; i++)
        FNX_REQUIRE(palette[4 * i + 3] == 255, "palette entries must be opaque (medianCut only emits A = 255, targetsize.go:407-410)");
    FNX_TRY(check_img(src, sstride, w, h, "src"));
    FNX_REQUIRE(indices != nullptr || quantized != nullptr, "no output requested");
    if (indices) FNX_REQUIRE(istride >= w, "index stride");
    if (quantized) FNX_TRY(check_img(quantized, qstride, w, h, "quantized"));
    if (w <= 0 || h <= 0) return FNX_OK;
    DevImg s;
    FNX_TRY(stage_in(ctx, space, src, sstride, w, h, SLOT_IN_A, &s));
    DevOut q;
    if (quantized) FNX_TRY(stage_out(ctx, space, quantized, qstride, w, h, SLOT_OUT, &q));
    uint8_t *didx = indices;
    int dpitch = istride;
    if (indices && space == FNX_HOST) {
        dpitch = (w + 3) & ~3;
        void *t = nullptr;
        FNX_TRY(scratch(ctx, SLOT_TMP0, static_cast<size_t>(dpitch) * h + 16, &t));
        didx = static_cast<uint8_t *>(t);
    }
    FNX_TRY(launch_apply_palette(ctx, s.p, s.stride, w, h, palette, ncolors, didx, dpitch,
                                 quantized ? q.p : nullptr, quantized ? q.stride : 0));
    if (indices && space == FNX_HOST)
        FNX_HIP(hipMemcpy2DAsync(indices, istride, didx, dpitch, w, h, hipMemcpyDeviceToHost, ctx->stream));
    if (quantized) return finish(ctx, space, &q);
    if (space == FNX_HOST) FNX_HIP(hipStreamSynchronize(ctx->stream));
    return FNX_OK;
}

// ---- orientation -------------------------------------------------------------------------
int fnx_orient(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h, int orient,
               uint8_t *dst, int dstride)
{
    FNX_ENTER(ctx);
    FNX_TRY(check_space_io(space));
    if (orient < 2 || orient > 8) return FNX_NOOP;   // exif.go:180-181,200-201
    const bool swap = orient >= 5;
    const int ow = swap ? h : w, oh = swap ? w : h;
    FNX_TRY(check_img(src, sstride, w, h, "src"));
    FNX_TRY(check_img(dst, dstride, ow, oh, "dst"));
    if (w <= 0 || h <= 0) return FNX_OK;
    DevImg s;
    DevOut d;
    FNX_TRY(stage_in(ctx, space, src, sstride, w, h, SLOT_IN_A, &s));
    FNX_TRY(stage_out(ctx, space, dst, dstride, ow, oh, SLOT_OUT, &d));
    FNX_TRY(launch_orient(ctx, s.p, s.stride, w, h, orient, d.p, d.stride));
    return finish(ctx, space, &d);
}

}  // extern "C"
