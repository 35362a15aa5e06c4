// fennec_*: the reference's function set (ssim.go, resize.go, effects.go, exif.go, batch.go)
// mirrored in C++ above the fnx_* kernel layer -- guards, control flow and weight-table
// generation, i.e. everything the Go side of a cgo shim keeps.  No pixel arithmetic happens
// here: every image operation is a fnx_* call (HIP kernels); there is no CPU path.
#include <array>
#include <atomic>
#include <cmath>
#include <cstring>
#include <memory>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#include "common.hpp"

using namespace fnx;

namespace {

struct Taps {
    std::vector<int32_t> off, idx;
    std::vector<double> wt;
    uint64_t id = 0;      // process-unique, never reused: a ctx's resize plans are keyed by it
    TapTable table(int nout) const { return TapTable{off.data(), idx.data(), wt.data(), nout, id}; }
};

void build_taps(int dstSize, int srcSize, Taps &t)
{
    t.off.assign(static_cast<size_t>(dstSize) + 1, 0);
    int n = fennec_precomputeWeights(dstSize, srcSize, t.off.data(), nullptr, nullptr);
    t.idx.assign(static_cast<size_t>(n > 0 ? n : 1), 0);
    t.wt.assign(static_cast<size_t>(n > 0 ? n : 1), 0.0);
    fennec_precomputeWeights(dstSize, srcSize, t.off.data(), t.idx.data(), t.wt.data());
}

// precomputeWeights costs two sin() per tap (tens of thousands for a 4K axis): far more than the
// resize kernels themselves, and callers resize to the same few geometries over and over
// (SSIM's implicit resize, smartResize to a fixed box, the target-size scale search).  Keep the
// most recent tables; entries are immutable once built and shared by pointer.
std::shared_ptr<const Taps> make_taps(int dstSize, int srcSize)
{
    static std::mutex mu;
    static std::vector<std::pair<std::pair<int, int>, std::shared_ptr<const Taps>>> cache;
    const std::pair<int, int> key(dstSize, srcSize);
    {
        std::lock_guard<std::mutex> lk(mu);
        for (auto &e : cache)
            if (e.first == key) return e.second;
    }
    auto t = std::make_shared<Taps>();
    build_taps(dstSize, srcSize, *t);
    static std::atomic<uint64_t> next_id{1};
    t->id = next_id++;
    std::lock_guard<std::mutex> lk(mu);
    if (cache.size() >= 32) cache.erase(cache.begin());
    cache.emplace_back(key, t);
    return t;
}

// The 8x8 window never changes (windowSize 8, sigma 1.5: ssim.go:74-77); built once, thread-safely.
const double *ssim_window()
{
    static const std::array<double, 64> k = [] {
        std::array<double, 64> t{};
        fennec_gaussianKernel(8, 1.5, t.data());
        return t;
    }();
    return k.data();
}

// b resized to a's dims on the device when dims differ (ssim.go:31-33, 320-322).
// Returns device pointer/stride of the image to compare against.
int resize_b_to(fnx_ctx *ctx, int space, const uint8_t *b, int bstride, int bw, int bh, int w, int h,
                const uint8_t **out, int *ostride)
{
    const auto pth = make_taps(w, bw), ptv = make_taps(h, bh);
    const Taps &th = *pth, &tv = *ptv;
    void *d = nullptr;
    FNX_TRY(scratch(ctx, SLOT_TMP3, static_cast<size_t>(w) * h * 4 + 16, &d));
    const uint8_t *src = b;
    int sstride = bstride;
    if (space == FNX_HOST) {   // stage b ourselves; the resize then runs device -> device
        DevImg s;
        FNX_TRY(stage_in(ctx, FNX_HOST, b, bstride, bw, bh, SLOT_IN_B, &s));
        src = s.p;
        sstride = s.stride;
    }
    int rc = lanczos_resize_tables(ctx, FNX_DEVICE, src, sstride, bw, bh, th.table(w), tv.table(h),
                                   static_cast<uint8_t *>(d), w * 4, w, h);
    if (rc < 0) return rc;
    *out = static_cast<const uint8_t *>(d);
    *ostride = w * 4;
    return FNX_OK;
}

}  // namespace

extern "C" {

void fennec_gaussianKernel(int size, double sigma, double *kernel)
{   // ssim.go:223-241
    if (!kernel) return;
    const int half = size / 2;
    double sum = 0;
    int idx = 0;
    for (int y = -half; y < half; y++)
        for (int x = -half; x < half; x++) {
            double val = std::exp(-double(x * x + y * y) / (2 * sigma * sigma));
            kernel[idx++] = val;
            sum += val;
        }
    for (int i = 0; i < size * size; i++) kernel[i] /= sum;
}

int fennec_blurKernel(double sigma, double *kernel)
{   // effects.go:153-165
    const int radius = int(std::ceil(sigma * 3));
    if (!kernel) return radius;
    const int n = radius * 2 + 1;
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double x = double(i - radius);
        kernel[i] = std::exp(-(x * x) / (2 * sigma * sigma));
        sum += kernel[i];
    }
    for (int i = 0; i < n; i++) kernel[i] /= sum;
    return radius;
}

double fennec_lanczosKernel(double x)
{   // resize.go:57-69
    const double lanczosA = 3.0;
    if (x == 0) return 1.0;
    if (x < 0) x = -x;
    if (x >= lanczosA) return 0.0;
    double xpi = x * M_PI;
    return (lanczosA * std::sin(xpi) * std::sin(xpi / lanczosA)) / (xpi * xpi);
}

int fennec_precomputeWeights(int dstSize, int srcSize, int32_t *offset, int32_t *index, double *weight)
{   // resize.go:164-197, ratio/support as resize.go:81-87
    const double ratio = double(srcSize) / double(dstSize);
    double support = 3.0;
    if (ratio > 1) support = 3.0 * ratio;
    const double filterScale = std::fmax(ratio, 1.0);
    int total = 0;
    for (int d = 0; d < dstSize; d++) {
        double center = (double(d) + 0.5) * ratio - 0.5;
        int left = int(std::ceil(center - support));
        int right = int(std::floor(center + support));
        if (left < 0) left = 0;
        if (right >= srcSize) right = srcSize - 1;
        double wsum = 0;
        const int first = total;
        if (offset) offset[d] = total;
        for (int s = left; s <= right; s++) {
            double w = fennec_lanczosKernel((double(s) - center) / filterScale);
            if (w != 0) {
                wsum += w;
                if (index && weight) {
                    index[total] = s;
                    weight[total] = w;
                }
                total++;
            }
        }
        if (wsum != 0 && index && weight)
            for (int i = first; i < total; i++) weight[i] /= wsum;
    }
    if (offset) offset[dstSize] = total;
    return total;
}

int fennec_smartResizeDims(int srcW, int srcH, int maxW, int maxH, int *dstW, int *dstH)
{   // resize.go:12-32
    if (!dstW || !dstH) {
        fnx::set_error("invalid argument: fennec_smartResizeDims: dstW / dstH is null");
        return FNX_ERR_INVALID;
    }
    if (maxW <= 0) maxW = srcW;
    if (maxH <= 0) maxH = srcH;
    *dstW = srcW;
    *dstH = srcH;
    if (srcW <= maxW && srcH <= maxH) return 0;
    double ratio = std::fmin(double(maxW) / double(srcW), double(maxH) / double(srcH));
    *dstW = int(std::fmax(1, std::round(double(srcW) * ratio)));
    *dstH = int(std::fmax(1, std::round(double(srcH) * ratio)));
    return 1;
}

int fennec_ssimFastDims(int w, int h, int *newW, int *newH)
{   // ssim.go:52-56
    if (!newW || !newH) {
        fnx::set_error("invalid argument: fennec_ssimFastDims: newW / newH is null");
        return FNX_ERR_INVALID;
    }
    *newW = w;
    *newH = h;
    if (w > 512 || h > 512) {
        double scale = 512.0 / std::fmax(double(w), double(h));
        *newW = int(std::fmax(8, std::round(double(w) * scale)));
        *newH = int(std::fmax(8, std::round(double(h) * scale)));
        return 1;
    }
    return 0;
}

int fennec_SSIM(fnx_ctx *ctx, int space, const uint8_t *a, int astride, int aw, int ah,
                const uint8_t *b, int bstride, int bw, int bh, double *out)
{
    if (aw == bw && ah == bh) return fnx_ssim(ctx, space, a, astride, b, bstride, aw, ah, ssim_window(), out);
    FNX_ENTER(ctx);
    if (space != FNX_HOST && space != FNX_DEVICE) {
        set_error("invalid argument: space must be FNX_HOST or FNX_DEVICE");
        return FNX_ERR_INVALID;
    }
    if (aw <= 0 || ah <= 0 || bw <= 0 || bh <= 0) {
        // lanczosResize hands back a 0x0 image; pixelSSIM then indexes past it for a non-empty
        // `a` (the reference panics) or returns 1.0 for an empty `a`.
        if (aw <= 0 || ah <= 0) { *out = 1.0; return FNX_OK; }
        set_error("SSIM: second image is empty (the reference panics)");
        return FNX_ERR_INVALID;
    }
    const uint8_t *rb;
    int rbs;
    FNX_TRY(resize_b_to(ctx, space, b, bstride, bw, bh, aw, ah, &rb, &rbs));
    if (space == FNX_DEVICE) return fnx_ssim(ctx, FNX_DEVICE, a, astride, rb, rbs, aw, ah, ssim_window(), out);
    DevImg da;
    FNX_TRY(stage_in(ctx, FNX_HOST, a, astride, aw, ah, SLOT_IN_A, &da));
    return fnx_ssim(ctx, FNX_DEVICE, da.p, da.stride, rb, rbs, aw, ah, ssim_window(), out);
}

int fennec_SSIMFast(fnx_ctx *ctx, int space, const uint8_t *a, int astride, const uint8_t *b,
                    int bstride, int w, int h, double *out)
{
    return fnx_ssim_fast(ctx, space, a, astride, b, bstride, w, h, ssim_window(), out);
}

int fennec_MSSSIM(fnx_ctx *ctx, int space, const uint8_t *a, int astride, int aw, int ah,
                  const uint8_t *b, int bstride, int bw, int bh, double *out)
{
    if (aw == bw && ah == bh)
        return fnx_msssim(ctx, space, a, astride, b, bstride, aw, ah, ssim_window(), out, nullptr);
    FNX_ENTER(ctx);
    if (space != FNX_HOST && space != FNX_DEVICE) {
        set_error("invalid argument: space must be FNX_HOST or FNX_DEVICE");
        return FNX_ERR_INVALID;
    }
    if (aw <= 0 || ah <= 0 || bw <= 0 || bh <= 0) {
        if (aw <= 0 || ah <= 0) return fnx_msssim(ctx, space, a, astride, a, astride, aw, ah, ssim_window(), out, nullptr);
        set_error("MSSSIM: second image is empty (the reference panics)");
        return FNX_ERR_INVALID;
    }
    const uint8_t *rb;
    int rbs;
    FNX_TRY(resize_b_to(ctx, space, b, bstride, bw, bh, aw, ah, &rb, &rbs));
    if (space == FNX_DEVICE)
        return fnx_msssim(ctx, FNX_DEVICE, a, astride, rb, rbs, aw, ah, ssim_window(), out, nullptr);
    DevImg da;
    FNX_TRY(stage_in_front(ctx, FNX_HOST, a, aw, ah, SLOT_IN_A, &da));      // toNRGBA(a): the flat front of a.Pix (ssim.go:345)
    return fnx_msssim(ctx, FNX_DEVICE, da.p, da.stride, rb, rbs, aw, ah, ssim_window(), out, nullptr);
}

int fennec_MSSSIM_enqueue(fnx_ctx *ctx, const uint8_t *a, int astride, int aw, int ah, const uint8_t *b, int bstride,
                          int bw, int bh)
{
    FNX_ENTER(ctx);
    if (aw <= 0 || ah <= 0 || bw <= 0 || bh <= 0) {
        set_error("invalid argument: MSSSIM_enqueue takes non-empty device images");
        return FNX_ERR_INVALID;
    }
    if (aw == bw && ah == bh) return fnx_msssim_enqueue(ctx, a, astride, b, bstride, aw, ah, ssim_window());
    const uint8_t *rb;
    int rbs;
    FNX_TRY(resize_b_to(ctx, FNX_DEVICE, b, bstride, bw, bh, aw, ah, &rb, &rbs));   // ssim.go:320-322
    return fnx_msssim_enqueue(ctx, a, astride, rb, rbs, aw, ah, ssim_window());
}

int fennec_MSSSIM_batch_enqueue(fnx_ctx *ctx, int n, const uint8_t *const *as, int astride, int aw, int ah,
                                const uint8_t *const *bs, int bstride, int bw, int bh)
{
    FNX_ENTER(ctx);
    if (n < 0 || aw <= 0 || ah <= 0 || bw <= 0 || bh <= 0 || (n > 0 && (!as || !bs))) {
        set_error("invalid argument: MSSSIM_batch_enqueue takes non-empty device images");
        return FNX_ERR_INVALID;
    }
    if (n == 0) return FNX_OK;
    if (aw == bw && ah == bh) return fnx_msssim_batch_enqueue(ctx, n, as, astride, bs, bstride, aw, ah, ssim_window());
    // ssim.go:320-322 for the whole batch: every b resized to a's dims in ONE set of launches, n images side by side in the slot
    const size_t img = (static_cast<size_t>(aw) * ah * 4 + 255) & ~size_t(255);
    void *d = nullptr;
    FNX_TRY(scratch(ctx, SLOT_TMP3, img * n + 16, &d));
    std::vector<uint8_t *> ups(static_cast<size_t>(n));
    for (int i = 0; i < n; i++) ups[i] = static_cast<uint8_t *>(d) + img * i;
    const auto pth = make_taps(aw, bw), ptv = make_taps(ah, bh);
    FNX_TRY(lanczos_resize_tables_batch(ctx, n, bs, bstride, bw, bh, pth->table(aw), ptv->table(ah), ups.data(), aw * 4, aw, ah));
    return fnx_msssim_batch_enqueue(ctx, n, as, astride, const_cast<const uint8_t *const *>(ups.data()), aw * 4, aw, ah, ssim_window());
}

int fennec_GaussianBlur(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h,
                        double sigma, uint8_t *dst, int dstride)
{
    if (sigma <= 0) return FNX_NOOP;   // effects.go:147-149: same pointer
    const int radius = fennec_blurKernel(sigma, nullptr);
    std::vector<double> k(static_cast<size_t>(2 * radius + 1));
    fennec_blurKernel(sigma, k.data());
    // The drop-in is bit-exact where that is free: a host-space call is PCIe-bound (1.2 ms per 4K image,
    // the exact kernel 0.03 ms of it).  Device-resident callers get the fast kernel (<= 1 LSB on
    // <= 0.1 % of samples) and can ask fnx_gaussian_blur for FNX_BLUR_EXACT themselves.
    const int mode = space == FNX_DEVICE ? FNX_BLUR_FAST : FNX_BLUR_EXACT;
    return fnx_gaussian_blur(ctx, space, src, sstride, w, h, k.data(), radius, mode, dst, dstride);
}

int fennec_Sharpen(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h,
                   double strength, uint8_t *dst, int dstride)
{
    if (strength <= 0) return FNX_NOOP;   // effects.go:11-13
    if (strength > 1) strength = 1;
    if (w < 3 || h < 3) return FNX_NOOP;  // effects.go:20-22
    return fnx_sharpen(ctx, space, src, sstride, w, h, 1.0 + strength * 1.5, dst, dstride);
}

int fennec_AdaptiveSharpen(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h,
                           double strength, uint8_t *dst, int dstride)
{
    if (strength <= 0) return FNX_NOOP;   // effects.go:50-52
    if (strength > 1) strength = 1;
    if (w < 3 || h < 3) return FNX_NOOP;  // effects.go:59-61
    return fnx_adaptive_sharpen(ctx, space, src, sstride, w, h, 1.0 + strength * 2.0, dst, dstride);
}

int fennec_ApplyOrientation(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w,
                            int h, int orient, uint8_t *dst, int dstride)
{
    return fnx_orient(ctx, space, src, sstride, w, h, orient, dst, dstride);
}

int fennec_lanczosResize(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int srcW,
                         int srcH, uint8_t *dst, int dstride, int dstW, int dstH)
{
    if (srcW <= 0 || srcH <= 0 || dstW <= 0 || dstH <= 0) return FNX_EMPTY;
    if (srcW == dstW && srcH == dstH)
        return fnx_lanczos_resize(ctx, space, src, sstride, srcW, srcH, nullptr, nullptr, nullptr, nullptr,
                                  nullptr, nullptr, dst, dstride, dstW, dstH);
    const auto pth = make_taps(dstW, srcW), ptv = make_taps(dstH, srcH);
    const Taps &th = *pth, &tv = *ptv;
    return lanczos_resize_tables(ctx, space, src, sstride, srcW, srcH, th.table(dstW), tv.table(dstH), dst, dstride,
                                 dstW, dstH);
}

int fennec_lanczosResizeBatch(fnx_ctx *ctx, int n, const uint8_t *const *srcs, int sstride, int srcW, int srcH,
                              uint8_t *const *dsts, int dstride, int dstW, int dstH)
{
    if (srcW <= 0 || srcH <= 0 || dstW <= 0 || dstH <= 0) return FNX_EMPTY;
    if (srcW == dstW && srcH == dstH) {
        for (int i = 0; i < n; i++) {
            const int rc = fnx_lanczos_resize(ctx, FNX_DEVICE, srcs ? srcs[i] : nullptr, sstride, srcW, srcH, nullptr, nullptr, nullptr, nullptr,
                                              nullptr, nullptr, dsts ? dsts[i] : nullptr, dstride, dstW, dstH);
            if (rc < 0) return rc;
        }
        return FNX_OK;
    }
    const auto pth = make_taps(dstW, srcW), ptv = make_taps(dstH, srcH);
    return lanczos_resize_tables_batch(ctx, n, srcs, sstride, srcW, srcH, pth->table(dstW), ptv->table(dstH), dsts, dstride, dstW, dstH);
}

int fennec_boxDownsample(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int srcW,
                         int srcH, uint8_t *dst, int dstride, int dstW, int dstH)
{
    return fnx_box_downsample(ctx, space, src, sstride, srcW, srcH, dst, dstride, dstW, dstH);
}

double fennec_Summarize(int n, const int32_t *failed, const int32_t *has_result,
                        const int64_t *original_size, const int64_t *compressed_size,
                        const double *ssim, int64_t out4[4])
{   // batch.go:140-158
    int64_t succeeded = 0, nfailed = 0, saved = 0;
    double ssimSum = 0;
    if (!failed || !has_result || !original_size || !compressed_size || !ssim) n = 0;   // (a summary of nothing, not a crash)
    for (int i = 0; i < n; i++) {
        if (failed[i]) {
            nfailed++;
            continue;
        }
        succeeded++;
        if (has_result[i]) {
            saved += original_size[i] - compressed_size[i];
            ssimSum += ssim[i];
        }
    }
    if (out4) {
        out4[0] = n > 0 ? n : 0;
        out4[1] = succeeded;
        out4[2] = nfailed;
        out4[3] = saved;
    }
    return succeeded > 0 ? ssimSum / double(succeeded) : 0.0;
}

// ---- CompressBatch (batch.go:58-128), per-item work = compressJPEGOptimal on the device ----------------------------
// Workers' contexts outlive a batch: creating one (streams, events) and growing its scratch to 4K size costs ~10 ms, a
// 32-image batch ~25.  Idle contexts wait here, per device, until fennec_pool_release().
namespace {
std::mutex g_pool_mu;
std::vector<std::pair<int, fnx_ctx *>> g_pool;

fnx_ctx *pool_take(int device)
{
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (size_t i = 0; i < g_pool.size(); i++)
            if (g_pool[i].first == device) {
                fnx_ctx *c = g_pool[i].second;
                g_pool.erase(g_pool.begin() + static_cast<long>(i));
                return c;
            }
    }
    fnx_ctx *c = nullptr;
    return fnx_ctx_create(device, &c) == FNX_OK ? c : nullptr;
}

void pool_give(int device, fnx_ctx *c)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool.emplace_back(device, c);
}
}  // namespace

void fennec_pool_release(void)
{
    std::vector<std::pair<int, fnx_ctx *>> all;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        all.swap(g_pool);
    }
    for (auto &e : all) fnx_ctx_destroy(e.second);
}

// CompressFile for a JPEG source in standard mode (fennec.go:30-76 -> compressImageInternal :107-141 -> handleStandardMode
// :162-205) from the file's bytes, every pixel stage on the device: image.Decode + toNRGBA, ApplyOrientation (AutoOrient;
// the caller read the tag, exif.go), smartResize (MaxWidth / MaxHeight), analyzeFormat (Format: Auto), compressJPEGOptimal.
int fennec_CompressFileJPEG(fnx_ctx *ctx, const uint8_t *data, size_t n, const fennec_FileOptions *o, uint8_t *out, size_t cap,
                            size_t *nbytes, int *quality, double *ssim, int *steps, int dims[4])
{
    if (!ctx || !data || !o || !nbytes || !quality || !ssim || !dims) {
        set_error("invalid argument: CompressFileJPEG");
        return FNX_ERR_INVALID;
    }
    *nbytes = 0;
    int w = 0, h = 0;
    FNX_TRY(fnx_jpeg_decode(ctx, data, n, FNX_HOST, nullptr, 0, &w, &h));          // dimensions; refuses what the device decoder does not take
    if (!(o->orient > 1 && o->orient <= 8) && o->max_w <= 0 && o->max_h <= 0 && !o->auto_format) {
        // nothing between the decode and the search: the item body without the decoded image (fnx_jpeg_recompress, r3)
        dims[0] = dims[2] = w; dims[1] = dims[3] = h;
        return fnx_jpeg_recompress(ctx, data, n, o->target_ssim, ssim_window(), out, cap, nbytes, quality, ssim, steps, &w, &h);
    }
    void *b0 = nullptr;
    FNX_TRY(scratch(ctx, SLOT_FILE0, static_cast<size_t>(w) * h * 4 + 16, &b0));
    FNX_TRY(fnx_jpeg_decode(ctx, data, n, FNX_DEVICE, static_cast<uint8_t *>(b0), w * 4, &w, &h));
    const uint8_t *img = static_cast<const uint8_t *>(b0);
    bool in0 = true;
    if (o->orient > 1 && o->orient <= 8) {                                         // fennec.go:119-122 (OrientNormal == 1)
        const bool swap = o->orient >= 5;
        const int ow = swap ? h : w, oh = swap ? w : h;
        void *b1 = nullptr;
        FNX_TRY(scratch(ctx, SLOT_FILE1, static_cast<size_t>(ow) * oh * 4 + 16, &b1));
        FNX_TRY(fnx_orient(ctx, FNX_DEVICE, img, w * 4, w, h, o->orient, static_cast<uint8_t *>(b1), ow * 4));
        img = static_cast<const uint8_t *>(b1); w = ow; h = oh; in0 = false;
    }
    dims[0] = w; dims[1] = h;                                                      // OriginalDimensions
    if (o->max_w > 0 || o->max_h > 0) {                                            // fennec.go:127-129
        int nw = w, nh = h;
        if (fennec_smartResizeDims(w, h, o->max_w, o->max_h, &nw, &nh)) {
            void *b = nullptr;
            FNX_TRY(scratch(ctx, in0 ? SLOT_FILE1 : SLOT_FILE0, static_cast<size_t>(nw) * nh * 4 + 16, &b));
            FNX_TRY(fennec_lanczosResize(ctx, FNX_DEVICE, img, w * 4, w, h, static_cast<uint8_t *>(b), nw * 4, nw, nh));
            img = static_cast<const uint8_t *>(b); w = nw; h = nh; in0 = !in0;
        }
    }
    dims[2] = w; dims[3] = h;                                                      // FinalDimensions
    if (o->auto_format) {
        // analyzeFormat (convert.go:105-146) on what a JPEG decodes to (opaque): PNG when fewer than 256 distinct colours
        // among the sampled pixels.  The samples come to the host (<= 80 KB); the set is the reference's, early stop included.
        const long long total = static_cast<long long>(w) * h;
        const long long step = total > 10000 ? total / 10000 : 1;
        const int ns = static_cast<int>((total + step - 1) / step);
        void *ds = nullptr;
        FNX_TRY(scratch(ctx, SLOT_FILE_SAMPLES, sizeof(uint32_t) * static_cast<size_t>(ns) + 16, &ds));
        FNX_TRY(launch_sample_pixels(ctx, img, w * 4, w, h, step, static_cast<uint32_t *>(ds), ns));
        std::vector<uint32_t> smp(static_cast<size_t>(ns));
        FNX_TRY(fetch_bytes(ctx, ds, smp.data(), sizeof(uint32_t) * smp.size()));
        std::vector<uint32_t> seen;
        seen.reserve(512);
        bool alpha = false;
        for (int k = 0; k < ns && seen.size() < 512; k++) {
            if ((smp[static_cast<size_t>(k)] >> 24) < 255u) alpha = true;
            bool dup = false;
            for (uint32_t v : seen)
                if (v == smp[static_cast<size_t>(k)]) { dup = true; break; }
            if (!dup) seen.push_back(smp[static_cast<size_t>(k)]);
        }
        if (alpha || seen.size() < 256) return FNX_NOOP;                          // Format PNG: the caller's compressPNG
    }
    return fnx_jpeg_compress(ctx, FNX_DEVICE, img, w * 4, w, h, o->target_ssim, ssim_window(), out, cap, nbytes, quality, ssim, steps);
}

}  // extern "C"

// The pool of batch.go:58-128: `workers` threads over ONE closed queue of indices; item(ctx, idx, &result) does the work.
// A device LIST makes it the node's pool: workers = g x k goroutines' worth of threads, worker i bound to a
// context on devices[i mod g], still ONE queue -- a GPU whose items are cheaper simply takes more of them.
template <typename Item>
static int run_batch_pool(const int *devices, int ndev, int workers, int n, fennec_BatchResult *results, const volatile int *cancel,
                          fennec_on_item on_item, void *user, Item item)
{
    if (!devices || ndev <= 0) {
        set_error("invalid argument: CompressBatch needs at least one device");
        return FNX_ERR_INVALID;
    }
    if (workers <= 0) workers = static_cast<int>(std::thread::hardware_concurrency());      // batch.go:63-66
    if (workers <= 0) workers = 1;
    if (workers > n) workers = n;                                // batch.go:67-69
    for (int i = 0; i < n; i++) {
        results[i] = fennec_BatchResult{};
        results[i].index = i;
        results[i].failed = 1;                                   // until a worker says otherwise
        results[i].status = FNX_ERR_INVALID;
        results[i].device = -1;
    }
    std::atomic<int> next{0}, started{0};
    std::mutex done_mu;
    int completed = 0;
    const char *tr = std::getenv("FNX_POOL_TRACE");              // per-item wall times on stderr
    const bool trace = tr && tr[0] == '1';
    const auto t_batch = std::chrono::steady_clock::now();
    auto worker = [&](int wid) {
        const int device = devices[wid % ndev];
        fnx_ctx *ctx = pool_take(device);
        if (!ctx) return;
        started.fetch_add(1);
        for (;;) {
            const int idx = next.fetch_add(1);                   // the closed channel of indices (batch.go:72-81, 88)
            if (idx >= n) break;
            fennec_BatchResult &r = results[idx];
            if (cancel && *cancel) {                             // ctx.Done() before starting new work (batch.go:90-99)
                r.failed = 1; r.status = FNX_NOOP;
                continue;
            }
            const auto t_item = std::chrono::steady_clock::now();
            item(ctx, idx, r);
            if (trace)
                std::fprintf(stderr, "[fennec pool] item %d: %.3f ms (started %.3f ms into the batch)\n", idx,
                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_item).count(),
                             std::chrono::duration<double, std::milli>(t_item - t_batch).count());
            if (on_item) {                                       // batch.go:113-119
                std::lock_guard<std::mutex> lk(done_mu);
                on_item(++completed, n, user);
            }
        }
        pool_give(device, ctx);
    };
    std::vector<std::thread> pool;
    for (int w = 0; w < workers; w++) pool.emplace_back(worker, w);
    for (auto &t : pool) t.join();
    if (started.load() == 0) {
        set_error("CompressBatch: no worker could create a context on any of the %d listed devices (first: %d)", ndev, devices[0]);
        return FNX_ERR_HIP;
    }
    return FNX_OK;
}

template <typename Item>
static int run_batch_pool(int device, int workers, int n, fennec_BatchResult *results, const volatile int *cancel, fennec_on_item on_item,
                          void *user, Item item)
{
    return run_batch_pool(&device, 1, workers, n, results, cancel, on_item, user, item);
}

extern "C" {

int fennec_CompressBatchNRGBA(int device, int workers, int n, int space, const uint8_t *const *srcs, const int *strides,
                              const int *widths, const int *heights, const int64_t *original_sizes, double target_ssim,
                              uint8_t *const *outs, const size_t *caps, fennec_BatchResult *results, const volatile int *cancel,
                              fennec_on_item on_item, void *user)
{
    return fennec_CompressBatchNRGBADevices(&device, 1, workers, n, space, srcs, strides, widths, heights, original_sizes, target_ssim, outs,
                                            caps, results, cancel, on_item, user);
}

int fennec_CompressBatchNRGBADevices(const int *devices, int ndev, int workers, int n, int space, const uint8_t *const *srcs,
                                     const int *strides, const int *widths, const int *heights, const int64_t *original_sizes,
                                     double target_ssim, uint8_t *const *outs, const size_t *caps, fennec_BatchResult *results,
                                     const volatile int *cancel, fennec_on_item on_item, void *user)
{
    if (n <= 0) return FNX_OK;                                   // batch.go:59-61
    if (!srcs || !strides || !widths || !heights || !outs || !caps || !results) {
        set_error("invalid argument: CompressBatch arrays");
        return FNX_ERR_INVALID;
    }
    if (space != FNX_HOST && devices && ndev > 1) {
        for (int i = 1; i < ndev; i++)
            if (devices[i] != devices[0]) {
                set_error("invalid argument: device-resident items live on ONE device; a device list takes host-space items");
                return FNX_ERR_INVALID;
            }
    }
    return run_batch_pool(devices, ndev, workers, n, results, cancel, on_item, user, [&](fnx_ctx *ctx, int idx, fennec_BatchResult &r) {
        r.device = fnx_ctx_device(ctx);
        size_t nbytes = 0;
        int q = 0, steps = 0;
        double s = 0;
        const int rc = fnx_jpeg_compress(ctx, space, srcs[idx], strides[idx], widths[idx], heights[idx], target_ssim, ssim_window(),
                                         outs[idx], caps[idx], &nbytes, &q, &s, &steps);
        r.status = rc;
        r.failed = rc == FNX_OK ? 0 : 1;
        r.has_result = rc == FNX_OK ? 1 : 0;
        r.quality = q; r.steps = steps; r.ssim = s;
        r.original_size = original_sizes ? original_sizes[idx] : static_cast<int64_t>(widths[idx]) * heights[idx] * 4;
        r.compressed_size = static_cast<int64_t>(nbytes);
    });
}

int fennec_CompressBatchJPEG(int device, int workers, int n, const uint8_t *const *files, const size_t *sizes, double target_ssim,
                             uint8_t *const *outs, const size_t *caps, fennec_BatchResult *results, const volatile int *cancel,
                             fennec_on_item on_item, void *user)
{
    return fennec_CompressBatchJPEGDevices(&device, 1, workers, n, files, sizes, target_ssim, outs, caps, results, cancel, on_item, user);
}

int fennec_CompressBatchJPEGDevices(const int *devices, int ndev, int workers, int n, const uint8_t *const *files, const size_t *sizes,
                                    double target_ssim, uint8_t *const *outs, const size_t *caps, fennec_BatchResult *results,
                                    const volatile int *cancel, fennec_on_item on_item, void *user)
{
    if (n <= 0) return FNX_OK;                                   // batch.go:59-61
    if (!files || !sizes || !outs || !caps || !results) {
        set_error("invalid argument: CompressBatch arrays");
        return FNX_ERR_INVALID;
    }
    return run_batch_pool(devices, ndev, workers, n, results, cancel, on_item, user, [&](fnx_ctx *ctx, int idx, fennec_BatchResult &r) {
        r.device = fnx_ctx_device(ctx);
        size_t nbytes = 0;
        int q = 0, steps = 0, w = 0, h = 0;
        double s = 0;
        const int rc = fnx_jpeg_recompress(ctx, files[idx], sizes[idx], target_ssim, ssim_window(), outs[idx], caps[idx], &nbytes, &q, &s,
                                           &steps, &w, &h);
        r.status = rc;                                           // FNX_ERR_UNSUPPORTED: the caller decodes this one on the host
        r.failed = rc == FNX_OK ? 0 : 1;
        r.has_result = rc == FNX_OK ? 1 : 0;
        r.quality = q; r.steps = steps; r.ssim = s;
        r.original_size = static_cast<int64_t>(sizes[idx]);      // batch.go:103: the source FILE's size
        r.compressed_size = static_cast<int64_t>(nbytes);
    });
}

int fennec_CompressBatchJPEGOpts(int device, int workers, int n, const uint8_t *const *files, const size_t *sizes,
                                 const fennec_FileOptions *default_opts, const fennec_FileOptions *const *item_opts, uint8_t *const *outs,
                                 const size_t *caps, fennec_BatchResult *results, int *dims, const volatile int *cancel, fennec_on_item on_item,
                                 void *user)
{
    if (n <= 0) return FNX_OK;                                   // batch.go:59-61
    if (!files || !sizes || !default_opts || !outs || !caps || !results) {
        set_error("invalid argument: CompressBatch arrays");
        return FNX_ERR_INVALID;
    }
    return run_batch_pool(device, workers, n, results, cancel, on_item, user, [&](fnx_ctx *ctx, int idx, fennec_BatchResult &r) {
        r.device = fnx_ctx_device(ctx);
        const fennec_FileOptions &o = item_opts && item_opts[idx] ? *item_opts[idx] : *default_opts;      // batch.go:101-105
        size_t nbytes = 0;
        int q = 0, steps = 0, d4[4] = {0, 0, 0, 0};
        double s = 0;
        const int rc = fennec_CompressFileJPEG(ctx, files[idx], sizes[idx], &o, outs[idx], caps[idx], &nbytes, &q, &s, &steps, d4);
        if (dims) std::memcpy(dims + 4 * idx, d4, sizeof(d4));
        r.status = rc;                                           // FNX_NOOP: analyzeFormat chose PNG; FNX_ERR_UNSUPPORTED: host decode
        r.failed = rc == FNX_OK ? 0 : 1;
        r.has_result = rc == FNX_OK ? 1 : 0;
        r.quality = q; r.steps = steps; r.ssim = s;
        r.original_size = static_cast<int64_t>(sizes[idx]);
        r.compressed_size = static_cast<int64_t>(nbytes);
    });
}

double fennec_SummarizeResults(int n, const fennec_BatchResult *results, int64_t out4[4])
{   // batch.go:140-158
    int64_t succeeded = 0, nfailed = 0, saved = 0;
    double ssimSum = 0;
    if (!results) n = 0;
    for (int i = 0; i < n; i++) {
        if (results[i].failed) {
            nfailed++;
            continue;
        }
        succeeded++;
        if (results[i].has_result) {
            saved += results[i].original_size - results[i].compressed_size;
            ssimSum += results[i].ssim;
        }
    }
    if (out4) { out4[0] = n > 0 ? n : 0; out4[1] = succeeded; out4[2] = nfailed; out4[3] = saved; }
    return succeeded > 0 ? ssimSum / double(succeeded) : 0.0;
}

// ---- Analyze (analyze.go:26-230) -----------------------------------------------------------
void fennec_statsFromAnalysis(const fnx_analysis *a, int w, int h, fennec_ImageStats *st)
{
    std::memset(st, 0, sizeof(*st));
    st->Width = w;
    st->Height = h;
    if (w <= 0 || h <= 0 || !a) return;
    const double n = double(static_cast<long long>(w) * h);
    st->HasAlpha = a->has_alpha;
    st->IsGrayscale = a->is_grayscale;
    st->UniqueColors = a->unique_colors;
    st->MeanBrightness = a->bright_sum / n;                                   // analyze.go:90
    if (a->sample_count > 0) st->Contrast = std::sqrt(a->variance_sum / double(a->sample_count));   // :111-113
    double entropy = 0;                                                       // computeEntropy, :127-139
    for (int i = 0; i < 256; i++) {
        const double count = double(a->histogram[i]);
        if (count > 0) {
            const double p = count / n;
            entropy -= p * std::log2(p);
        }
    }
    st->Entropy = entropy;
    if (a->edge_total > 0) st->EdgeDensity = double(a->edge_count) / double(a->edge_total);   // :180-183
    // recommendFormat (:191-202)
    if (st->HasAlpha) st->RecommendedFormat = 2;
    else if (st->UniqueColors <= 256) st->RecommendedFormat = 2;
    else if (st->EdgeDensity > 0.3 && st->UniqueColors < 1000) st->RecommendedFormat = 2;
    else st->RecommendedFormat = 1;
    // recommendQuality (:204-215)
    if (st->Entropy > 6 && st->EdgeDensity < 0.15) st->RecommendedQuality = 0;
    else if (st->Entropy < 4) st->RecommendedQuality = 4;
    else if (st->EdgeDensity > 0.25) st->RecommendedQuality = 3;
    else st->RecommendedQuality = 0;
    // estimateCompression (:217-230)
    if (st->RecommendedFormat == 2) {
        if (st->UniqueColors <= 256) st->EstimatedCompression = 5.0 + (256 - double(st->UniqueColors)) / 50;
        else if (st->IsGrayscale) st->EstimatedCompression = 3.0;
        else st->EstimatedCompression = 2.0;
    } else {
        double base = 10.0;
        if (st->Entropy > 7) base = 5.0;
        else if (st->Entropy > 5) base = 8.0;
        if (st->EdgeDensity > 0.2) base *= 0.7;
        st->EstimatedCompression = base;
    }
}

int fennec_Analyze(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h, fennec_ImageStats *out)
{
    if (!out) return FNX_ERR_INVALID;
    fnx_analysis a;
    const int rc = fnx_analyze(ctx, space, src, sstride, w, h, &a);
    if (rc < 0) return rc;
    fennec_statsFromAnalysis(&a, w, h, out);
    return rc;
}

static int flat_scan(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h, int *opaque, int *gray)
{
    const size_t len = (w > 0 && h > 0) ? size_t(h - 1) * size_t(sstride) + size_t(w) * 4 : 0;   // image.NRGBA Pix of a (sub)image
    return fnx_scan_flags(ctx, space, src, len, opaque, gray);
}

int fennec_isOpaque(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h, int *out)
{
    return flat_scan(ctx, space, src, sstride, w, h, out, nullptr);
}

int fennec_isGrayscale(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h, int *out)
{
    return flat_scan(ctx, space, src, sstride, w, h, nullptr, out);
}

}  // extern "C"
