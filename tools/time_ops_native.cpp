// time_ops_native: wall-clock latency of ONE C-ABI call at a time on device-resident 4K images, from a C++ host -- what a
// cgo caller of libfennec_hip.so pays per call (tools/time_ops.py measures the same calls through the Python binding,
// whose marshalling adds 5-15 us).  Scalar-returning ops include their result's arrival on the host; image ops are
// followed by fnx_ctx_sync.   time_ops_native [W H]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#include "../include/fennec_hip.h"

#define FK(x) do { int r_ = (x); if (r_ < 0) { fprintf(stderr, "%s: %s\n", #x, fnx_last_error()); exit(1); } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const int W = argc > 2 ? atoi(argv[1]) : 3840, H = argc > 2 ? atoi(argv[2]) : 2160, NI = 8;
    fnx_ctx *ctx;
    FK(fnx_ctx_create(0, &ctx));
    const size_t S = (size_t)W * H * 4;
    std::vector<uint8_t> host(S);
    std::vector<uint8_t *> img(NI), blur(NI), out(NI);
    double sigma = 2.0;
    const int radius = fennec_blurKernel(sigma, nullptr);
    std::vector<double> kern(2 * radius + 1);
    fennec_blurKernel(sigma, kern.data());
    double win[64];
    fennec_gaussianKernel(8, 1.5, win);
    for (int k = 0; k < NI; k++) {
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                uint8_t *p = &host[((size_t)y * W + x) * 4];
                p[0] = (uint8_t)((x * y + 3 * x + 17 * k) & 255);
                p[1] = (uint8_t)((x * y + 7 * y + 31 * k) & 255);
                p[2] = (uint8_t)((x + 11 * y + 5 * k) & 255);
                p[3] = 255;
            }
        void *d, *b, *o;
        FK(fnx_malloc(ctx, S, &d)); FK(fnx_malloc(ctx, S, &b)); FK(fnx_malloc(ctx, S, &o));
        FK(fnx_upload(ctx, d, W * 4, host.data(), W * 4, W, H));
        img[k] = (uint8_t *)d; blur[k] = (uint8_t *)b; out[k] = (uint8_t *)o;
        FK(fnx_gaussian_blur(ctx, FNX_DEVICE, img[k], W * 4, W, H, kern.data(), radius, 0, blur[k], W * 4));
    }
    FK(fnx_ctx_sync(ctx));
    double r = 0;
    fnx_analysis an;
    int fo = 0, fg = 0;
    struct Op { const char *name; std::function<void(int)> fn; double bytes; bool sync; };
    std::vector<Op> ops = {
        {"GaussianBlur sigma=2 (fast)", [&](int k) { FK(fnx_gaussian_blur(ctx, FNX_DEVICE, img[k], W * 4, W, H, kern.data(), radius, 0, out[k], W * 4)); }, 2.0 * S, true},
        {"GaussianBlur sigma=2 (exact)", [&](int k) { FK(fnx_gaussian_blur(ctx, FNX_DEVICE, img[k], W * 4, W, H, kern.data(), radius, FNX_BLUR_EXACT, out[k], W * 4)); }, 2.0 * S, true},
        {"Sharpen 0.5", [&](int k) { FK(fnx_sharpen(ctx, FNX_DEVICE, img[k], W * 4, W, H, 0.5, out[k], W * 4)); }, 2.0 * S, true},
        {"AdaptiveSharpen 0.5", [&](int k) { FK(fnx_adaptive_sharpen(ctx, FNX_DEVICE, img[k], W * 4, W, H, 2.0, out[k], W * 4)); }, 2.0 * S, true},
        {"lanczosResize -> 1/2", [&](int k) { FK(fennec_lanczosResize(ctx, FNX_DEVICE, img[k], W * 4, W, H, out[k], (W / 2) * 4, W / 2, H / 2)); }, 1.25 * S, true},
        {"boxDownsample -> 512x288", [&](int k) { FK(fnx_box_downsample(ctx, FNX_DEVICE, img[k], W * 4, W, H, out[k], 512 * 4, 512, 288)); }, 1.0 * S, true},
        {"SSIMFast", [&](int k) { FK(fnx_ssim_fast(ctx, FNX_DEVICE, img[k], W * 4, blur[k], W * 4, W, H, win, &r)); }, 2.0 * S, false},
        {"GaussianBlur + SSIMFast (2 calls)", [&](int k) { FK(fnx_gaussian_blur(ctx, FNX_DEVICE, img[k], W * 4, W, H, kern.data(), radius, 0, out[k], W * 4));
                                                          FK(fnx_ssim_fast(ctx, FNX_DEVICE, img[k], W * 4, out[k], W * 4, W, H, win, &r)); }, 4.0 * S, false},
        {"... with FNX_BLUR_KEEP_BOX_SUMS", [&](int k) { FK(fnx_gaussian_blur(ctx, FNX_DEVICE, img[k], W * 4, W, H, kern.data(), radius, FNX_BLUR_KEEP_BOX_SUMS, out[k], W * 4));
                                                        FK(fnx_ssim_fast(ctx, FNX_DEVICE, img[k], W * 4, out[k], W * 4, W, H, win, &r)); }, 4.0 * S, false},
        {"GaussianBlur + SSIMFast (1 call, r6)", [&](int k) { FK(fnx_gaussian_blur_ssim_fast(ctx, FNX_DEVICE, img[k], W * 4, W, H, kern.data(), radius, 0, out[k], W * 4, win, &r)); }, 4.0 * S, false},
        {"SSIM (full resolution)", [&](int k) { FK(fnx_ssim(ctx, FNX_DEVICE, img[k], W * 4, blur[k], W * 4, W, H, win, &r)); }, 2.0 * S, false},
        {"MSSSIM", [&](int k) { FK(fnx_msssim(ctx, FNX_DEVICE, img[k], W * 4, blur[k], W * 4, W, H, win, &r, nullptr)); }, 3.33 * S, false},
        {"Analyze", [&](int k) { FK(fnx_analyze(ctx, FNX_DEVICE, img[k], W * 4, W, H, &an)); }, 1.0 * S, false},
        {"isOpaque", [&](int k) { FK(fnx_scan_flags(ctx, FNX_DEVICE, img[k], S, &fo, &fg)); }, 1.0 * S, false},
    };
    printf("%dx%d, device-resident, one C-ABI call at a time from a C++ host (launch + result latency included)\n", W, H);
    printf("%-34s %9s %9s %8s %9s\n", "op", "us/call", "min us", "GB/s", "of 8 TB/s");
    for (auto &op : ops) {
        double t0 = now();
        while (now() - t0 < 0.25) { op.fn(0); if (op.sync) FK(fnx_ctx_sync(ctx)); }
        const int n = 200;
        std::vector<double> ts(n);
        for (int i = 0; i < n; i++) {
            const double a = now();
            op.fn(i % NI);
            if (op.sync) FK(fnx_ctx_sync(ctx));
            ts[i] = now() - a;
        }
        double mean = 0;
        for (double t : ts) mean += t;
        mean /= n;
        const double mn = *std::min_element(ts.begin(), ts.end());
        printf("%-34s %9.1f %9.1f %8.0f %9.3f\n", op.name, mean * 1e6, mn * 1e6, op.bytes / mean / 1e9, op.bytes / mean / 8e12);
    }
    (void)r; (void)fo; (void)fg;
    fnx_ctx_destroy(ctx);
    return 0;
}
