// kbench: kernel-level timing of the batched C-ABI entry points with HIP events on the ctx
// stream (development tool; bench.py is the contract benchmark).
//   kbench [what] [batch] [iters] [W] [H]   what = blur | ssim | both | single | onepass | pipelined | overlap ...
// Variants are selected inside the library with FNX_* environment variables.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../include/fennec_hip.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define FK(x) do { int r = (x); if (r < 0) { fprintf(stderr, "%s: %s\n", #x, fnx_last_error()); exit(1); } } while (0)

int main(int argc, char **argv)
{
    std::string what = argc > 1 ? argv[1] : "both";
    int B = argc > 2 ? atoi(argv[2]) : 32;
    int iters = argc > 3 ? atoi(argv[3]) : 20;
    int W = argc > 4 ? atoi(argv[4]) : 3840, H = argc > 5 ? atoi(argv[5]) : 2160;
    fnx_ctx *ctx;
    FK(fnx_ctx_create(0, &ctx));
    hipStream_t st = (hipStream_t)fnx_ctx_stream(ctx);
    size_t S = (size_t)W * H * 4;
    std::vector<uint8_t> host(S);
    std::vector<const uint8_t *> srcs(B);
    std::vector<uint8_t *> dsts(B);
    for (int k = 0; k < B; k++) {
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                uint8_t *p = &host[((size_t)y * W + x) * 4];
                p[0] = (uint8_t)((x * y + 3 * x + 17 * k) & 255);
                p[1] = (uint8_t)((x * y + 7 * y + 31 * k) & 255);
                p[2] = (uint8_t)((x + 11 * y + 5 * k) & 255);
                p[3] = 255;
            }
        void *d, *o;
        FK(fnx_malloc(ctx, S, &d));
        FK(fnx_malloc(ctx, S, &o));
        FK(fnx_upload(ctx, d, W * 4, host.data(), W * 4, W, H));
        srcs[k] = (const uint8_t *)d;
        dsts[k] = (uint8_t *)o;
    }
    double sigma = 2.0;
    int radius = fennec_blurKernel(sigma, nullptr);
    std::vector<double> kern(2 * radius + 1);
    fennec_blurKernel(sigma, kern.data());
    double win[64];
    fennec_gaussianKernel(8, 1.5, win);
    std::vector<double> out(B);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double mp = (double)W * H / 1e6;

    auto time_it = [&](const char *name, auto &&fn, double bytes_per_image) {
        for (int i = 0; i < 3; i++) fn();
        FK(fnx_ctx_sync(ctx));
        float tot = 0, best = 1e30f;
        for (int i = 0; i < iters; i++) {
            CK(hipEventRecord(e0, st));
            fn();
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            tot += ms;
            if (ms < best) best = ms;
        }
        double avg = tot / iters;
        printf("%-28s B=%d avg %.3f ms (%.2f us/img, %.0f MP/s, %.0f GB/s algorithmic)  best %.3f ms (%.2f us/img)\n", name, B,
               avg, avg * 1e3 / B, mp * B / (avg * 1e-3), bytes_per_image * B / (avg * 1e-3) / 1e9, best, best * 1e3 / B);
    };

    if (what == "onepass") {   // config 2 from a C++ host: wall clock over `iters` blocking calls after a clock pre-warm
        auto fn = [&] { FK(fnx_gaussian_blur_ssim_fast_batch(ctx, B, srcs.data(), W * 4, W, H, kern.data(), radius, FNX_BLUR_FAST,
                                                             dsts.data(), W * 4, win, out.data())); };
        auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.3) fn();
        t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < iters; i++) fn();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("gaussian_blur_ssim_fast_batch  B=%d %dx%d: %.3f ms per call, %.2f us/img, %.0f MP/s (wall clock, %d calls), ssim[0]=%.9f\n", B, W, H,
               dt / iters * 1e3, dt / iters / B * 1e6, mp * B * iters / dt, iters, out[0]);
    }
    if (what == "pipelined") {   // config 2 from a C++ host with two steps in flight: step s+1 is enqueued on a second
                                 // context (its own stream and destinations) before step s's scores are fetched
        fnx_ctx *c2;
        FK(fnx_ctx_create(0, &c2));
        std::vector<uint8_t *> dsts2(B);
        for (int k = 0; k < B; k++) {
            void *o;
            FK(fnx_malloc(c2, S, &o));
            dsts2[k] = (uint8_t *)o;
        }
        fnx_ctx *cs[2] = {ctx, c2};
        uint8_t *const *ds[2] = {dsts.data(), dsts2.data()};
        auto enq = [&](int d) { FK(fnx_gaussian_blur_ssim_fast_batch_enqueue(cs[d], B, srcs.data(), W * 4, W, H, kern.data(), radius,
                                                                               FNX_BLUR_FAST, ds[d], W * 4, win)); };
        auto run = [&](int n) {
            for (int s = 0; s <= n; s++) {
                if (s < n) enq(s & 1);
                if (s >= 1) FK(fnx_results_fetch(cs[(s - 1) & 1], B, out.data()));
            }
        };
        auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.3) run(4);
        t0 = std::chrono::steady_clock::now();
        run(iters);
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("gaussian_blur_ssim_fast_batch, 2 steps in flight  B=%d %dx%d: %.3f ms per step, %.2f us/img, %.0f MP/s (wall clock, %d steps), ssim[0]=%.9f\n",
               B, W, H, dt / iters * 1e3, dt / iters / B * 1e6, mp * B * iters / dt, iters, out[0]);
        fnx_ctx_destroy(c2);
    }
    if (what == "blur" || what == "both")
        time_it("gaussian_blur_batch", [&] { FK(fnx_gaussian_blur_batch(ctx, B, srcs.data(), W * 4, W, H, kern.data(), radius, FNX_BLUR_FAST, dsts.data(), W * 4)); }, 2.0 * S);
    if (what == "ssim" || what == "both")
        time_it("ssim_fast_batch", [&] { FK(fnx_ssim_fast_batch(ctx, B, srcs.data(), W * 4, dsts.data(), W * 4, W, H, win, out.data())); }, 2.0 * S);
    if (what == "single" || what == "both") {
        int k = 0;
        time_it("gaussian_blur x B (per call)", [&] { for (int i = 0; i < B; i++) FK(fnx_gaussian_blur(ctx, FNX_DEVICE, srcs[i], W * 4, W, H, kern.data(), radius, FNX_BLUR_FAST, dsts[i], W * 4)); }, 2.0 * S);
        time_it("ssim_fast x B (per call)", [&] { for (int i = 0; i < B; i++) FK(fnx_ssim_fast(ctx, FNX_DEVICE, srcs[i], W * 4, dsts[i], W * 4, W, H, win, &out[i])); }, 2.0 * S);
        (void)k;
    }
    if (what == "overlap") {    // two contexts, half a batch each, staggered: blur (VALU bound) of one
                                // overlaps SSIMFast (load bound) of the other
        fnx_ctx *c2;
        FK(fnx_ctx_create(0, &c2));
        int hB = B / 2;
        std::vector<double> o2(B);
        auto seq = [&] {
            FK(fnx_gaussian_blur_batch(ctx, B, srcs.data(), W * 4, W, H, kern.data(), radius, FNX_BLUR_FAST, dsts.data(), W * 4));
            FK(fnx_ssim_fast_batch(ctx, B, srcs.data(), W * 4, dsts.data(), W * 4, W, H, win, out.data()));
        };
        auto ovl = [&] {
            // ctx: half A, c2: half B
            FK(fnx_gaussian_blur_batch(ctx, hB, srcs.data(), W * 4, W, H, kern.data(), radius, FNX_BLUR_FAST, dsts.data(), W * 4));
            FK(fnx_ssim_fast_batch_enqueue(ctx, hB, srcs.data(), W * 4, dsts.data(), W * 4, W, H, win));
            FK(fnx_gaussian_blur_batch(c2, hB, srcs.data() + hB, W * 4, W, H, kern.data(), radius, FNX_BLUR_FAST, dsts.data() + hB, W * 4));
            FK(fnx_ssim_fast_batch_enqueue(c2, hB, srcs.data() + hB, W * 4, dsts.data() + hB, W * 4, W, H, win));
            FK(fnx_results_fetch(ctx, hB, o2.data()));
            FK(fnx_results_fetch(c2, hB, o2.data() + hB));
        };
        auto stag = [&] {   // complementary phases: ctx blurs then scores, c2 scores (last iteration's blur) then blurs
            FK(fnx_gaussian_blur_batch(ctx, hB, srcs.data(), W * 4, W, H, kern.data(), radius, FNX_BLUR_FAST, dsts.data(), W * 4));
            FK(fnx_ssim_fast_batch_enqueue(c2, hB, srcs.data() + hB, W * 4, dsts.data() + hB, W * 4, W, H, win));
            FK(fnx_ssim_fast_batch_enqueue(ctx, hB, srcs.data(), W * 4, dsts.data(), W * 4, W, H, win));
            FK(fnx_gaussian_blur_batch(c2, hB, srcs.data() + hB, W * 4, W, H, kern.data(), radius, FNX_BLUR_FAST, dsts.data() + hB, W * 4));
            FK(fnx_results_fetch(ctx, hB, o2.data()));
            FK(fnx_results_fetch(c2, hB, o2.data() + hB));
        };
        for (const char *nm : {"sequential 1 ctx", "2 ctx half batches", "x staggered 2 ctx"}) {
            auto &fn = (nm[0] == 's') ? (std::function<void()> &)*new std::function<void()>(seq) : (nm[0] == 'x') ? (std::function<void()> &)*new std::function<void()>(stag) : (std::function<void()> &)*new std::function<void()>(ovl);
            for (int i = 0; i < 3; i++) fn();
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < iters; i++) fn();
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / iters;
            printf("%-24s %.3f ms per %d images (%.2f us/img, %.0f MP/s)\n", nm, ms, B, ms * 1e3 / B, mp * B / (ms * 1e-3));
        }
        for (int i = 0; i < B; i++) if (out[i] != o2[i]) { printf("MISMATCH %d\n", i); break; }
        fnx_ctx_destroy(c2);
    }
    if (what == "overlap2") {   // fine-grained: stream A blurs chunk k+1 while stream B scores chunk k
        fnx_ctx *c2;
        FK(fnx_ctx_create(0, &c2));
        hipStream_t s2 = (hipStream_t)fnx_ctx_stream(c2);
        for (int chunk : {2, 4, 8, 16}) {
            int nch = B / chunk;
            std::vector<hipEvent_t> evs(nch);
            for (size_t ei = 0; ei < evs.size(); ei++) CK(hipEventCreateWithFlags(&evs[ei], hipEventDisableTiming));
            auto run = [&] {
                for (int c = 0; c < nch; c++) {
                    FK(fnx_gaussian_blur_batch(ctx, chunk, srcs.data() + c * chunk, W * 4, W, H, kern.data(), radius, FNX_BLUR_FAST, dsts.data() + c * chunk, W * 4));
                    CK(hipEventRecord(evs[c], st));
                    CK(hipStreamWaitEvent(s2, evs[c], 0));
                    FK(fnx_ssim_fast_batch_enqueue(c2, chunk, srcs.data() + c * chunk, W * 4, dsts.data() + c * chunk, W * 4, W, H, win));
                }
                FK(fnx_ctx_sync(ctx));
                FK(fnx_ctx_sync(c2));
            };
            for (int i = 0; i < 3; i++) run();
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < iters; i++) run();
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / iters;
            printf("chunk %2d: %.3f ms per %d images (%.2f us/img, %.0f MP/s)\n", chunk, ms, B, ms * 1e3 / B, mp * B / (ms * 1e-3));
        }
        fnx_ctx_destroy(c2);
    }
    if (what == "host" || what == "all") {      // FNX_HOST entry points: pageable host buffers, PCIe both ways
        std::vector<uint8_t> hdst(S);
        double res = 0;
        int nb = B < 4 ? B : 4;
        auto time_host = [&](const char *name, auto &&fn, double bytes) {
            fn();
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < nb; i++) fn();
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / nb;
            printf("%-28s host-space %.3f ms/img (%.0f MP/s, %.1f GB/s over PCIe+kernel)\n", name, ms, mp / (ms * 1e-3), bytes / (ms * 1e-3) / 1e9);
        };
        time_host("gaussian_blur FNX_HOST", [&] { FK(fnx_gaussian_blur(ctx, FNX_HOST, host.data(), W * 4, W, H, kern.data(), radius, FNX_BLUR_FAST, hdst.data(), W * 4)); }, 2.0 * S);
        time_host("ssim_fast FNX_HOST", [&] { FK(fnx_ssim_fast(ctx, FNX_HOST, host.data(), W * 4, hdst.data(), W * 4, W, H, win, &res)); }, 2.0 * S);
        fnx_prepared *pr = nullptr;
        FK(fnx_ssim_fast_prepare(ctx, FNX_HOST, host.data(), W * 4, W, H, &pr));
        time_host("ssim_fast_against FNX_HOST", [&] { FK(fnx_ssim_fast_against(ctx, pr, FNX_HOST, hdst.data(), W * 4, win, &res)); }, 1.0 * S);
        fnx_prepared_free(ctx, pr);
        printf("ssim=%.12f\n", res);
    }
    if (what == "cfg3" || what == "all") {      // BASELINE config 3 pieces (4K -> 1080p Lanczos, MSSSIM)
        void *small, *up;
        FK(fnx_malloc(ctx, (size_t)(W / 2) * (H / 2) * 4, &small));
        FK(fnx_malloc(ctx, S, &up));
        double res = 0;
        time_it("lanczosResize 1/2", [&] { for (int i = 0; i < B; i++) FK(fennec_lanczosResize(ctx, FNX_DEVICE, srcs[i], W * 4, W, H, (uint8_t *)small, W / 2 * 4, W / 2, H / 2)); }, 1.25 * S);
        time_it("lanczosResize x2 (back up)", [&] { for (int i = 0; i < B; i++) FK(fennec_lanczosResize(ctx, FNX_DEVICE, (uint8_t *)small, W / 2 * 4, W / 2, H / 2, (uint8_t *)up, W * 4, W, H)); }, 1.25 * S);
        time_it("MSSSIM (equal dims)", [&] { for (int i = 0; i < B; i++) FK(fennec_MSSSIM(ctx, FNX_DEVICE, srcs[i], W * 4, W, H, (uint8_t *)up, W * 4, W, H, &res)); }, 3.33 * S);
        time_it("MSSSIM (vs half size: cfg3)", [&] { for (int i = 0; i < B; i++) FK(fennec_MSSSIM(ctx, FNX_DEVICE, srcs[i], W * 4, W, H, (uint8_t *)small, W / 2 * 4, W / 2, H / 2, &res)); }, 4.58 * S);
        printf("msssim=%.12f\n", res);
    }
    if (what == "cfg4" || what == "all") {      // BASELINE config 4 pieces (AdaptiveSharpen, full SSIM)
        double res = 0;
        time_it("AdaptiveSharpen", [&] { for (int i = 0; i < B; i++) FK(fennec_AdaptiveSharpen(ctx, FNX_DEVICE, srcs[i], W * 4, W, H, 0.5, dsts[i], W * 4)); }, 2.0 * S);
        time_it("Sharpen", [&] { for (int i = 0; i < B; i++) FK(fennec_Sharpen(ctx, FNX_DEVICE, srcs[i], W * 4, W, H, 0.5, dsts[i], W * 4)); }, 2.0 * S);
        time_it("SSIM (full res)", [&] { for (int i = 0; i < B; i++) FK(fennec_SSIM(ctx, FNX_DEVICE, srcs[i], W * 4, W, H, dsts[i], W * 4, W, H, &res)); }, 2.0 * S);
        time_it("ApplyOrientation 6", [&] { for (int i = 0; i < B; i++) FK(fennec_ApplyOrientation(ctx, FNX_DEVICE, srcs[i], W * 4, W, H, 6, dsts[i], H * 4)); }, 2.0 * S);
        time_it("ApplyOrientation 3", [&] { for (int i = 0; i < B; i++) FK(fennec_ApplyOrientation(ctx, FNX_DEVICE, srcs[i], W * 4, W, H, 3, dsts[i], W * 4)); }, 2.0 * S);
        time_it("GaussianBlur exact", [&] { for (int i = 0; i < B; i++) FK(fnx_gaussian_blur(ctx, FNX_DEVICE, srcs[i], W * 4, W, H, kern.data(), radius, FNX_BLUR_EXACT, dsts[i], W * 4)); }, 2.0 * S);
        printf("ssim=%.12f\n", res);
    }
    printf("ssim[0]=%.12f\n", out[0]);
    fnx_ctx_destroy(ctx);
    return 0;
}
