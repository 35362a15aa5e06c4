// Where do the microseconds of ONE device-space C-ABI call go?  enqueue (the call returns) | wait (fnx_ctx_sync) | and, for
// comparison, a stream of 50 calls with one sync (per-call cost when nothing waits).
// hipcc -O2 -std=c++17 --offload-arch=gfx950 callcost.cpp -o callcost -L../../fennec_amd -lfennec_hip -Wl,-rpath,'$ORIGIN/../../fennec_amd'
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>
#include "../../include/fennec_hip.h"
#define FK(x) do { int r_ = (x); if (r_ < 0) { fprintf(stderr, "%s: %s\n", #x, fnx_last_error()); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const int W = 3840, H = 2160;
    fnx_ctx *ctx; FK(fnx_ctx_create(0, &ctx));
    const size_t S = (size_t)W * H * 4;
    std::vector<uint8_t> host(S);
    for (size_t i = 0; i < S; i++) host[i] = (uint8_t)((i * 2654435761u) >> 13);
    void *d, *o; FK(fnx_malloc(ctx, S, &d)); FK(fnx_malloc(ctx, S, &o));
    FK(fnx_upload(ctx, d, W * 4, host.data(), W * 4, W, H));
    const int radius = fennec_blurKernel(2.0, nullptr);
    std::vector<double> kern(2 * radius + 1); fennec_blurKernel(2.0, kern.data());
    struct Op { const char *name; std::function<void()> fn; };
    std::vector<Op> ops = {
        {"GaussianBlur s=2", [&] { FK(fnx_gaussian_blur(ctx, FNX_DEVICE, (uint8_t *)d, W * 4, W, H, kern.data(), radius, 0, (uint8_t *)o, W * 4)); }},
        {"Sharpen", [&] { FK(fnx_sharpen(ctx, FNX_DEVICE, (uint8_t *)d, W * 4, W, H, 0.5, (uint8_t *)o, W * 4)); }},
        {"lanczosResize 1/2", [&] { FK(fennec_lanczosResize(ctx, FNX_DEVICE, (uint8_t *)d, W * 4, W, H, (uint8_t *)o, (W / 2) * 4, W / 2, H / 2)); }},
        {"boxDownsample", [&] { FK(fnx_box_downsample(ctx, FNX_DEVICE, (uint8_t *)d, W * 4, W, H, (uint8_t *)o, 512 * 4, 512, 288)); }},
    };
    printf("%-20s %10s %10s %10s %14s\n", "op", "enqueue us", "wait us", "total us", "50 in a row us");
    for (auto &op : ops) {
        for (int i = 0; i < 300; i++) { op.fn(); FK(fnx_ctx_sync(ctx)); }
        std::vector<double> te, tw;
        for (int i = 0; i < 400; i++) {
            const double a = now(); op.fn(); const double b = now(); FK(fnx_ctx_sync(ctx)); const double c = now();
            te.push_back(b - a); tw.push_back(c - b);
        }
        std::sort(te.begin(), te.end()); std::sort(tw.begin(), tw.end());
        std::vector<double> tr;
        for (int i = 0; i < 20; i++) { const double a = now(); for (int k = 0; k < 50; k++) op.fn(); FK(fnx_ctx_sync(ctx)); tr.push_back((now() - a) / 50); }
        std::sort(tr.begin(), tr.end());
        printf("%-20s %10.1f %10.1f %10.1f %14.1f\n", op.name, 1e6 * te[200], 1e6 * tw[200], 1e6 * (te[200] + tw[200]), 1e6 * tr[10]);
    }
    return 0;
}
