// Experiment: what the exact loops of resize.hip cost per fp64 instruction -- six accumulator chains, per tap a
// conversion-like add, a multiply and a dependent add (no contraction), at 1 / 2 / 4 / 8 resident waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o fp64chain experiments/fp64chain.hip && ./fp64chain
#include <hip/hip_runtime.h>

#include <cstdio>

constexpr int ITERS = 2048, TAPS = 16;

template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, const double *w, double seed)
{
    __shared__ double s_w[2 * TAPS * 64];
    for (int i = threadIdx.x; i < 2 * TAPS * 64; i += 256) s_w[i] = w[i % 64] + i * 1e-9;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    double r0 = 0, g0 = 0, b0 = 0, r1 = 0, g1 = 0, b1 = 0;
    double fr = seed + threadIdx.x, fg = fr + 1, fb = fr + 2;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < TAPS; i++) {
            double aw0, aw1;
            if (MODE == 0) { aw0 = s_w[i * 64 + lane]; aw1 = s_w[(TAPS + i) * 64 + lane]; }     // weights from LDS, as the H exact loop
            else { aw0 = seed * (i + 1); aw1 = seed * (i + 2); }                                   // weights in registers / constants
            fr = fr + 4503599627370496.0 - 4503599627370496.0;                                     // the magic-number conversion's add
            fg = fg + 4503599627370496.0 - 4503599627370496.0;
            fb = fb + 4503599627370496.0 - 4503599627370496.0;
            r0 = r0 + fr * aw0; g0 = g0 + fg * aw0; b0 = b0 + fb * aw0;
            r1 = r1 + fr * aw1; g1 = g1 + fg * aw1; b1 = b1 + fb * aw1;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = r0 + g0 + b0 + r1 + g1 + b1;
}

template <int MODE>
static void run(const char *name, int blocks)
{
    double *d, *w;
    (void)hipMalloc(&d, sizeof(double) * 256 * 8192);
    (void)hipMalloc(&w, sizeof(double) * 64);
    (void)hipMemset(w, 0, sizeof(double) * 64);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 5; rep++) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, d, w, 1.5);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int rep = 0; rep < 5; rep++) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, d, w, 1.5);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // per tap: 6 conversion adds (2 per channel) + 6 mul + 6 add = 18 fp64 instructions
    const double per_wave = double(ITERS) * TAPS * 18;
    const double waves_per_simd = blocks * 4 / 1024.0;
    const double clk = (ms / 5 * 1e-3) * 2.4e9 / (per_wave * waves_per_simd);
    printf("%-28s %5d blocks (%.0f waves/SIMD) %8.3f ms  %.2f clk per fp64 instruction per SIMD at 2.4 GHz\n", name, blocks, waves_per_simd, ms / 5, clk);
}

int main()
{
    for (int b : {256, 512, 1024, 2048}) run<0>("weights from LDS", b);
    for (int b : {256, 512, 1024, 2048}) run<1>("weights in registers", b);
}
