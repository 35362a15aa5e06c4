// Experiment (r3): what does a u8 -> f64 convert cost on gfx950?  The exact (reference-order fp64) loops of resize.hip spend
// 2-3 instructions per channel on it ((2^52 | v) - 2^52: byte extract + v_add_f64).
//   hipcc --offload-arch=gfx950 -O3 -o experiments/cvt64_bin experiments/cvt64.hip && experiments/cvt64_bin
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITERS = 4096;
template <int KIND, int UNROLL>
__global__ __launch_bounds__(256) void k(double *out, unsigned useed)
{
    unsigned u[UNROLL]; double d[UNROLL]; float f[UNROLL];
    for (int i = 0; i < UNROLL; i++) { u[i] = useed * (i + 1) + threadIdx.x; d[i] = u[i]; f[i] = u[i]; }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            if (KIND == 0) { double t; asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(t) : "v"(u[i])); d[i] = t; }
            if (KIND == 1) { double t; asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(t) : "v"(f[i])); d[i] = t; }
            if (KIND == 2) { double t; asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(t) : "v"(u[i])); d[i] = t; }
            if (KIND == 3) { double t; asm volatile("v_add_f64 %0, %1, %1" : "=v"(t) : "v"(d[i])); d[i] = t; }
            if (KIND == 4) { double t; asm volatile("v_trunc_f64 %0, %1" : "=v"(t) : "v"(d[i])); d[i] = t; }
            if (KIND == 5) { unsigned t; asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(t) : "v"(d[i])); u[i] = t; }
            if (KIND == 6) { double t; asm volatile("v_mul_f64 %0, %1, %1" : "=v"(t) : "v"(d[i])); d[i] = t; }
            if (KIND == 7) { double t; asm volatile("v_fma_f64 %0, %1, %1, %1" : "=v"(t) : "v"(d[i])); d[i] = t; }
            if (KIND == 8) { double t; asm volatile("v_min_f64 %0, %1, %1" : "=v"(t) : "v"(d[i])); d[i] = t; }
            if (KIND == 9) { double t; asm volatile("v_rndne_f64 %0, %1" : "=v"(t) : "v"(d[i])); d[i] = t; }
            if (KIND == 10) { double t; asm volatile("v_floor_f64 %0, %1" : "=v"(t) : "v"(d[i])); d[i] = t; }
            if (KIND == 11) { unsigned t; asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(t) : "v"(d[i])); u[i] = t; }
        }
    }
    double s = 0;
    for (int i = 0; i < UNROLL; i++) s += d[i] + u[i] + f[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND>
static void run(const char *name)
{
    constexpr int UNROLL = 12;
    double *d; (void)hipMalloc(&d, sizeof(double) * 256 * 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 20; rep++) hipLaunchKernelGGL((k<KIND, UNROLL>), dim3(4096), dim3(256), 0, 0, d, 77u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int rep = 0; rep < 10; rep++) hipLaunchKernelGGL((k<KIND, UNROLL>), dim3(4096), dim3(256), 0, 0, d, 77u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = 10.0 * 4096 * 4 * double(ITERS) * UNROLL;
    printf("%-28s %8.3f ms -> %.2f clk each at 2.4 GHz\n", name, ms / 10, 2.4e9 / (wave_instr / (ms * 1e-3) / 1024.0));
    (void)hipFree(d);
}
int main()
{
    run<3>("v_add_f64"); run<6>("v_mul_f64"); run<7>("v_fma_f64"); run<8>("v_min_f64");
    run<0>("v_cvt_f64_u32"); run<2>("v_cvt_f64_i32"); run<1>("v_cvt_f64_f32");
    run<4>("v_trunc_f64"); run<9>("v_rndne_f64"); run<10>("v_floor_f64"); run<5>("v_cvt_u32_f64"); run<11>("v_cvt_i32_f64");
    return 0;
}
