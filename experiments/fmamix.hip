// Experiment (r3): can v_fma_mix_f32 feed the blur's FMAs straight from integer byte fields?
//   * f16 DENORMAL inputs: a 16-bit field holding an integer v <= 1023 IS the f16 denormal v * 2^-24.  If
//     v_fma_mix_f32 honours f16 denormals, acc += (w * 2^24) * field needs no u8 -> f32 convert at all, and the
//     products are bit-identical to fmaf(w, (float)v, acc) (scaling by a power of two is exact).
//   * issue cost of v_fma_mix_f32 (lo / hi half), v_add_u32, v_pk_add_u16, v_dot2_f32_f16 next to v_fma_f32.
//   hipcc --offload-arch=gfx950 -O3 -o experiments/fmamix_bin experiments/fmamix.hip && experiments/fmamix_bin
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>

constexpr int ITERS = 4096;

template <int KIND, int UNROLL>
__global__ __launch_bounds__(256) void k(float *out, float seed, unsigned useed)
{
    float f[UNROLL];
    unsigned u[UNROLL];
    for (int i = 0; i < UNROLL; i++) { f[i] = seed + i + threadIdx.x; u[i] = (useed * (i + 1) + threadIdx.x) & 0x00ff00ffu; }
    const float w = 1.0000001f;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            if (KIND == 0) f[i] = __builtin_fmaf(f[i], w, 0.25f);
            if (KIND == 1) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "+v"(f[i]) : "v"(w), "v"(u[i]));
            if (KIND == 2) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(f[i]) : "v"(w), "v"(u[i]));
            if (KIND == 3) { unsigned t; asm volatile("v_add_u32 %0, %1, %2" : "=v"(t) : "v"(u[i]), "v"(useed)); u[i] = t; }
            if (KIND == 4) { unsigned t; asm volatile("v_pk_add_u16 %0, %1, %2" : "=v"(t) : "v"(u[i]), "v"(useed)); u[i] = t; }
            if (KIND == 5) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(f[i]) : "v"(useed), "v"(u[i]));
            if (KIND == 6) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "+v"(f[i]) : "s"(w), "v"(u[i]));
            if (KIND == 7) { float t; asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(t) : "v"(u[i])); u[i] = __float_as_uint(t) & 0x3ffu; }
            if (KIND == 8) { unsigned t; asm volatile("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(t) : "v"(u[i]), "v"(useed)); u[i] = t; }
            if (KIND == 9) { unsigned t; asm volatile("v_alignbit_b32 %0, %1, %2, 8" : "=v"(t) : "v"(u[i]), "v"(useed)); u[i] = t; }
            if (KIND == 10) { unsigned t; asm volatile("v_alignbyte_b32 %0, %1, %2, 1" : "=v"(t) : "v"(u[i]), "v"(useed)); u[i] = t; }
            if (KIND == 11) { unsigned t; asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(t) : "v"(u[i]), "v"(useed), "v"(u[(i + 1) % UNROLL])); u[i] = t; }
            if (KIND == 12) { unsigned t; asm volatile("v_pk_fma_f16 %0, %1, %2, %1" : "=v"(t) : "v"(u[i]), "v"(useed)); u[i] = t; }
        }
    }
    float s = 0;
    for (int i = 0; i < UNROLL; i++) s += f[i] + u[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// semantics: every byte value through both halves against fmaf on the converted value, with a weight scaled by 2^24
__global__ void check(unsigned *bad, float *sample)
{
    const unsigned v = threadIdx.x;                // 0..1023: 10-bit fields (pair sums of bytes reach 510)
    const float w = 0.1234567f, acc0 = 0.4998f;
    const float ws = w * 16777216.0f;
    const unsigned word = v | ((1023u - v) << 16);
    float lo = acc0, hi = acc0;
    asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "+v"(lo) : "v"(ws), "v"(word));
    asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(hi) : "v"(ws), "v"(word));
    const float want_lo = __builtin_fmaf(w, static_cast<float>(v), acc0);
    const float want_hi = __builtin_fmaf(w, static_cast<float>(1023u - v), acc0);
    if (lo != want_lo || hi != want_hi) atomicAdd(bad, 1u);
    if (v == 200) { sample[0] = lo; sample[1] = want_lo; sample[2] = hi; sample[3] = want_hi; }
}

template <int KIND, int UNROLL>
static void run(const char *name)
{
    float *d;
    (void)hipMalloc(&d, sizeof(float) * 256 * 4096);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 40; rep++) hipLaunchKernelGGL((k<KIND, UNROLL>), dim3(4096), dim3(256), 0, 0, d, 1.0f, 77u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int rep = 0; rep < 10; rep++) hipLaunchKernelGGL((k<KIND, UNROLL>), dim3(4096), dim3(256), 0, 0, d, 1.0f, 77u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = 10.0 * 4096 * 4 * double(ITERS) * UNROLL;
    const double per_simd_per_s = wave_instr / (ms * 1e-3) / (256.0 * 4);
    printf("%-44s x%-2d %8.3f ms -> %.2f clk each at 2.4 GHz\n", name, UNROLL, ms / 10, 2.4e9 / per_simd_per_s);
    (void)hipFree(d);
}

int main()
{
    unsigned *bad;
    float *sample;
    (void)hipMalloc(&bad, 4);
    (void)hipMalloc(&sample, 16);
    (void)hipMemset(bad, 0, 4);
    hipLaunchKernelGGL(check, dim3(1), dim3(1024), 0, 0, bad, sample);
    unsigned hbad;
    float hs[4];
    (void)hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hs, sample, 16, hipMemcpyDeviceToHost);
    printf("f16-denormal fma_mix vs fmaf on the converted value: %u of 1024 fields differ (v=200: lo %.9g want %.9g, hi %.9g want %.9g)\n",
           hbad, hs[0], hs[1], hs[2], hs[3]);
    run<0, 16>("v_fma_f32");
    run<1, 16>("v_fma_mix_f32 (f16 lo, denormal data)");
    run<2, 16>("v_fma_mix_f32 (f16 hi, denormal data)");
    run<6, 16>("v_fma_mix_f32 (weight in an SGPR)");
    run<3, 16>("v_add_u32");
    run<4, 16>("v_pk_add_u16");
    run<11, 16>("v_add3_u32");
    run<5, 16>("v_dot2_f32_f16");
    run<12, 16>("v_pk_fma_f16");
    run<7, 16>("v_cvt_f32_f16 + v_and");
    run<8, 16>("v_lshl_or_b32");
    run<9, 16>("v_alignbit_b32");
    run<10, 16>("v_alignbyte_b32");
    run<1, 1>("v_fma_mix_f32 dependent");
    run<1, 2>("v_fma_mix_f32 dependent");
    run<1, 4>("v_fma_mix_f32 dependent");
    run<0, 1>("v_fma_f32 dependent");
    return 0;
}
