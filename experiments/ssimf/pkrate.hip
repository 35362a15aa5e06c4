// r6 experiment: what a wave of the fp32 SSIM march can issue when only 2 (or 1, 4) waves share a SIMD.
//   KIND 0: 32 independent v_pk_fma_f32 (VGPR pair x SGPR pair + VGPR pair) per iteration   (the vertical pass)
//   KIND 1: 64 independent v_fma_f32 (the same FMAs unpacked)
//   KIND 2: 4 chains x 8 dependent v_pk_fma_f32 (the horizontal pass)
//   KIND 3: 8 chains x 8 dependent v_fma_f32
//   KIND 4: 32 independent v_fma_f64
//   hipcc --offload-arch=gfx950 -O3 -o pkrate experiments/ssimf/pkrate.hip && ./pkrate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITERS = 2048;

template <int KIND>
__global__ __launch_bounds__(256, 2) void k(float *out, float s0, float s1, float s2, float s3)
{
    v2f m[32];
    float f[64];
    double d[32];
    for (int i = 0; i < 32; i++) { m[i] = (v2f){s0 + i + threadIdx.x, s1 + i}; d[i] = s0 + i + threadIdx.x; }
    for (int i = 0; i < 64; i++) f[i] = s0 + i + threadIdx.x;
    const v2f c0 = {s0, s0}, c1 = {s1, s1}, c2 = {s2, s2}, c3 = {s3, s3};
    v2f h = {s2 + threadIdx.x, s3};
    float hs = s2 + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
        if (KIND == 0) {
#pragma unroll
            for (int i = 0; i < 32; i++) m[i] = __builtin_elementwise_fma(h, (i & 3) == 0 ? c0 : (i & 3) == 1 ? c1 : (i & 3) == 2 ? c2 : c3, m[i]);
        }
        if (KIND == 1) {
#pragma unroll
            for (int i = 0; i < 64; i++) f[i] = __builtin_fmaf(hs, (i & 3) == 0 ? s0 : (i & 3) == 1 ? s1 : (i & 3) == 2 ? s2 : s3, f[i]);
        }
        if (KIND == 2) {
#pragma unroll
            for (int t = 0; t < 8; t++)
#pragma unroll
                for (int i = 0; i < 4; i++) m[i] = __builtin_elementwise_fma(m[8 + 4 * t + (i & 1)], (t & 1) ? c0 : c1, m[i]);
        }
        if (KIND == 3) {
#pragma unroll
            for (int t = 0; t < 8; t++)
#pragma unroll
                for (int i = 0; i < 8; i++) f[i] = __builtin_fmaf(f[16 + 4 * t + (i & 3)], (t & 1) ? s0 : s1, f[i]);
        }
        if (KIND == 4) {
#pragma unroll
            for (int i = 0; i < 32; i++) d[i] = __builtin_fma((double)hs, (i & 1) ? (double)s0 : (double)s1, d[i]);
        }
        asm volatile("" : "+v"(h), "+v"(hs));
    }
    float s = 0;
    for (int i = 0; i < 32; i++) s += m[i].x + m[i].y + (float)d[i];
    for (int i = 0; i < 64; i++) s += f[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
static void run(const char *name, int instr_per_iter, int blocks_per_cu)
{
    float *d;
    (void)hipMalloc(&d, sizeof(float) * 256 * 256 * 16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int blocks = 256 * blocks_per_cu;
    for (int rep = 0; rep < 20; rep++) hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, d, 1.0f, 0.5f, 0.25f, 0.125f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int rep = 0; rep < 10; rep++) hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, d, 1.0f, 0.5f, 0.25f, 0.125f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // every SIMD holds blocks_per_cu waves (one wave of each block): the kernel's duration = one wave's run
    const double clk_per_instr = (ms / 10 * 1e-3) * 2.4e9 / (double(ITERS) * instr_per_iter * blocks_per_cu);
    printf("%-44s %d waves/SIMD %8.3f ms  %.2f clk per wave-instruction per SIMD at 2.4 GHz\n", name, blocks_per_cu, ms / 10, clk_per_instr);
    (void)hipFree(d);
}

int main()
{
    for (int w : {1, 2}) {
        run<0>("32 independent v_pk_fma_f32 (sgpr pair)", 32, w);
        run<1>("64 independent v_fma_f32", 64, w);
        run<2>("4 chains x 8 dependent v_pk_fma_f32", 32, w);
        run<3>("8 chains x 8 dependent v_fma_f32", 64, w);
        run<4>("32 independent v_fma_f64", 32, w);
    }
    return 0;
}
