// Experiment: issue cost of VALU instruction classes on gfx950 (per wave64 instruction, per SIMD),
// and the dependent-issue latency of v_pk_fma_f32 / v_fma_f32 (UNROLL independent chains per wave).
//   hipcc --offload-arch=gfx950 -O3 -o valurate experiments/valurate.hip && ./valurate
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096;

template <int KIND, int UNROLL>
__global__ __launch_bounds__(256) void k(double *out, double seed, unsigned useed)
{
    double a[UNROLL];
    float f[UNROLL];
    v2f p[UNROLL];
    unsigned u[UNROLL];
    for (int i = 0; i < UNROLL; i++) {
        a[i] = seed + i + threadIdx.x; f[i] = static_cast<float>(a[i]); p[i] = (v2f){f[i], f[i] + 1};
        u[i] = useed * (i + 1) + threadIdx.x;
    }
    const double m = 1.0000001, c = 0.25;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            if (KIND == 0) a[i] = __builtin_fma(a[i], m, c);
            if (KIND == 1) a[i] = a[i] + c;
            if (KIND == 2) a[i] = a[i] * m;
            if (KIND == 3) f[i] = __builtin_fmaf(f[i], 1.0000001f, 0.25f);
            if (KIND == 4) p[i] = __builtin_elementwise_fma(p[i], (v2f){1.0000001f, 1.0000001f}, (v2f){0.25f, 0.25f});
            if (KIND == 6) { float t; asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(t) : "v"(u[i])); u[i] = __float_as_uint(t) + 3u; }   // cvt + add
            if (KIND == 7) u[i] = __builtin_amdgcn_perm(u[i], useed, 0x0c010c00u) + 1u;                                                  // perm + add
            if (KIND == 8) u[i] = __builtin_amdgcn_cvt_pk_u8_f32(__uint_as_float(u[i] | 0x40000000u), 1, u[i]);                           // or + cvt_pk_u8
            if (KIND == 9) u[i] = (u[i] << 3) + useed;                                                                                    // v_lshl_add_u32
            if (KIND == 10) u[i] = __builtin_amdgcn_udot4(u[i], useed, u[i], false);                                                      // v_dot4_u32_u8
            if (KIND == 11) u[i] = __builtin_amdgcn_perm(u[i], useed, 0x0c010c00u);                                                       // v_perm_b32 alone
            if (KIND == 12) { float t; asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(t) : "v"(u[i])); u[i] = __float_as_uint(t); }      // cvt alone
            if (KIND == 13) { unsigned t; asm volatile("v_and_b32 %0, 0x00ff00ff, %1" : "=v"(t) : "v"(u[i])); u[i] = t; }
            if (KIND == 14) { unsigned t; asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(t) : "v"(u[i])); u[i] = t; }
            if (KIND == 15) { unsigned t; asm volatile("v_lshrrev_b32 %0, 8, %1" : "=v"(t) : "v"(u[i])); u[i] = t; }
            if (KIND == 16) { unsigned t; asm volatile("v_and_or_b32 %0, %1, %3, %2" : "=v"(t) : "v"(u[i]), "v"(useed), "s"(0xffu)); u[i] = t; }
            if (KIND == 17) { unsigned t; asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %2" : "=v"(t) : "v"(u[i]), "v"(useed)); u[i] = t; }
            if (KIND == 18) { float t; asm volatile("v_add_f32 %0, 1.0, %1" : "=v"(t) : "v"(f[i])); f[i] = t; }
            if (KIND == 19) { float t; asm volatile("v_floor_f32 %0, %1" : "=v"(t) : "v"(f[i])); f[i] = t; }
            if (KIND == 20) { float t; asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(t) : "v"(u[i])); u[i] = __float_as_uint(t); }
            if (KIND == 21) { float t; asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(t) : "v"(u[i])); u[i] = __float_as_uint(t); }
            if (KIND == 22) { float t; asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(t) : "v"(u[i])); u[i] = __float_as_uint(t); }
            if (KIND == 23) { unsigned t; asm volatile("v_mad_u32_u24 %0, %1, %2, %1" : "=v"(t) : "v"(u[i]), "v"(useed)); u[i] = t; }
            if (KIND == 24) { unsigned t; asm volatile("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(t) : "v"(u[i]), "v"(useed)); u[i] = t; }
            if (KIND == 25) { float t = f[i]; asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(t) : "v"(f[i]), "v"(1.0000001f)); f[i] = t; }
            if (KIND == 26) { float t; asm volatile("v_cvt_f32_ubyte3 %0, %1" : "=v"(t) : "v"(u[i])); u[i] = __float_as_uint(t); }
            if (KIND == 27) { unsigned t; asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(t) : "v"(u[i]), "v"(useed)); u[i] = t; }
            if (KIND == 28) { float t; asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(t) : "v"(u[i])); u[i] = __float_as_uint(t); }
            if (KIND == 29) { unsigned t; asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(t) : "v"(u[i]), "v"(useed)); u[i] = t; }
            if (KIND == 30) { unsigned t; asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(t) : "v"(u[i]), "v"(useed)); u[i] = t; }
            if (KIND == 31) { unsigned t; asm volatile("v_mad_u32_u16 %0, %1, %2, %1" : "=v"(t) : "v"(u[i]), "v"(useed)); u[i] = t; }
            if (KIND == 32) { unsigned t; asm volatile("v_pk_mad_u16 %0, %1, %2, %1" : "=v"(t) : "v"(u[i]), "v"(useed)); u[i] = t; }
            if (KIND == 33) { unsigned t; asm volatile("v_dot2_u32_u16 %0, %1, %2, %1" : "=v"(t) : "v"(u[i]), "v"(useed)); u[i] = t; }
        }
    }
    double s = 0;
    for (int i = 0; i < UNROLL; i++) s += a[i] + f[i] + p[i].x + p[i].y + u[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int UNROLL>
static void run(const char *name, int instr_per_elem)
{
    double *d;
    (void)hipMalloc(&d, sizeof(double) * 256 * 4096);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 40; rep++) hipLaunchKernelGGL((k<KIND, UNROLL>), dim3(4096), dim3(256), 0, 0, d, 1.0, 77u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int rep = 0; rep < 10; rep++) hipLaunchKernelGGL((k<KIND, UNROLL>), dim3(4096), dim3(256), 0, 0, d, 1.0, 77u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = 10.0 * 4096 * 4 * double(ITERS) * UNROLL * instr_per_elem;   // 4 waves per block
    const double per_simd_per_s = wave_instr / (ms * 1e-3) / (256.0 * 4);
    printf("%-34s x%-2d %8.3f ms  %.3f G wave-instr/s/SIMD -> %.2f clk each at 2.4 GHz\n", name, UNROLL, ms / 10,
           per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
    (void)hipFree(d);
}

int main()
{
    run<3, 16>("v_fma_f32", 1);
    run<4, 16>("v_pk_fma_f32", 1);
    run<0, 16>("v_fma_f64", 1);
    run<1, 16>("v_add_f64", 1);
    run<2, 16>("v_mul_f64", 1);
    run<6, 16>("v_cvt_f32_ubyte1 + v_add_u32", 2);
    run<7, 16>("v_perm_b32 + v_add_u32", 2);
    run<8, 16>("v_or_b32 + v_cvt_pk_u8_f32", 2);
    run<9, 16>("v_lshl_add_u32", 1);
    run<10, 16>("v_dot4_u32_u8", 1);
    run<11, 16>("v_perm_b32", 1);
    run<12, 16>("v_cvt_f32_ubyte1", 1);
    run<13, 16>("v_and_b32 (literal)", 1);
    run<14, 16>("v_bfe_u32", 1);
    run<15, 16>("v_lshrrev_b32", 1);
    run<16, 16>("v_and_or_b32", 1);
    run<17, 16>("v_cvt_pk_u8_f32", 1);
    run<18, 16>("v_add_f32", 1);
    run<19, 16>("v_floor_f32", 1);
    run<20, 16>("v_cvt_f32_u32", 1);
    run<21, 16>("v_cvt_f32_u32_sdwa BYTE_1", 1);
    run<22, 16>("v_cvt_f32_ubyte0", 1);
    run<26, 16>("v_cvt_f32_ubyte3", 1);
    run<28, 16>("v_cvt_f32_i32", 1);
    run<23, 16>("v_mad_u32_u24", 1);
    run<24, 16>("v_mul_u32_u24_sdwa BYTE_1", 1);
    run<29, 16>("v_mul_u32_u24", 1);
    run<30, 16>("v_mul_lo_u32", 1);
    run<25, 16>("v_fmac_f32", 1);
    run<27, 16>("v_add_u32_sdwa BYTE_2", 1);
    run<31, 16>("v_mad_u32_u16", 1);
    run<32, 16>("v_pk_mad_u16", 1);
    run<33, 16>("v_dot2_u32_u16", 1);
    // dependent chains: 256-thread blocks, 4096 blocks -> 16 waves per SIMD resident; x1 = every
    // instruction of a wave depends on the previous one
    run<3, 1>("v_fma_f32 dependent", 1);
    run<3, 2>("v_fma_f32 dependent", 1);
    run<4, 1>("v_pk_fma_f32 dependent", 1);
    run<4, 2>("v_pk_fma_f32 dependent", 1);
    run<4, 4>("v_pk_fma_f32 dependent", 1);
    return 0;
}
