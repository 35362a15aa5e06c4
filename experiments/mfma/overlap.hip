// How many VALU instructions hide under a 16-cycle i8 MFMA? (experiments only)
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
template <int NV, int SAMEWAVE>
__global__ __launch_bounds__(256) void k(int* out, int iters) {
    v4i A = {(int)threadIdx.x, 1, 2, 3}, B = {4, 5, 6, (int)threadIdx.x};
    v4i C0 = {0,0,0,0}, C1 = C0, C2 = C0;
    int x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    const bool mf = SAMEWAVE || ((threadIdx.x >> 6) & 1) == 0;   // SAMEWAVE=0: even waves do MFMA, odd waves do VALU
    const bool va = SAMEWAVE || ((threadIdx.x >> 6) & 1) == 1;
    for (int i = 0; i < iters; i++) {
        if (mf) {
            C0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, C0, 0, 0, 0);
        }
        if (va) {
#pragma unroll
            for (int j = 0; j < NV; j++) {
                asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x0) : "v"(x1));
                asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x2) : "v"(x3));
            }
        }
        if (mf) {
            C1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, C1, 0, 0, 0);
            C2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, C2, 0, 0, 0);
        }
        if (va) {
#pragma unroll
            for (int j = 0; j < 2 * NV; j++) {
                asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x0) : "v"(x1));
                asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x2) : "v"(x3));
            }
        }
    }
    v4i S = C0 + C1 + C2;
    out[blockIdx.x * blockDim.x + threadIdx.x] = S[0] + S[1] + S[2] + S[3] + x0 + x2;
}
template <int NV, int SW> void go(int* out) {
    int iters = 20000, blocks = 256 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) { hipEventRecord(e0); k<NV, SW><<<blocks, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1); }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 waves (one per block... blocks=1024 of 256 threads -> 4 blocks/CU -> 4 waves/SIMD)
    double per_iter_ns = ms * 1e6 / iters;
    printf("samewave=%d VALU per MFMA=%d: %.3f ms, %.1f ns per iteration (3 MFMA + %d VALU per wave)\n", SW, 2 * NV, ms, per_iter_ns, 6 * NV);
}
int main() {
    int* out; hipMalloc(&out, 4 << 20);
    go<0,1>(out); go<1,1>(out); go<2,1>(out); go<3,1>(out); go<4,1>(out);
    go<1,0>(out); go<2,0>(out); go<3,0>(out); go<4,0>(out); go<6,0>(out);
    return 0;
}
