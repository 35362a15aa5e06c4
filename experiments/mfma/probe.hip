// Probe: operand layout and issue rate of the gfx950 i8 MFMAs (experiments only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

// out[m][n] = sum_k A[m][k]*B[k][n]; A given as bytes a[m*64+k], B as b[n*64+k] (B^T rows)
__global__ void layout64(const int8_t* a, const int8_t* b, int* c) {
    int l = threadIdx.x, r = l & 15, g = l >> 4;
    v4i A = *(const v4i*)(a + r * 64 + g * 16);
    v4i B = *(const v4i*)(b + r * 64 + g * 16);
    v4i C = {0, 0, 0, 0};
    C = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, C, 0, 0, 0);
    for (int i = 0; i < 4; i++) c[(4 * g + i) * 16 + r] = C[i];
}
__global__ void layout32(const int8_t* a, const int8_t* b, int* c) {
    int l = threadIdx.x, r = l & 15, g = l >> 4;
    long A = *(const long*)(a + r * 32 + g * 8);
    long B = *(const long*)(b + r * 32 + g * 8);
    v4i C = {0, 0, 0, 0};
    C = __builtin_amdgcn_mfma_i32_16x16x32_i8(A, B, C, 0, 0, 0);
    for (int i = 0; i < 4; i++) c[(4 * g + i) * 16 + r] = C[i];
}
template <int K64>
__global__ void rate(int* out, int iters) {
    v4i A = {(int)threadIdx.x, 1, 2, 3}, B = {4, 5, 6, (int)threadIdx.x};
    v4i C0 = {0,0,0,0}, C1 = C0, C2 = C0, C3 = C0, C4 = C0, C5 = C0;
    for (int i = 0; i < iters; i++) {
        if (K64) {
            C0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, C0, 0, 0, 0);
            C1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, C1, 0, 0, 0);
            C2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, C2, 0, 0, 0);
            C3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, C3, 0, 0, 0);
            C4 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, C4, 0, 0, 0);
            C5 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, C5, 0, 0, 0);
        } else {
            long a = ((long)A[0] << 32) | (unsigned)A[1], b = ((long)B[0] << 32) | (unsigned)B[3];
            C0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, C0, 0, 0, 0);
            C1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, C1, 0, 0, 0);
            C2 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, C2, 0, 0, 0);
            C3 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, C3, 0, 0, 0);
            C4 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, C4, 0, 0, 0);
            C5 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, C5, 0, 0, 0);
        }
    }
    v4i S = C0 + C1 + C2 + C3 + C4 + C5;
    out[blockIdx.x * blockDim.x + threadIdx.x] = S[0] + S[1] + S[2] + S[3];
}
int main() {
    for (int K : {64, 32}) {
        std::vector<int8_t> a(16 * K), b(16 * K);
        for (int i = 0; i < 16 * K; i++) { a[i] = (int8_t)((i * 37 + 11) % 251 - 125); b[i] = (int8_t)((i * 53 + 7) % 241 - 120); }
        int8_t *da, *db; int* dc;
        hipMalloc(&da, 16 * K); hipMalloc(&db, 16 * K); hipMalloc(&dc, 1024);
        hipMemcpy(da, a.data(), 16 * K, hipMemcpyHostToDevice);
        hipMemcpy(db, b.data(), 16 * K, hipMemcpyHostToDevice);
        if (K == 64) layout64<<<1, 64>>>(da, db, dc); else layout32<<<1, 64>>>(da, db, dc);
        int c[256]; hipMemcpy(c, dc, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int m = 0; m < 16; m++) for (int n = 0; n < 16; n++) {
            int s = 0; for (int k = 0; k < K; k++) s += (int)a[m * K + k] * (int)b[n * K + k];
            if (s != c[m * 16 + n]) bad++;
        }
        printf("K=%d layout mismatches: %d of 256\n", K, bad);
    }
    int* out; hipMalloc(&out, 4 * 256 * 1024 * 4);
    for (int K64 : {1, 0}) for (int wpb : {64, 256, 512}) {
        int iters = 20000, blocks = 256 * 4;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (K64) rate<1><<<blocks, wpb>>>(out, iters); else rate<0><<<blocks, wpb>>>(out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double n = (double)blocks * (wpb / 64) * iters * 6;
        double macs = n * 16 * 16 * (K64 ? 64 : 32);
        printf("K=%d threads/block=%d: %.3f ms, %.1f TOPS, %.2f ns per MFMA per SIMD-slot\n", K64 ? 64 : 32, wpb, ms,
               2 * macs / ms / 1e9, ms * 1e6 / (n / 1024.0));
    }
    return 0;
}
