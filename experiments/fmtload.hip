// Experiment: do MUBUF format loads (8_8_8_8 USCALED -> 3 floats) convert u8 -> f32 for free on
// gfx950, and what do they cost next to dwordx4 loads + v_cvt_f32_ubyteN?
//   hipcc --offload-arch=gfx950 -O3 -o fmtload experiments/fmtload.hip && ./fmtload
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f3 __attribute__((ext_vector_type(3)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

extern "C" __device__ f3 buf_load_fmt_xyz(i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v3f32");

__device__ __forceinline__ i4 make_rsrc(const void *p, unsigned bytes)
{
    // gfx9 V#: word0-1 base (stride 0), word2 num_records (bytes), word3: dst_sel xyzw = 4,5,6,7,
    // num_format USCALED (2) at [14:12], data_format 8_8_8_8 (10) at [18:15]
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    i4 r;
    r.x = static_cast<int>(a);
    r.y = static_cast<int>(a >> 32) & 0xffff;
    r.z = static_cast<int>(bytes);
    r.w = (4) | (5 << 3) | (6 << 6) | (7 << 9) | (2 << 12) | (10 << 15);
    return r;
}

constexpr int NPX = 20;

__global__ __launch_bounds__(256) void k_fmt(const unsigned char *src, float *out, int w, int h, int stride)
{
    const int x = (blockIdx.x * 256 + threadIdx.x) * 8, y = blockIdx.y;
    if (x + NPX >= w) return;
    const i4 rs = make_rsrc(src, static_cast<unsigned>(stride) * h);
    float acc = 0.f;
    const int base = y * stride + x * 4;
#pragma unroll
    for (int i = 0; i < NPX; i++) {
        const f3 v = buf_load_fmt_xyz(rs, base + 4 * i, 0, 0);
        acc += v.x * 1.0f + v.y * 2.0f + v.z * 3.0f;
    }
    out[static_cast<size_t>(y) * (w / 8) + x / 8] = acc;
}

__global__ __launch_bounds__(256) void k_cvt(const unsigned char *src, float *out, int w, int h, int stride)
{
    const int x = (blockIdx.x * 256 + threadIdx.x) * 8, y = blockIdx.y;
    if (x + NPX >= w) return;
    float acc = 0.f;
    const unsigned char *p = src + static_cast<size_t>(y) * stride + x * 4;
#pragma unroll
    for (int q = 0; q < NPX / 4; q++) {
        const u4 v = *reinterpret_cast<const u4 *>(p + 16 * q);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const unsigned px = v[e];
            acc += static_cast<float>(px & 0xffu) * 1.0f + static_cast<float>((px >> 8) & 0xffu) * 2.0f +
                   static_cast<float>((px >> 16) & 0xffu) * 3.0f;
        }
    }
    out[static_cast<size_t>(y) * (w / 8) + x / 8] = acc;
}

int main()
{
    const int w = 3840, h = 2160, stride = w * 4, n = 8;
    std::vector<unsigned char> host(static_cast<size_t>(stride) * h);
    for (size_t i = 0; i < host.size(); i++) host[i] = static_cast<unsigned char>((i * 2654435761u) >> 13);
    unsigned char *d;
    float *o1, *o2;
    hipMalloc(&d, host.size());
    hipMalloc(&o1, sizeof(float) * (w / 8) * h);
    hipMalloc(&o2, sizeof(float) * (w / 8) * h);
    hipMemcpy(d, host.data(), host.size(), hipMemcpyHostToDevice);
    hipMemset(o1, 0, sizeof(float) * (w / 8) * h);
    hipMemset(o2, 0, sizeof(float) * (w / 8) * h);
    dim3 grid((w / 8 + 255) / 256, h);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int variant = 0; variant < 2; variant++) {
        auto launch = [&]() {
            if (variant) hipLaunchKernelGGL(k_cvt, grid, dim3(256), 0, 0, d, o2, w, h, stride);
            else hipLaunchKernelGGL(k_fmt, grid, dim3(256), 0, 0, d, o1, w, h, stride);
        };
        for (int rep = 0; rep < 200; rep++) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int rep = 0; rep < n * 25; rep++) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.2f us per launch\n", variant ? "dwordx4 + cvt" : "format xyz   ", ms * 1000 / (n * 25));
    }
    std::vector<float> a((w / 8) * h), b((w / 8) * h);
    hipMemcpy(a.data(), o1, a.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), o2, b.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < a.size(); i++) bad += a[i] != b[i];
    printf("mismatches: %zu of %zu (sample %.1f vs %.1f)\n", bad, a.size(), a[1000], b[1000]);
    return 0;
}
