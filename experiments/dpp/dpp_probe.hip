// Which lane does a DPP wave_shl:1 / row_shl:n / wave_rol:1 read on gfx950?  Prints, per control, out[lane] = source lane (or -1: kept old).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__global__ void k(int *out)
{
    const int lane = threadIdx.x;
    out[lane] = __builtin_amdgcn_update_dpp(-1, lane, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
void run(const char *name)
{
    int *d, h[64];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k<CTRL>, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-12s", name);
    for (int i = 0; i < 64; i++) printf(" %d", h[i]);
    printf("\n");
    hipFree(d);
}
int main()
{
    run<0x130>("wave_shl:1");
    run<0x134>("wave_rol:1");
    run<0x138>("wave_shr:1");
    run<0x101>("row_shl:1");
    run<0x104>("row_shl:4");
    run<0x111>("row_shr:1");
    run<0x142>("row_bcast15");
    run<0x143>("row_bcast31");
    return 0;
}
