// Probe: rounding / saturation of v_cvt_pk_u8_f32 and the lane movement of the wave_shr / wave_shl DPP controls on gfx950.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_cvt.hip -o /tmp/ubench_cvt && /tmp/ubench_cvt
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float *in, unsigned *out, int n, unsigned *dpp)
{
    int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0u);
    dpp[i] = __builtin_amdgcn_update_dpp(1000, i, 0x138, 0xf, 0xf, false);        // wave_shr:1
    dpp[64 + i] = __builtin_amdgcn_update_dpp(1000, i, 0x130, 0xf, 0xf, false);   // wave_shl:1
}
int main()
{
    const float h[] = {0.0f, 0.49f, 0.5f, 0.51f, 0.999f, 1.5f, 2.5f, 3.5f, 127.999f, 254.5f, 254.999f, 255.0f, 255.5f, 300.0f, -0.5f, -1.0f, 1e9f, -1e9f};
    const int n = sizeof(h) / sizeof(h[0]);
    float *d; unsigned *o, *p;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 64 * 4); hipMalloc(&p, 128 * 4);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n, p);
    unsigned ho[64], hp[128];
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    hipMemcpy(hp, p, sizeof(hp), hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("cvt_pk_u8_f32(%g) = %u\n", h[i], ho[i] & 255u);
    printf("wave_shr:1 lanes 0,1,2,31,32,33,63 <- %u %u %u %u %u %u %u\n", hp[0], hp[1], hp[2], hp[31], hp[32], hp[33], hp[63]);
    printf("wave_shl:1 lanes 0,1,30,31,32,62,63 <- %u %u %u %u %u %u %u\n", hp[64], hp[65], hp[94], hp[95], hp[96], hp[126], hp[127]);
    return 0;
}
