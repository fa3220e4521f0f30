// What v_cvt_pkrtz_f16_f32 returns for values beyond the f16 range (decides whether split2_f16 needs its clamp).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float *x, unsigned *o, int n) {
    int i = threadIdx.x;
    if (i < n) { auto h = __builtin_amdgcn_cvt_pkrtz(x[i], -x[i]); o[i] = __builtin_bit_cast(unsigned, h); }
}
int main() {
    float h[8] = {1.0f, 65504.f, 65520.f, 70000.f, 1e6f, 3e38f, __builtin_inff(), 1e-8f};
    float *d; unsigned *o, r[8];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 8);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i) printf("%g -> lo16 0x%04x hi16 0x%04x\n", h[i], r[i] & 0xffff, r[i] >> 16);
    return 0;
}
