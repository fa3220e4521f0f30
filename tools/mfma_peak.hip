// Measures what the matrix cores of the GPU in this box sustain (register-resident operands, no memory
// traffic): the practical ceiling the conv / GEMM kernels are compared with.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak && tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256, 2) void peak_kernel(float *out, int iters, float seed) {
    f32x16 acc[8];
    f32x4 acc4[8];
    for (int i = 0; i < 8; ++i) {
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f;
    }
    // MODE 3: as MODE 0 but with operands whose bits differ from one MFMA to the next (four random
    // register sets), which is what real data looks like to the datapath's power draw
    f16x8 a, b, ar[4], br[4];
    for (int i = 0; i < 8; ++i) { a[i] = static_cast<_Float16>(seed + threadIdx.x * 1e-3f); b[i] = static_cast<_Float16>(seed * 0.5f + i); }
    unsigned h = (threadIdx.x + 1) * 2654435761u + blockIdx.x * 40503u;
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 8; ++i) {
            h = h * 1664525u + 1013904223u;
            ar[k][i] = static_cast<_Float16>((static_cast<int>(h >> 16 & 0xfff) - 2048) * (1.0f / 1024));
            h = h * 1664525u + 1013904223u;
            br[k][i] = static_cast<_Float16>((static_cast<int>(h >> 16 & 0xfff) - 2048) * (1.0f / 4096));
        }
    const float af = seed + threadIdx.x, bf = seed * 2;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
            if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[i], 0, 0, 0);
            if (MODE == 2) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc4[i], 0, 0, 0);
            if (MODE == 3) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar[i & 3], br[(i + (i >> 2)) & 3], acc[i], 0, 0, 0);
            if (MODE == 4) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ar[i & 3], br[(i + (i >> 2)) & 3], acc4[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) {
        for (int r = 0; r < 16; ++r) s += acc[i][r];
        for (int r = 0; r < 4; ++r) s += acc4[i][r];
    }
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
void run(const char *name, double flop_per_mfma, float *d, int grid, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(peak_kernel<MODE>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double flops = flop_per_mfma * 8.0 * iters * 4.0 * grid;
        printf("%-28s grid %5d iters %6d  %8.3f ms  %8.1f TFLOP/s\n", name, grid, iters, ms, flops / (ms * 1e-3) / 1e12);
    }
}

int main() {
    float *d;
    CHECK(hipMalloc(&d, 4));
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    printf("%s, %d CUs, clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
    const int grid = p.multiProcessorCount * 2;
    for (int iters : {20000, 200000}) {
        run<0>("v_mfma_f32_32x32x16_f16", 2.0 * 32 * 32 * 16, d, grid, iters);
        run<2>("v_mfma_f32_16x16x32_f16", 2.0 * 16 * 16 * 32, d, grid, iters);
    }
    run<3>("32x32x16_f16 random operands", 2.0 * 32 * 32 * 16, d, grid, 200000);
    run<4>("16x16x32_f16 random operands", 2.0 * 16 * 16 * 32, d, grid, 400000);
    run<1>("v_mfma_f32_32x32x2_f32", 2.0 * 32 * 32 * 2, d, grid, 20000);
    return 0;
}
