// Do matrix-core and vector-ALU instructions of DIFFERENT waves on one SIMD overlap?  One workgroup of
// 8 waves per CU (two per SIMD); each wave runs either an MFMA loop or a VALU loop.
//   hipcc -O3 --offload-arch=gfx950 tools/coexec.hip -o tools/coexec && tools/coexec
// Output: time of "all MFMA", "all VALU", and mixed assignments; mixed ~= max(...) means they overlap,
// mixed ~= sum means they serialise.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// role_mask bit w: wave w runs the VALU loop (else the MFMA loop); bit 8+w: wave w idles
// VMODE 0: full-rate fma chain; 1: exp2 + rcp (quarter rate) mix like a sigmoid
// PRIO 1: the VALU waves raise their priority (s_setprio 3); 2: the MFMA waves do
template <int VMODE, int PRIO = 0>
__global__ __launch_bounds__(512, 1) void coexec_kernel(float *out, int mfma_iters, int valu_iters, unsigned role_mask,
                                                        float seed) {
    const int wave = threadIdx.x >> 6;
    if (role_mask >> (8 + wave) & 1) return;
    float s = 0.f;
    if (role_mask >> wave & 1) {
        if (PRIO == 1) __builtin_amdgcn_s_setprio(3);
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = seed + i * 0.01f + threadIdx.x * 1e-4f;
        for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (VMODE == 0) {
                    x[i] = fmaf(x[i], 0.999f, 0.001f);
                    x[i] = fmaf(x[i], 1.001f, -0.001f);
                    x[i] = fmaf(x[i], 0.999f, 0.001f);
                    x[i] = fmaf(x[i], 1.001f, -0.001f);
                } else {
                    const float e = __builtin_amdgcn_exp2f(x[i] * -1.442695f);
                    x[i] = fmaf(__builtin_amdgcn_rcpf(1.f + e), 0.5f, x[i] * 0.5f);
                }
            }
        }
        for (int i = 0; i < 8; ++i) s += x[i];
    } else {
        if (PRIO == 2) __builtin_amdgcn_s_setprio(3);
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        f16x8 a, b;
        for (int i = 0; i < 8; ++i) {
            a[i] = static_cast<_Float16>(seed + threadIdx.x * 1e-3f);
            b[i] = static_cast<_Float16>(seed * 0.5f + i);
        }
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 16; ++r) s += acc[i][r];
    }
    if (s == 12345.678f) out[0] = s;
}

// one instruction stream: each MFMA followed by NV independent full-rate VALU ops and NT transcendental ones
template <int NV, int NT = 0>
__global__ __launch_bounds__(256, 2) void interleave_kernel(float *out, int iters, float seed) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = static_cast<_Float16>(seed + threadIdx.x * 1e-3f);
        b[i] = static_cast<_Float16>(seed * 0.5f + i);
    }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = seed + i * 0.01f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) x[(i + v) & 7] = fmaf(x[(i + v) & 7], 0.999f, 0.001f);
#pragma unroll
            for (int v = 0; v < NT; ++v) x[(i + 4 + v) & 7] = __builtin_amdgcn_exp2f(x[(i + 4 + v) & 7]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) {
        s += x[i];
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    }
    if (s == 12345.678f) out[0] = s;
}

template <typename F>
float time_ms(F launch) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    float *d;
    CHECK(hipMalloc(&d, 4));
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int grid = p.multiProcessorCount;
    const int MI = 20000, VI = 20000;   // 160k MFMAs (x32 cycles) vs 640k VALU ops (x4 cycles) per wave
    struct Case { const char *name; unsigned mask; };
    const Case cases[] = {
        {"8 waves MFMA", 0x00},
        {"4 waves MFMA (w0-3), w4-7 idle", 0xf000},
        {"8 waves VALU", 0xff},
        {"4 waves VALU (w4-7), w0-3 idle", 0x0ff0},
        {"w0-3 MFMA + w4-7 VALU", 0xf0},
        {"even MFMA + odd VALU", 0xaa},
    };
    for (int vm = 0; vm < 2; ++vm) {
        printf("-- VALU loop: %s\n", vm == 0 ? "fma (full rate)" : "exp2+rcp+fma (sigmoid-like)");
        for (const Case &c : cases) {
            const unsigned mask = c.mask;
            const float ms = vm == 0
                ? time_ms([&] { hipLaunchKernelGGL(coexec_kernel<0>, dim3(grid), dim3(512), 0, 0, d, MI, VI, mask, 1.0f); })
                : time_ms([&] { hipLaunchKernelGGL(coexec_kernel<1>, dim3(grid), dim3(512), 0, 0, d, MI, VI, mask, 1.0f); });
            printf("%-36s %8.3f ms\n", c.name, ms);
        }
    }
    printf("-- priorities, sigmoid-like VALU loop, w0-3 MFMA + w4-7 VALU\n");
    printf("VALU waves prio 3   %8.3f ms\n", time_ms([&] { hipLaunchKernelGGL((coexec_kernel<1, 1>), dim3(grid), dim3(512), 0, 0, d, MI, VI, 0xf0u, 1.0f); }));
    printf("MFMA waves prio 3   %8.3f ms\n", time_ms([&] { hipLaunchKernelGGL((coexec_kernel<1, 2>), dim3(grid), dim3(512), 0, 0, d, MI, VI, 0xf0u, 1.0f); }));
    printf("-- priorities, fma VALU loop, w0-3 MFMA + w4-7 VALU\n");
    printf("VALU waves prio 3   %8.3f ms\n", time_ms([&] { hipLaunchKernelGGL((coexec_kernel<0, 1>), dim3(grid), dim3(512), 0, 0, d, MI, VI, 0xf0u, 1.0f); }));
    printf("MFMA waves prio 3   %8.3f ms\n", time_ms([&] { hipLaunchKernelGGL((coexec_kernel<0, 2>), dim3(grid), dim3(512), 0, 0, d, MI, VI, 0xf0u, 1.0f); }));
    printf("-- one stream, 2 waves per SIMD: MFMA + NV fma each\n");
    printf("NV=0  %8.3f ms\n", time_ms([&] { hipLaunchKernelGGL(interleave_kernel<0>, dim3(grid * 2), dim3(256), 0, 0, d, MI, 1.0f); }));
    printf("NV=2  %8.3f ms\n", time_ms([&] { hipLaunchKernelGGL(interleave_kernel<2>, dim3(grid * 2), dim3(256), 0, 0, d, MI, 1.0f); }));
    printf("NV=4  %8.3f ms\n", time_ms([&] { hipLaunchKernelGGL(interleave_kernel<4>, dim3(grid * 2), dim3(256), 0, 0, d, MI, 1.0f); }));
    printf("NV=8  %8.3f ms\n", time_ms([&] { hipLaunchKernelGGL(interleave_kernel<8>, dim3(grid * 2), dim3(256), 0, 0, d, MI, 1.0f); }));
    printf("NV=0 NT=1  %8.3f ms\n", time_ms([&] { hipLaunchKernelGGL((interleave_kernel<0, 1>), dim3(grid * 2), dim3(256), 0, 0, d, MI, 1.0f); }));
    printf("NV=0 NT=2  %8.3f ms\n", time_ms([&] { hipLaunchKernelGGL((interleave_kernel<0, 2>), dim3(grid * 2), dim3(256), 0, 0, d, MI, 1.0f); }));
    printf("NV=2 NT=1  %8.3f ms\n", time_ms([&] { hipLaunchKernelGGL((interleave_kernel<2, 1>), dim3(grid * 2), dim3(256), 0, 0, d, MI, 1.0f); }));
    printf("NV=0 NT=4  %8.3f ms\n", time_ms([&] { hipLaunchKernelGGL((interleave_kernel<0, 4>), dim3(grid * 2), dim3(256), 0, 0, d, MI, 1.0f); }));
    printf("-- one stream, 1 wave per SIMD\n");
    printf("NV=0       %8.3f ms\n", time_ms([&] { hipLaunchKernelGGL((interleave_kernel<0, 0>), dim3(grid), dim3(256), 0, 0, d, MI, 1.0f); }));
    printf("NV=4       %8.3f ms\n", time_ms([&] { hipLaunchKernelGGL((interleave_kernel<4, 0>), dim3(grid), dim3(256), 0, 0, d, MI, 1.0f); }));
    printf("NV=6       %8.3f ms\n", time_ms([&] { hipLaunchKernelGGL((interleave_kernel<6, 0>), dim3(grid), dim3(256), 0, 0, d, MI, 1.0f); }));
    printf("NV=0 NT=2  %8.3f ms\n", time_ms([&] { hipLaunchKernelGGL((interleave_kernel<0, 2>), dim3(grid), dim3(256), 0, 0, d, MI, 1.0f); }));
    return 0;
}
