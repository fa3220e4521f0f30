// Issue rate of packed float32 instructions against plain ones on gfx950: the same number of FMAs per lane as
// 16 chains of v_fma_f32 or as 8 chains of v_pk_fma_f32 (also with a swapped operand and with an SGPR pair).
//   hipcc -O3 --offload-arch=gfx950 tools/pk_rate.hip -o tools/pk_rate && tools/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float c0, float c1) {
    f2 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f2{threadIdx.x * 1e-3f + i, 1.0f - i};
    const f2 m = f2{c0, c1}, a = f2{c1, c0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) {          // plain: two v_fma_f32
                    asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %3, %3, %4, %5"
                                 : "+v"(acc[i].x), "+v"(acc[i].y) : "v"(m.x), "v"(a.x), "v"(m.y), "v"(a.y));
                } else if (MODE == 1) {   // packed
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(m), "v"(a));
                } else if (MODE == 2) {   // packed, operand 0 swapped, operand 1 negated in its low half
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "+v"(acc[i]) : "v"(m), "v"(a));
                } else if (MODE == 3) {   // packed add
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(m));
                } else {                  // plain adds
                    asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %3" : "+v"(acc[i].x), "+v"(acc[i].y) : "v"(m.x), "v"(m.y));
                }
            }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, float *out, int waves_per_simd) {
    const int iters = 20000, grid = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 10, 0.999f, 1e-3f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 0.999f, 1e-3f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double lane_ops = double(grid) * 256 * iters * 32 * 2;     // float results per launch
    std::printf("%-44s %d waves/SIMD  %8.3f ms  %7.2f T results/s\n", name, waves_per_simd, ms, lane_ops / ms / 1e9);
}

int main() {
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32 x2", out, w);
        run<1>("v_pk_fma_f32", out, w);
        run<2>("v_pk_fma_f32 op_sel/neg", out, w);
        run<4>("v_add_f32 x2", out, w);
        run<3>("v_pk_add_f32", out, w);
    }
    return 0;
}
