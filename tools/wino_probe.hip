// Would a Winograd F(2x2, 3x3) main loop beat the direct 9-tap main loop of conv_split_kernel<64,128,16>?
//
// Both kernels below run ONLY the matrix-core main loop of one 256-position x 128-channel output tile of CRNN
// block 3 (64 -> 128 channels) in split-f16 arithmetic (three f16 MFMAs per float32-grade product), with the
// operand traffic the respective algorithm needs and nothing else (no staging, no BatchNorm / GLU epilogue, no
// stores): A fragments (activations, hi / lo planes) come from LDS with ds_read_b128, B fragments (weights,
// hi / lo planes) stream from an L2-resident array one step ahead of use, accumulators live in registers.
//
//   direct    4 waves as 2 x 2, each 128 positions x 64 channels = 4 x 2 MFMA blocks of 32 x 32 (128
//             accumulator registers).  36 steps (9 taps x 4 k-steps of 16 channels); a step reads 4 A blocks x
//             (hi, lo) from LDS and 2 B blocks x (hi, lo) from L2 and issues 24 MFMAs.  864 MFMAs per wave.
//   winograd  the same output tile as 64 Winograd tiles of 2 x 2 outputs.  The 16 transform components are 16
//             independent GEMMs [64 tiles] x [64 channels] x [128 channels]; their results must all meet in the
//             inverse transform, so a wave owns every component of its outputs: 4 waves as 1 x 4, each 64 tiles
//             (2 MFMA blocks) x 32 channels; one component at a time is accumulated over the whole K (32
//             registers) and folded into the four outputs of its 2 x 2 tile (128 registers) with the +-1
//             coefficients of A^T . A.  16 components x 4 k-steps; a step reads 2 A blocks x (hi, lo) from LDS
//             and 1 B block x (hi, lo) from L2 and issues 6 MFMAs.  384 MFMAs per wave (2.25x fewer) and
//             48 x 32 register multiply-adds for the inverse transform.
//
// The Winograd form needs 2.7x the operand bytes per MFMA (a component's accumulators cannot share operands
// with another component's, and 16 of them do not fit in registers next to a 4 x 2 block), all of K of a
// component resident in LDS before its GEMM starts, and 4x the conversions in front of it (16 transformed
// values per 4 outputs and channel instead of ~4.5) -- the latter two are NOT in this probe, which gives
// Winograd its best case.
//
//   hipcc -O3 --offload-arch=gfx950 tools/wino_probe.hip -o tools/wino_probe && tools/wino_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int LDS_BYTES = 64 * 1024;

__device__ __forceinline__ f16x8 lds_frag(const char *smem, int byte_off) {
    return *reinterpret_cast<const f16x8 *>(smem + byte_off);
}

__device__ __forceinline__ void fill_lds(char *smem, unsigned seed) {
    unsigned h = (threadIdx.x + 1) * 2654435761u + seed;
    unsigned short *p = reinterpret_cast<unsigned short *>(smem);
    for (int i = threadIdx.x; i < LDS_BYTES / 2; i += 256) {
        h = h * 1664525u + 1013904223u;
        const _Float16 v = static_cast<_Float16>((static_cast<int>(h >> 16 & 0xfff) - 2048) * (1.0f / 1024));
        p[i] = __builtin_bit_cast(unsigned short, v);
    }
    __syncthreads();
}

// ---- direct: 36 steps x (8 LDS reads + 4 L2 loads + 24 MFMAs) per tile
__global__ __launch_bounds__(256, 2) void direct_kernel(const uint4 *__restrict__ w, float *out, int tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    fill_lds(smem, blockIdx.x);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // weights [step 36][plane 2][n 128][16 f16]: a lane's B fragment = 8 f16 of column n, k half lane >> 5
    const uint4 *wl = w + (64 * wn + (lane & 31)) * 2 + (lane >> 5);
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float sink = 0.f;
    for (int t = 0; t < tiles; ++t) {
        uint4 bn[2][2];
        for (int pl = 0; pl < 2; ++pl)
            for (int j = 0; j < 2; ++j) bn[pl][j] = wl[(0 * 2 + pl) * 256 + 64 * j];
#pragma unroll 4
        for (int s = 0; s < 36; ++s) {
            uint4 b[2][2];
            for (int pl = 0; pl < 2; ++pl)
                for (int j = 0; j < 2; ++j) b[pl][j] = bn[pl][j];
            const int sn = s + 1 < 36 ? s + 1 : 0;
            for (int pl = 0; pl < 2; ++pl)
                for (int j = 0; j < 2; ++j) bn[pl][j] = wl[(sn * 2 + pl) * 256 + 64 * j];
            __builtin_amdgcn_sched_barrier(0);
            // A: [plane][position 256 + halo ...][16 ch + 16 B of padding] as the conv kernel lays a chunk out (48 B
            // per position: ds_read_b128 of 32 consecutive positions is conflict free); the tap moves the base
            const int base = ((s % 9) * 48 * 11 + (s / 9) * 48 + wm * 128 * 48) & (LDS_BYTES / 2 - 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int off = (base + (32 * i + (lane & 31)) * 48 + (lane >> 5) * 16) % (LDS_BYTES / 2 - 16) & ~15;
                const f16x8 ah = lds_frag(smem, off), al = lds_frag(smem, off + LDS_BYTES / 2);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f16x8 bh = __builtin_bit_cast(f16x8, b[0][j]), bl = __builtin_bit_cast(f16x8, b[1][j]);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[i][j], 0, 0, 0);
                }
            }
        }
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 2; ++j)
                for (int r = 0; r < 16; ++r) { sink += acc[i][j][r]; acc[i][j][r] = 0.f; }
    }
    if (sink == 12345.678f) out[0] = sink;
}

// ---- winograd: 16 components x 4 k-steps x (4 LDS reads + 2 L2 loads + 6 MFMAs), then the inverse transform
template <bool FOLD>
__global__ __launch_bounds__(256, 2) void wino_kernel(const uint4 *__restrict__ w, float *out, int tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    fill_lds(smem, blockIdx.x);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // weights [component 16][k-step 4][plane 2][n 128][16 f16]
    const uint4 *wl = w + (32 * wave + (lane & 31)) * 2 + (lane >> 5);
    f32x16 y[2][2][2];                         // [m block][output row][output column] of the 2 x 2 tile
    for (int i = 0; i < 2; ++i)
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b)
                for (int r = 0; r < 16; ++r) y[i][a][b][r] = 0.f;
    float sink = 0.f;
    for (int t = 0; t < tiles; ++t) {
        uint4 bn[2];
        for (int pl = 0; pl < 2; ++pl) bn[pl] = wl[(0 * 2 + pl) * 256];
#pragma unroll 1
        for (int cr = 0; cr < 4; ++cr) {
        // row coefficients of A^T = [[1, 1, 1, 0], [0, 1, -1, -1]] for component row cr, as run-time floats (the
        // component rows are a run-time loop: unrolling all 16 components lets the scheduler hoist their loads
        // and spills ~450 registers)
        const float ar[2] = {cr < 3 ? 1.f : 0.f, cr == 0 ? 0.f : cr == 1 ? 1.f : -1.f};
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) {
            const int c = cr * 4 + cq;
            f32x16 m[2];
            for (int i = 0; i < 2; ++i)
                for (int r = 0; r < 16; ++r) m[i][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                uint4 b[2];
                for (int pl = 0; pl < 2; ++pl) b[pl] = bn[pl];
                const int sn = (c * 4 + ks + 1) & 63;
                for (int pl = 0; pl < 2; ++pl) bn[pl] = wl[(sn * 2 + pl) * 256];
                __builtin_amdgcn_sched_barrier(0);
                // A: [component][k-step][plane][tile 64][16 ch + pad] = 3 KB per (component, k-step, plane): 384 KB for a
                // whole tile, of which the probe keeps 64 KB and wraps (a real kernel would have to stage all of it)
                const int base = ((c * 4 + ks) * 3072) % (LDS_BYTES / 2 - 3072);
                const f16x8 bh = __builtin_bit_cast(f16x8, b[0]), bl = __builtin_bit_cast(f16x8, b[1]);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int off = base + (32 * i + (lane & 31)) * 48 + (lane >> 5) * 16;
                    const f16x8 ah = lds_frag(smem, off), al = lds_frag(smem, off + LDS_BYTES / 2);
                    m[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, m[i], 0, 0, 0);
                    m[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, m[i], 0, 0, 0);
                    m[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, m[i], 0, 0, 0);
                }
            }
            if (FOLD) {
                // Y = A^T M A with A^T = [[1, 1, 1, 0], [0, 1, -1, -1]]: component (r, q) reaches output row a with
                // coefficient AT[a][r] and output column b with AT[b][q]
                constexpr int AT[2][4] = {{1, 1, 1, 0}, {0, 1, -1, -1}};
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b2 = 0; b2 < 2; ++b2) {
                        if (AT[b2][cq] == 0) continue;
                        const float coef = ar[a] * static_cast<float>(AT[b2][cq]);
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int r = 0; r < 16; ++r) y[i][a][b2][r] = fmaf(coef, m[i][r], y[i][a][b2][r]);
                    }
            } else {
                for (int i = 0; i < 2; ++i) y[i][cq & 1][(cq >> 1) & 1] += m[i];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        for (int i = 0; i < 2; ++i)
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b)
                    for (int r = 0; r < 16; ++r) { sink += y[i][a][b][r]; y[i][a][b][r] = 0.f; }
    }
    if (sink == 12345.678f) out[0] = sink;
}

template <typename K>
double time_kernel(K kern, const uint4 *w, float *out, int grid, int tiles) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS_BYTES, 0, w, out, tiles);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    CHECK(hipGetLastError());
    return best;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    // weights: direct 36 x 2 x 128 x 32 B = 288 KB, winograd 64 x 2 x 128 x 32 B = 512 KB (both L2 resident)
    std::vector<unsigned short> hw(64 * 2 * 128 * 16);
    unsigned h = 12345;
    for (auto &v : hw) {
        h = h * 1664525u + 1013904223u;
        const _Float16 f = static_cast<_Float16>((static_cast<int>(h >> 16 & 0xfff) - 2048) * (1.0f / 4096));
        v = __builtin_bit_cast(unsigned short, f);
    }
    uint4 *dw;
    float *dout;
    CHECK(hipMalloc(&dw, hw.size() * 2));
    CHECK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dout, 4));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(direct_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(wino_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(wino_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    // block 3 at 512 clips: 250 x 16 positions per clip / 256 per tile (the conv kernel's tiles: 16 rows x 16
    // columns, 16 tiles per clip incl. the partly empty last one) = 8192 tiles, spread over 2 workgroups per CU
    const int grid = 2 * cus, total_tiles = 8192, tiles = (total_tiles + grid - 1) / grid;
    const double n_tiles = static_cast<double>(tiles) * grid;
    const double d = time_kernel(direct_kernel, dw, dout, grid, tiles);
    const double wf = time_kernel(wino_kernel<true>, dw, dout, grid, tiles);
    const double wn = time_kernel(wino_kernel<false>, dw, dout, grid, tiles);
    const double flop_d = n_tiles * 4 * 864 * 32768.0, flop_w = n_tiles * 4 * 384 * 32768.0;
    printf("%d CUs, %d workgroups x %d tiles of 256 positions x 128 channels (block 3, 512 clips: 8192 tiles)\n", cus, grid, tiles);
    printf("direct   main loop: %.3f ms  (864 MFMAs per wave and tile, %.0f TFLOP/s of f16 MFMA issue)\n", d, flop_d / d * 1e-9);
    printf("winograd main loop: %.3f ms  with the inverse transform (384 MFMAs + 1536 multiply-adds per wave and tile, %.0f TFLOP/s of issue)\n", wf, flop_w / wf * 1e-9);
    printf("winograd main loop: %.3f ms  without it (one addition per component instead of 1, 2 or 4)\n", wn);
    printf("ratio winograd / direct: %.2f (2.25x fewer MFMAs would be 0.44)\n", wf / d);
    return 0;
}
