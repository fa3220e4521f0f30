// Baseline CRNN eval-mode forward for gfx950 (MI355X).
//
// Stands in for CRNN.forward (nnet/CRNN.py:191-268) in the one configuration
// the reference builds (training.py:207-212):
//   7 x [Conv3x3(pad 1) -> BatchNorm2d(eps 1e-3, running stats) -> GLU
//        (Linear_{C->C}(x) * sigmoid(x), nnet/CNN.py:5-16) -> AvgPool]
//   -> squeeze / permute -> 2-layer bidirectional GRU(128) (nnet/RNN.py:19-30)
//   -> sigmoid(Linear(256 -> n_class))                     (nnet/CRNN.py:142-143)
//
// Data layout in HBM: activations are channels-last, [clip][time][freq][chan]
// float32, so the contraction index of every implicit GEMM is contiguous and
// the last block's output [clip][time][1][128] already is the [B, T', C]
// sequence the GRU reads (no permute kernel).
//
// Arithmetic: by default ("split-f16") every float32 operand of a contraction is split into two f16
// values and each product formed from three f16 MFMAs with float32 accumulation (conv_split.h), which
// keeps float32-grade results at several times the f32 MFMA rate.  sedmi_crnn_set_precision selects the
// exact-f32 twins instead (v_mfma_f32_32x32x2_f32 is bit-for-bit an fmaf chain, so the only differences
// to the reference's CPU result come from summation order).
//
// Kernels (split-f16 path / exact-f32 twin)
//   conv0_split_kernel / conv0_kernel   block 0 (1 input channel), GLU linear folded into the conv
//                     weights; on the matrix cores as a 32 x 9 matrix times 3x3 patches / on the VALU,
//                     one lane = one pooled output position x 16 channels.
//   conv_split_kernel (conv_split.h) / conv_mfma_kernel   blocks 1..6: implicit GEMM
//                     [positions] x [9*Cin] x [Cout].  A workgroup stages its input rows (+halo) once
//                     into LDS, streams packed weights from L2 straight into registers (no LDS round
//                     trip, no barrier in the main loop), then runs the GLU's C x C linear as a second
//                     MFMA GEMM on the BatchNorm output kept in LDS, multiplies by sigmoid(z) from
//                     registers and average-pools pairs that sit in adjacent accumulator registers of
//                     the same lane.
//   linear_split_kernel (conv_split.h) / linear_kernel   GRU input projections for all timesteps at once.
//   gru_split_kernel / gru_kernel   persistent recurrence: one workgroup = 32 clips x one direction for
//                     all T' steps; W_hh lives in registers as MFMA B fragments for the whole kernel, h in
//                     LDS (double buffered, one barrier per step), gates are lane-local because each wave
//                     owns the r/z/n columns of the same hidden units.
//   head_small_kernel (<= 32 classes, f32 MFMA) / head_kernel   Linear(256 -> n_class) + sigmoid.
#include "common.h"

#include <algorithm>
#include <cmath>
#include <string>
#include <type_traits>

namespace {

#include "mfma_common.h"
#include "conv_split.h"
// wave tiling of the 128-channel blocks: CONV_WM x (4 / CONV_WM) waves, CONV_MT m-tiles per wave.
// 2 x 2 waves with 4 x 2 tiles measured best; 1 x 4 waves with 8 x 1 tiles (half the weight loads per
// wave, twice the LDS reads) ties within 2 %.
#ifndef CONV_MT
#define CONV_MT 4
#endif

#ifndef CONV_WM
#define CONV_WM 2
#endif
// Blocks 4-6 (8, 4, 2 columns): 1 x 4 waves with 8 x 1 tiles, so that a wave's m-tiles are whole columns of the image
// and which of them are edge columns is the same in every wave (conv_split.h, column tiles): per 4096 clips 6.50 ->
// 5.88 ms, 3.13 -> 2.73 ms and 1.63 -> 1.21 ms against the 2 x 2 form with position-major tiles.
// Block 3 (16 columns): the same in two launches, one per half of the image's width (a workgroup tile is 32 rows x 8
// columns; the neighbouring half's edge column is staged as real data, the image's own edge column skips its outer
// tap): 7.39 -> 7.22 ms per 4096 clips.
#ifndef B3_WM
#define B3_WM 1
#define B3_MT 8
#define B3_XT (COLUMN_TILES ? 2 : 1)      // half-width workgroup tiles exist as column tiles only
#endif
#ifndef B4_WM
#define B4_WM 1
#define B4_MT 8
#endif
#ifndef B56_WM
#define B56_WM 1
#define B56_MT 8
#endif

// One wave's [64 x 32*NT] += A[64 x 8*kgroups] * B over `kgroups` groups of 8
// contraction indices.  a0/a1: this lane's row of the two 32-row A tiles
// (LDS, already offset by 4*(lane>>5)); bp: packed B [kgroup][n][8] in float4
// units, already offset to this lane's (column, half); b_kg_stride in float4s.
// Within a k-group lanes 0-31 carry k = 0..3 and lanes 32-63 k = 4..7, on both
// operands, so each of the four MFMAs contracts two indices.
template <int NT>
__device__ __forceinline__ void mma_block(f32x16 (&acc)[2][NT], const float *a0, const float *a1,
                                          const float4 *bp, int kgroups, int b_kg_stride) {
#pragma unroll 2
    for (int kg = 0; kg < kgroups; ++kg) {
        const float4 va0 = *reinterpret_cast<const float4 *>(a0 + kg * 8);
        const float4 va1 = *reinterpret_cast<const float4 *>(a1 + kg * 8);
        float4 vb[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) vb[nt] = bp[kg * b_kg_stride + nt * 64];
        const float fa0[4] = {va0.x, va0.y, va0.z, va0.w};
        const float fa1[4] = {va1.x, va1.y, va1.z, va1.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float fb = i == 0 ? vb[nt].x : (i == 1 ? vb[nt].y : (i == 2 ? vb[nt].z : vb[nt].w));
                acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[i], fb, acc[0][nt], 0, 0, 0);
                acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[i], fb, acc[1][nt], 0, 0, 0);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// block 0: 1 -> 16 channels, pool (2,2)
// ---------------------------------------------------------------------------
// Both halves of the gated unit are affine in the 3x3 input patch p of a position:
//   z   = s * (W p) + t                       (conv, bias, BatchNorm)
//   lin = G z + gb = (G diag(s) W) p + (G t + gb)   (the GLU's 16 x 16 linear)
// so the host folds them into two 16 x 9 matrices (in float64) and the kernel never forms G z.
struct Conv0Params {
    float wz[16][9];     // s[co] * conv weight [co][ky*3+kx]
    float wl[16][9];     // (G diag(s) W)[co][tap]
    float t[16], b2[16]; // z bias, lin bias
};

// Weights sit in LDS as float4 rows per output channel (every lane reads the same address: a
// broadcast, no bank conflict): [co][0..4] = 9 z taps, 9 lin taps, t, b2.
__global__ __launch_bounds__(256, 2) void conv0_kernel(const float *__restrict__ x, const Conv0Params *__restrict__ prm,
                                                    float *__restrict__ y, long n_clips, int T, int F, int Tp) {
    // x [clip][T][F], y [clip][Tp][F/2][16]
    __shared__ float wsm[16][20];
    for (int i = threadIdx.x; i < 16 * 20; i += 256) {
        const int co = i / 20, j = i - co * 20;
        wsm[co][j] = j < 9 ? prm->wz[co][j] : (j < 18 ? prm->wl[co][j - 9] : (j == 18 ? prm->t[co] : prm->b2[co]));
    }
    __syncthreads();
    const int Fp = F / 2;
    const long total = n_clips * Tp * Fp;
    // one output position per thread, no grid-stride loop: with a loop the weight reads are
    // loop-invariant and the compiler hoists all of them into registers
    const long idx = blockIdx.x * 256L + threadIdx.x;
    if (idx < total) {
        const int w2 = static_cast<int>(idx % Fp);
        const long rest = idx / Fp;
        const int h2 = static_cast<int>(rest % Tp);
        const long b = rest / Tp;
        const float *xb = x + b * static_cast<long>(T) * F;
        float patch[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int hr = 2 * h2 - 1 + r;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int wc = 2 * w2 - 1 + c;
                patch[r][c] = (hr >= 0 && hr < T && wc >= 0 && wc < F) ? xb[static_cast<long>(hr) * F + wc] : 0.f;
            }
        }
        float o[16];
#pragma unroll
        for (int co = 0; co < 16; ++co) {
            float wv[20];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const float4 q = *reinterpret_cast<const float4 *>(&wsm[co][4 * j]);
                wv[4 * j] = q.x; wv[4 * j + 1] = q.y; wv[4 * j + 2] = q.z; wv[4 * j + 3] = q.w;
            }
            float z[4], l[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) { z[p] = wv[18]; l[p] = wv[19]; }
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const float xv = patch[(p >> 1) + k / 3][(p & 1) + k % 3];
                    z[p] = fmaf(xv, wv[k], z[p]);
                    l[p] = fmaf(xv, wv[9 + k], l[p]);
                }
            float sum = 0.f;
#pragma unroll
            for (int p = 0; p < 4; ++p) sum = fmaf(l[p], sigmoidf_fast(z[p]), sum);
            o[co] = 0.25f * sum;
            // keep each channel's weight reads next to their use (otherwise they are clustered up front and spill)
            __builtin_amdgcn_sched_barrier(0);
        }
        float4 *dst = reinterpret_cast<float4 *>(y + idx * 16);
#pragma unroll
        for (int v = 0; v < 4; ++v) dst[v] = make_float4(o[4 * v], o[4 * v + 1], o[4 * v + 2], o[4 * v + 3]);
    }
}

// Block 0 on the f16 matrix cores (split-f16, as conv_split_kernel): out[32 rows][32 positions] =
// Wfold[32][9 taps] . patches[9][32], one MFMA triple per member of the 2x2 pooling window, all four
// members of a pooled position in the same lane so that the pooling is a lane-local sum.
// The 32 rows are ordered so that a lane's 16 accumulator registers hold z (registers 0-7) and lin
// (8-15) of the SAME eight channels 8*half .. 8*half+7: no cross-lane traffic for the gate.
// z rows carry -log2(e) * z * 2^Sz (so sigmoid = 1 / (1 + exp2(acc * 2^-Sz))), lin rows 0.25 * lin * 2^Sl;
// the biases enter as the initial accumulator.  The contraction runs over two patch rows x four patch
// columns per lane half (12 real taps' worth of slots for 9 taps: the column the member does not cover meets a
// zero weight), which lets the four members share their split patch rows.
#ifndef C0_TH_V
#define C0_TH_V 8
#endif
#ifndef C0_OCC
#define C0_OCC 4
#endif
constexpr int C0_TH = C0_TH_V;                   // pooled rows per workgroup
constexpr int C0_HO = 4;                    // LDS column of input column 0 (keeps the float4 stores aligned)
constexpr int C0_RS = 128 + 2 * C0_HO;      // LDS row stride in floats
constexpr int C0_ROWS = 2 * C0_TH + 2;      // staged input rows
__global__ __launch_bounds__(256, C0_OCC) void conv0_split_kernel(const float *__restrict__ x,
                                                          const uint4 *__restrict__ wpk,     // [2 dx][2 planes][32 rows][16] f16
                                                          const float *__restrict__ bias32,  // [32] scaled row biases
                                                          float zscale, float lscale,        // 2^-Sz, 2^-Sl
                                                          float *__restrict__ y, int T, int Tp, int tiles_per_clip) {
    // x [clip][T][128], y [clip][Tp][64][16]
    __shared__ __attribute__((aligned(16))) float lds[(C0_ROWS + 4) * C0_RS];   // + 4 rows of zeros
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, col32 = lane & 31;
    const long clip = blockIdx.x / tiles_per_clip;
    const int h20 = static_cast<int>(blockIdx.x % tiles_per_clip) * C0_TH;
    const float *xc = x + clip * static_cast<long>(T) * 128;
    // The tile's features times 2^-E (exact) on their way into LDS; 2^E comes back with the output scales.  E is the
    // tile's own: block 0 reads the raw features (the power mel spectrogram: no log, no normalisation in front of
    // it), whose magnitude follows the audio's level squared -- int16-scale PCM or a loud recording puts them beyond
    // the f16 range (65504), a quiet one far down in the f16 subnormals where the hi/lo split loses its bits.  The
    // staged rows are therefore brought to max |x| in [2^13, 2^14) by an exact power of two before the split, and
    // the factor is undone, exactly again, behind the matrix product (the initial accumulator -- the biases -- is
    // scaled along).  E = 0 for an all-zero or non-finite tile; |E| <= 60.
    constexpr int C0_NLD = (C0_ROWS * 32 + 255) / 256;
    float4 stage[C0_NLD];
    unsigned mx = 0u;
#pragma unroll
    for (int jn = 0; jn < C0_NLD; ++jn) {
        const int i = tid + 256 * jn, rr = i >> 5, c4 = (i & 31) * 4, hr = 2 * h20 - 1 + rr;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < C0_ROWS * 32 && hr >= 0 && hr < T) v = *reinterpret_cast<const float4 *>(xc + static_cast<long>(hr) * 128 + c4);
        stage[jn] = v;
        const unsigned a = __builtin_bit_cast(unsigned, v.x) & 0x7fffffffu, b = __builtin_bit_cast(unsigned, v.y) & 0x7fffffffu;
        const unsigned c = __builtin_bit_cast(unsigned, v.z) & 0x7fffffffu, d = __builtin_bit_cast(unsigned, v.w) & 0x7fffffffu;
        mx = max(mx, max(max(a, b), max(c, d)));      // |x| as integers: same order as the floats, NaN / inf on top
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, static_cast<unsigned>(__shfl_xor(static_cast<int>(mx), o)));
    __shared__ unsigned tile_max[4];
    if (lane == 0) tile_max[wave] = mx;
    __syncthreads();
    const unsigned m = __builtin_amdgcn_readfirstlane(max(max(tile_max[0], tile_max[1]), max(tile_max[2], tile_max[3])));
    int E = 0;
    if (m != 0 && m < 0x7f800000u) {
        const int e = static_cast<int>(m >> 23) - 127;               // floor(log2 max|x|) (-127 for subnormals)
        E = e - 13;
        E = E < -60 ? -60 : (E > 60 ? 60 : E);
    }
    const float s_in = __builtin_ldexpf(1.0f, -E), s_out = __builtin_ldexpf(1.0f, E);
    zscale *= s_out;
    lscale *= s_out;
#pragma unroll
    for (int jn = 0; jn < C0_NLD; ++jn) {
        const int i = tid + 256 * jn, rr = i >> 5, c4 = (i & 31) * 4;
        const float4 v = stage[jn];
        if (i < C0_ROWS * 32)
            *reinterpret_cast<float4 *>(&lds[rr * C0_RS + C0_HO + c4]) = make_float4(v.x * s_in, v.y * s_in, v.z * s_in, v.w * s_in);
    }
    for (int i = tid; i < C0_ROWS * 2; i += 256) lds[(i >> 1) * C0_RS + ((i & 1) ? C0_HO + 128 : C0_HO - 1)] = 0.f;
    for (int i = tid; i < 4 * C0_RS; i += 256) lds[C0_ROWS * C0_RS + i] = 0.f;

    // A operands for the two horizontal members of the pooling window (dx = 0, 1): [dx][plane][32 rows][16]
    f16x8 a_hi[2], a_lo[2];
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
        a_hi[dx] = __builtin_bit_cast(f16x8, wpk[(dx * 2 + 0) * 64 + col32 * 2 + half]);
        a_lo[dx] = __builtin_bit_cast(f16x8, wpk[(dx * 2 + 1) * 64 + col32 * 2 + half]);
    }
    f32x16 cinit;
#pragma unroll
    for (int r = 0; r < 16; ++r) cinit[r] = bias32[acc_row(r, half)] * s_in;
    __syncthreads();

    // groups of 32 pooled positions (half a pooled row), dealt round-robin to the waves
#pragma unroll 1
    for (int gi = wave; gi < 2 * C0_TH; gi += 4) {
        const int prow = gi >> 1, w2 = 32 * (gi & 1) + col32, h2 = h20 + prow;
        // patch[r][c] of this pooled position = lds[(2 * prow + r) * RS + HO - 1 + 2 * w2 + c]
        const int pb = (2 * prow) * C0_RS + C0_HO - 1 + 2 * w2;
        // A lane contributes eight contraction indices = two patch rows x four patch columns: lanes 0-31 rows
        // dy, dy + 1 of vertical member dy, lanes 32-63 row dy + 2 and four zeros.  Every patch row is split
        // into f16 halves once and serves both vertical members and -- the column shift being in the A
        // operand -- both horizontal ones.  Row slots P, Q, R = patch rows 0, 1, 2 (lanes 0-31) / rows 2, 3 and
        // a zeroed LDS row (lanes 32-63): member dy = 1 contracts (Q, R) in every lane, dy = 0 (P, Q) or (P, R).
        const float *rp[3] = {lds + pb + (half ? 2 * C0_RS : 0), lds + pb + (half ? 3 * C0_RS : C0_RS),
                              lds + (half ? C0_ROWS * C0_RS : pb + 2 * C0_RS)};
        unsigned sh[3][2], sl[3][2];                     // [slot][column pair]: packed f16 hi / lo halves
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            split2_f16(rp[k][0], rp[k][1], sh[k][0], sl[k][0]);
            split2_f16(rp[k][2], rp[k][3], sh[k][1], sl[k][1]);
        }
        f32x16 acc[4];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            unsigned yh[2], yl[2];                       // second row of the member
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                yh[c] = dy ? sh[2][c] : (half ? sh[2][c] : sh[1][c]);
                yl[c] = dy ? sl[2][c] : (half ? sl[2][c] : sl[1][c]);
            }
            const f16x8 b_hi = __builtin_bit_cast(f16x8, make_uint4(sh[dy][0], sh[dy][1], yh[0], yh[1]));
            const f16x8 b_lo = __builtin_bit_cast(f16x8, make_uint4(sl[dy][0], sl[dy][1], yl[0], yl[1]));
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int p = 2 * dy + dx;
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo[dx], b_hi, cinit, 0, 0, 0);
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[dx], b_lo, acc[p], 0, 0, 0);
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[dx], b_hi, acc[p], 0, 0, 0);
            }
        }
        float o[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) o[r] = 0.f;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float gate = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[p][r] * zscale));
                o[r] = fmaf(acc[p][r + 8], gate, o[r]);
            }
        if (h2 < Tp) {
            float4 *dst = reinterpret_cast<float4 *>(y + ((clip * Tp + h2) * 64 + w2) * 16 + 8 * half);
            dst[0] = make_float4(o[0] * lscale, o[1] * lscale, o[2] * lscale, o[3] * lscale);
            dst[1] = make_float4(o[4] * lscale, o[5] * lscale, o[6] * lscale, o[7] * lscale);
        }
    }
}

// ---------------------------------------------------------------------------
// blocks 1..6: implicit-GEMM conv + BN + GLU + pool(1,2)
// ---------------------------------------------------------------------------
template <int CIN, int COUT, int W>
struct ConvGeom {
    static constexpr int WN = COUT >= 64 ? 64 : COUT;   // columns per wave
    static constexpr int NT = WN / 32;
    static constexpr int WAVES_N = COUT / WN;
    static constexpr int WAVES_M = 4 / WAVES_N;
    static constexpr int M = 64 * WAVES_M;               // positions per workgroup
    static constexpr int TH = M / W;                     // rows (time steps) per workgroup
    static constexpr int PS = CIN + 4;                   // LDS stride of one input position (floats)
    static constexpr int TW = W + 2;
    static constexpr int IN_FLOATS = (TH + 2) * TW * PS;
    static constexpr int ZS = COUT + 4;
    static constexpr int Z_FLOATS = M * ZS;
    static constexpr int LDS_FLOATS = IN_FLOATS > Z_FLOATS ? IN_FLOATS : Z_FLOATS;
    static constexpr int KG = CIN / 8;
    static_assert(M % W == 0 && CIN % 8 == 0 && COUT % 32 == 0 && W % 2 == 0, "unsupported conv geometry");
};

template <int CIN, int COUT, int W>
__global__ __launch_bounds__(256) void conv_mfma_kernel(
    const float *__restrict__ x,       // [clip][H][W][CIN]
    const float4 *__restrict__ wp,     // packed conv weight [9][CIN/8][COUT][8]
    const float *__restrict__ bn_s, const float *__restrict__ bn_t,   // z = conv * s + t
    const float4 *__restrict__ gp,     // packed GLU weight [COUT/8][COUT][8]
    const float *__restrict__ gb,
    float *__restrict__ y,             // [clip][H][W/2][COUT]
    int H, int tiles_per_clip) {
    using G = ConvGeom<CIN, COUT, W>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, col32 = lane & 31;
    const int wm = wave / G::WAVES_N, wn = wave % G::WAVES_N;
    const long clip = blockIdx.x / tiles_per_clip;
    const int h0 = static_cast<int>(blockIdx.x % tiles_per_clip) * G::TH;
    const float *xc = x + clip * static_cast<long>(H) * W * CIN;

    // ---- stage input rows h0-1 .. h0+TH (zero outside the clip) with a zero column each side
    constexpr int ROW_F4 = W * CIN / 4;
    for (int i = tid; i < (G::TH + 2) * ROW_F4; i += 256) {
        const int rr = i / ROW_F4, e = (i - rr * ROW_F4) * 4;
        const int w = e / CIN, c = e - w * CIN;
        const int hr = h0 - 1 + rr;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (hr >= 0 && hr < H) v = *reinterpret_cast<const float4 *>(xc + (static_cast<long>(hr) * W + w) * CIN + c);
        *reinterpret_cast<float4 *>(lds + (rr * G::TW + 1 + w) * G::PS + c) = v;
    }
    for (int i = tid; i < (G::TH + 2) * 2 * (CIN / 4); i += 256) {
        const int rr = i / (2 * (CIN / 4)), rem = i - rr * 2 * (CIN / 4);
        const int side = rem / (CIN / 4), c = (rem - side * (CIN / 4)) * 4;
        *reinterpret_cast<float4 *>(lds + (rr * G::TW + side * (W + 1)) * G::PS + c) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();

    // ---- main loop: 9 taps x CIN/8 k-groups, weights streamed from global/L2
    f32x16 acc[2][G::NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < G::NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    const int pos0 = 64 * wm + col32, pos1 = pos0 + 32;
    const float *a0 = lds + ((pos0 / W) * G::TW + pos0 % W) * G::PS + 4 * half;
    const float *a1 = lds + ((pos1 / W) * G::TW + pos1 % W) * G::PS + 4 * half;
    const int ncol = wn * G::WN + col32;
    const float4 *bp = wp + ncol * 2 + half;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int toff = ((tap / 3) * G::TW + (tap % 3)) * G::PS;
        mma_block<G::NT>(acc, a0 + toff, a1 + toff, bp + tap * G::KG * COUT * 2, G::KG, COUT * 2);
    }

    // ---- BatchNorm affine, stash z for the GLU contraction
#pragma unroll
    for (int nt = 0; nt < G::NT; ++nt) {
        const float sc = bn_s[ncol + 32 * nt], sh = bn_t[ncol + 32 * nt];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = fmaf(acc[mt][nt][r], sc, sh);
    }
    __syncthreads();                                   // every wave is done with the input tile
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < G::NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                lds[(64 * wm + 32 * mt + acc_row(r, half)) * G::ZS + ncol + 32 * nt] = acc[mt][nt][r];
    __syncthreads();

    // ---- GLU: lin = z . Wg^T (+ bias), out = lin * sigmoid(z), then average adjacent freq pairs
    f32x16 lin[2][G::NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < G::NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) lin[mt][nt][r] = 0.f;
    mma_block<G::NT>(lin, lds + pos0 * G::ZS + 4 * half, lds + pos1 * G::ZS + 4 * half, gp + ncol * 2 + half,
                     COUT / 8, COUT * 2);

    float *yc = y + clip * static_cast<long>(H) * (W / 2) * COUT;
#pragma unroll
    for (int nt = 0; nt < G::NT; ++nt) {
        const float bias = gb[ncol + 32 * nt];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float o0 = (lin[mt][nt][r] + bias) * sigmoidf_fast(acc[mt][nt][r]);
                const float o1 = (lin[mt][nt][r + 1] + bias) * sigmoidf_fast(acc[mt][nt][r + 1]);
                const int pos = 64 * wm + 32 * mt + acc_row(r, half);     // even position, partner is pos + 1
                const int hr = h0 + pos / W;
                if (hr < H)
                    yc[(static_cast<long>(hr) * (W / 2) + (pos % W) / 2) * COUT + ncol + 32 * nt] = 0.5f * (o0 + o1);
            }
    }
}

// ---------------------------------------------------------------------------
// C[M][N] = A[M][K] . B + bias, B packed [K/8][N][8]; N % 128 == 0, K % 128 == 0
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linear_kernel(const float *__restrict__ A, const float4 *__restrict__ bpk,
                                                     const float *__restrict__ bias, float *__restrict__ C, long M,
                                                     int N, int K) {
    constexpr int KC = 128, AS = KC + 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, col32 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const long m0 = static_cast<long>(blockIdx.x) * 128;
    const int n0 = blockIdx.y * 128;
    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    const int ncol = n0 + wn * 64 + col32;
    for (int k0 = 0; k0 < K; k0 += KC) {
        if (k0) __syncthreads();
        for (int i = tid; i < 128 * (KC / 4); i += 256) {
            const int r = i / (KC / 4), c = (i - r * (KC / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m0 + r < M) v = *reinterpret_cast<const float4 *>(A + (m0 + r) * K + k0 + c);
            *reinterpret_cast<float4 *>(lds + r * AS + c) = v;
        }
        __syncthreads();
        const float *a0 = lds + (64 * wm + col32) * AS + 4 * half;
        mma_block<2>(acc, a0, a0 + 32 * AS, bpk + (static_cast<long>(k0 / 8) * N + ncol) * 2 + half, KC / 8, N * 2);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const float bv = bias[ncol + 32 * nt];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = m0 + 64 * wm + 32 * mt + acc_row(r, half);
                if (row < M) C[row * N + ncol + 32 * nt] = acc[mt][nt][r] + bv;
            }
    }
}

// ---------------------------------------------------------------------------
// GRU recurrence, hidden size 128, one workgroup = 32 clips x 1 direction
// ---------------------------------------------------------------------------
constexpr int GRU_H = 128;
constexpr int GRU_NB = 32;           // clips per workgroup (= MFMA M)
constexpr int GRU_HS = GRU_H + 4;    // LDS row stride of h

__global__ __launch_bounds__(256, 1) void gru_kernel(
    const float *__restrict__ gi,      // [clip][T][2][3H]: W_ih x + b_ih (+ b_hh for r, z)
    const float4 *__restrict__ whh,    // packed [2][H/8][3H][8]
    const float *__restrict__ bhn,     // [2][H]
    float *__restrict__ out,           // [clip][T][2H]
    long n_clips, int T) {
    __shared__ __attribute__((aligned(16))) float hs[2][GRU_NB * GRU_HS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, col32 = lane & 31;
    const int dir = blockIdx.y;
    const long b0 = static_cast<long>(blockIdx.x) * GRU_NB;
    const int j = 32 * wave + col32;                  // hidden unit owned by this lane

    // W_hh fragments for gates r, z, n of unit j: resident for the whole kernel
    float4 wreg[3][GRU_H / 8];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int kg = 0; kg < GRU_H / 8; ++kg)
            wreg[g][kg] = whh[((static_cast<long>(dir) * (GRU_H / 8) + kg) * (3 * GRU_H) + g * GRU_H + j) * 2 + half];
    const float b_hn = bhn[dir * GRU_H + j];

    for (int i = tid; i < 2 * GRU_NB * GRU_HS; i += 256) (&hs[0][0])[i] = 0.f;
    float hprev[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) hprev[r] = 0.f;
    __syncthreads();

    int cur = 0;
    for (int s = 0; s < T; ++s) {
        const int t = dir ? T - 1 - s : s;
        // this step's input projections (consumed after the MFMA loop)
        float gir[16], giz[16], gin[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long b = b0 + acc_row(r, half);
            gir[r] = giz[r] = gin[r] = 0.f;
            if (b < n_clips) {
                const float *p = gi + ((b * T + t) * 2 + dir) * (3 * GRU_H) + j;
                gir[r] = p[0];
                giz[r] = p[GRU_H];
                gin[r] = p[2 * GRU_H];
            }
        }
        f32x16 acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
        const float *arow = &hs[cur][col32 * GRU_HS + 4 * half];
#pragma unroll
        for (int kg = 0; kg < GRU_H / 8; ++kg) {
            const float4 va = *reinterpret_cast<const float4 *>(arow + kg * 8);
            const float fa[4] = {va.x, va.y, va.z, va.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const float4 wv = wreg[g][kg];
                    const float fb = i == 0 ? wv.x : (i == 1 ? wv.y : (i == 2 ? wv.z : wv.w));
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb, acc[g], 0, 0, 0);
                }
        }
        float *hnext = &hs[cur ^ 1][0];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float rg = sigmoidf_fast(gir[r] + acc[0][r]);
            const float zg = sigmoidf_fast(giz[r] + acc[1][r]);
            const float ng = tanhf_fast(fmaf(rg, acc[2][r] + b_hn, gin[r]));
            const float hn = fmaf(zg, hprev[r] - ng, ng);          // (1 - z) * n + z * h
            hprev[r] = hn;
            const int row = acc_row(r, half);
            hnext[row * GRU_HS + j] = hn;
            const long b = b0 + row;
            if (b < n_clips) out[(b * T + t) * (2 * GRU_H) + dir * GRU_H + j] = hn;
        }
        __syncthreads();
        cur ^= 1;
    }
}

// The same recurrence with the h . W_hh^T product in split-f16 arithmetic on v_mfma_f32_16x16x32_f16.
// Eight waves, two per SIMD: wave w owns the r, z, n columns of hidden units 16 w .. 16 w + 15 for all 32
// clips (two 16-row tiles), so the gate arithmetic is lane-local on accumulator registers, W_hh sits in
// registers as f16 (hi, lo) B fragments (96 VGPRs per wave), and h goes through LDS as two f16 planes,
// double buffered, one barrier per step.  A step costs each wave 72 MFMAs of 16 cycles and ~220
// vector-ALU instructions of gate arithmetic; a wave cannot overlap its own vector work with its own
// MFMAs, but the two waves of a SIMD overlap each other's (four waves of twice the size, one per SIMD:
// 1.19 instead of 0.96 ms per layer at 4096 clips).  What bounds it now is reading gi: 3.1 GB per layer in
// 64-byte pieces, 3.3 TB/s.  Running the two waves of a SIMD deliberately out of phase (the two 16-clip
// tiles are independent recurrences, so one wave can be in its gates while the other is in its chain,
// at the price of a second barrier per step) changes nothing for that reason.
constexpr int GRU_PSB = GRU_H * 2 + 16;      // bytes per clip row of one h plane
using f32x4v = __attribute__((ext_vector_type(4))) float;

// HEAD: the kernel is the last recurrent layer of a model with at most 16 classes and forms its direction's share of
// the head's logits itself, h . Wd[:, dir half]^T, instead of writing h out for a separate head kernel to read back
// (1 GB each way per 4096 clips): the A fragments a wave loads for the products of step s are h of step s - 1 for
// all 128 units, so wave mt adds twelve MFMAs per step for tile mt and stores 16 floats per clip and step.
template <bool HEAD>
__global__ __launch_bounds__(512, 1) void gru_split_kernel(
    const float *__restrict__ gi,      // [clip][T][2][3H]
    const uint4 *__restrict__ whh,     // packed split [2][H/16][2 planes][3H][16] f16, pre-scaled by 2^S
    float whh_oscale,                  // 2^-S
    const float *__restrict__ bhn,     // [2][H]
    float *__restrict__ out,           // [clip][T][2H] (not written with HEAD)
    long n_clips, int T,
    const uint4 *__restrict__ wds,     // HEAD: packed split [2][H/16][2 planes][16 classes][16] f16, times 2^Sd
    float wd_oscale,                   // 2^-Sd
    float *__restrict__ plog) {        // HEAD: [2][clip][T][16] partial logits
    __shared__ __attribute__((aligned(16))) char hs[2][2][GRU_NB * GRU_PSB];     // [buffer][plane]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, kg = lane >> 4;        // column (A: row) within a 16-wide tile; k-group / row group
    const int dir = blockIdx.y;
    const long b0 = static_cast<long>(blockIdx.x) * GRU_NB;
    const int j = 16 * wave + n16;                    // hidden unit owned by this lane

    // B fragment of k-step ks (32 indices): this lane's 8 indices 32 ks + 8 kg .. + 7 of column g * H + j
    f16x8 wh[3][GRU_H / 32], wl[3][GRU_H / 32];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int ks = 0; ks < GRU_H / 32; ++ks) {
            const int ks16 = 2 * ks + (kg >> 1);
            const uint4 *p = whh + ((static_cast<long>(dir) * (GRU_H / 16) + ks16) * 2 * (3 * GRU_H) + g * GRU_H + j) * 2 + (kg & 1);
            wh[g][ks] = __builtin_bit_cast(f16x8, p[0]);
            wl[g][ks] = __builtin_bit_cast(f16x8, p[2 * 3 * GRU_H]);
        }
    // the gates' exponentials are bare exp2: -log2(e) (sigmoid) and -2 log2(e) (tanh as 2 sigmoid(2 a) - 1) are
    // folded into the product's output scale and into the additions of gi
    constexpr float kC1 = -1.4426950408889634f, kC2 = -2.8853900817779268f;
    const float sh1 = whh_oscale * kC1, sh2 = whh_oscale * kC2;
    const float b_hn = bhn[dir * GRU_H + j] * kC2;
    // head weights of this direction's half in LDS, in the order the B fragments are read back: 16-byte piece
    // number ((ks * 2 + plane) * 16 + class) * 4 + kg (HEAD only; registers are too scarce to keep them there)
    __shared__ __attribute__((aligned(16))) uint4 wd_lds[HEAD ? (GRU_H / 32) * 2 * 16 * 4 : 1];
    if constexpr (HEAD) {
        for (int i = tid; i < (GRU_H / 32) * 2 * 16 * 4; i += 512) {
            const int kq = i & 3, cls = (i >> 2) & 15, plane = (i >> 6) & 1, ks = i >> 7;
            const int ks16 = 2 * ks + (kq >> 1);
            wd_lds[i] = wds[((static_cast<long>(dir) * (GRU_H / 16) + ks16) * 2 + plane) * 16 * 2 + cls * 2 + (kq & 1)];
        }
    }
    // partial logits of tile mt at time index t from the A fragments of h: rows = clips 16 mt + 4 kg + i, column = class
    auto head_part = [&](int mt, long t, const f16x8 (&vah)[GRU_H / 32], const f16x8 (&val)[GRU_H / 32]) {
        f32x4v al = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < GRU_H / 32; ++ks) {
            const f16x8 dh = __builtin_bit_cast(f16x8, wd_lds[((ks * 2 + 0) * 16 + n16) * 4 + kg]);
            const f16x8 dl = __builtin_bit_cast(f16x8, wd_lds[((ks * 2 + 1) * 16 + n16) * 4 + kg]);
            al = __builtin_amdgcn_mfma_f32_16x16x32_f16(val[ks], dh, al, 0, 0, 0);
            al = __builtin_amdgcn_mfma_f32_16x16x32_f16(vah[ks], dl, al, 0, 0, 0);
            al = __builtin_amdgcn_mfma_f32_16x16x32_f16(vah[ks], dh, al, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long b = b0 + 16 * mt + 4 * kg + i;
            if (b < n_clips) plog[((static_cast<long>(dir) * n_clips + b) * T + t) * 16 + n16] = al[i] * wd_oscale;
        }
    };

    for (int i = tid; i < static_cast<int>(sizeof(hs) / 4); i += 512) reinterpret_cast<unsigned *>(&hs[0][0][0])[i] = 0u;
    // accumulator register i of tile mt is clip 16 mt + 4 kg + i
    float hprev[2][4];
    const float *gip[2][4];
    float *outp[2][4];
    bool live[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            hprev[mt][i] = 0.f;
            const long b = b0 + 16 * mt + 4 * kg + i;
            live[mt][i] = b < n_clips;
            const long bc = live[mt][i] ? b : n_clips - 1;      // clips past the end of the batch: computed, never stored
            gip[mt][i] = gi + (bc * T * 2 + dir) * (3 * GRU_H) + j;
            outp[mt][i] = out + bc * T * (2 * GRU_H) + dir * GRU_H + j;
        }
    __syncthreads();

    // h of step s lives in buffer s & 1: a chain of step s reads buffer (s - 1) & 1 while the gates of step s
    // write the other one, so one barrier per step is enough.
    auto load_gi = [&](int mt, int s, float (&g)[3][4]) {
        const long t = dir ? T - 1 - s : s;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float *pr = gip[mt][i] + t * (6 * GRU_H);
#pragma unroll
            for (int q = 0; q < 3; ++q) g[q][i] = pr[q * GRU_H];
        }
    };
    auto chain = [&](int mt, int s, f32x4v (&acc)[3]) {
        const int rd = (s & 1) ^ 1;
        // A fragment: row (clip) 16 mt + n16, indices 32 ks + 8 kg .. + 7
        const char *ah = &hs[rd][0][(16 * mt + n16) * GRU_PSB + 16 * kg], *al = &hs[rd][1][(16 * mt + n16) * GRU_PSB + 16 * kg];
#pragma unroll
        for (int q = 0; q < 3; ++q) acc[q] = f32x4v{0.f, 0.f, 0.f, 0.f};
        f16x8 vah[GRU_H / 32], val[GRU_H / 32];
#pragma unroll
        for (int ks = 0; ks < GRU_H / 32; ++ks) {
            vah[ks] = *reinterpret_cast<const f16x8 *>(ah + ks * 64);
            val[ks] = *reinterpret_cast<const f16x8 *>(al + ks * 64);
        }
#pragma unroll
        for (int ks = 0; ks < GRU_H / 32; ++ks) {
#pragma unroll
            for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(val[ks], wh[q][ks], acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vah[ks], wl[q][ks], acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vah[ks], wh[q][ks], acc[q], 0, 0, 0);
        }
        if constexpr (HEAD) {
            if (wave == mt && s > 0) head_part(mt, dir ? T - s : s - 1, vah, val);     // h of the previous step
        }
    };
    auto gates = [&](int mt, int s, const f32x4v (&acc)[3], const float (&g)[3][4]) {
        const long t = dir ? T - 1 - s : s;
        char *nh = &hs[s & 1][0][0], *nl = &hs[s & 1][1][0];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float rg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(acc[0][i], sh1, kC1 * g[0][i])));
            const float zg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(acc[1][i], sh1, kC1 * g[1][i])));
            const float na = fmaf(rg, fmaf(acc[2][i], sh2, b_hn), kC2 * g[2][i]);                 // -2 log2(e) * argument
            const float ng = fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(na)), -1.0f);
            const float hn = fmaf(zg, hprev[mt][i] - ng, ng);      // (1 - z) * n + z * h
            hprev[mt][i] = hn;
            const int row = 16 * mt + 4 * kg + i;
            const _Float16 hi = static_cast<_Float16>(hn);         // |h| < 1: no clamp needed
            const _Float16 lo = static_cast<_Float16>(hn - static_cast<float>(hi));
            *reinterpret_cast<_Float16 *>(nh + row * GRU_PSB + j * 2) = hi;
            *reinterpret_cast<_Float16 *>(nl + row * GRU_PSB + j * 2) = lo;
            if (!HEAD && live[mt][i]) outp[mt][i][t * (2 * GRU_H)] = hn;
        }
    };
    // one 16-clip tile at a time: chain, then gates (the next tile's chain of this wave and the other wave's
    // work cover the latency of the gate arithmetic's tail)
    // gi of step s + 1 is requested before the work of step s and taken over after its barrier
    if (T <= 0) return;
    float g[2][3][4], gn[2][3][4];
    load_gi(0, 0, g[0]);
    load_gi(1, 0, g[1]);
#pragma unroll 1
    for (int s = 0; s < T; ++s) {
        const int sn = s + 1 < T ? s + 1 : s;
        load_gi(0, sn, gn[0]);
        load_gi(1, sn, gn[1]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x4v acc[3];
            chain(mt, s, acc);
            gates(mt, s, acc, g[mt]);
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) g[mt][q][i] = gn[mt][q][i];
    }
    if constexpr (HEAD) {
        // the last step's h (complete behind the loop's last barrier) still owes its logits
        if (wave < 2) {
            const int mt = wave, rd = (T - 1) & 1;
            const char *ah = &hs[rd][0][(16 * mt + n16) * GRU_PSB + 16 * kg], *al = &hs[rd][1][(16 * mt + n16) * GRU_PSB + 16 * kg];
            f16x8 vah[GRU_H / 32], val[GRU_H / 32];
#pragma unroll
            for (int ks = 0; ks < GRU_H / 32; ++ks) {
                vah[ks] = *reinterpret_cast<const f16x8 *>(ah + ks * 64);
                val[ks] = *reinterpret_cast<const f16x8 *>(al + ks * 64);
            }
            head_part(mt, dir ? 0 : T - 1, vah, val);
        }
    }
}

// probabilities from the two directions' partial logits: sigmoid(fwd + bwd + bias)
__global__ __launch_bounds__(256) void head_combine_kernel(const float *__restrict__ plog, const float *__restrict__ bd,
                                                           float *__restrict__ probs, long n_rows, int n_cls) {
    const long total = n_rows * n_cls;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += static_cast<long>(gridDim.x) * 256) {
        const long r = i / n_cls;
        const int c = static_cast<int>(i - r * n_cls);
        const float v = plog[r * 16 + c] + plog[(n_rows + r) * 16 + c] + bd[c];
        probs[i] = 1.0f / (1.0f + expf(-v));
    }
}

// Layer 0 of the GRU with its input projection inside the recurrence: gi = x . W_ih^T is formed per step from the
// conv stack's output instead of being written to memory by a GEMM and read back (3.1 GB each way per 4096 clips,
// which is what bounded the separate kernels).  Same wave layout as gru_split_kernel; per wave W_hh (hi, lo) and
// the hi halves of W_ih live in registers (144 VGPRs) together with the lo half of W_ih's n gate (16); the lo halves
// of its r and z gates (64 KB for one direction) sit in LDS beside the h planes and the x planes (the step's input
// rows as f16 hi/lo, staged one step ahead by all threads: 32 clips x 128 values = 8 per thread).  The products
// with x and with h keep separate accumulators because the two weight matrices carry different power-of-two scales.
constexpr int GRU0_WLO_BYTES = (GRU_H / 32) * 2 * GRU_H * 64;            // [k-step of 32][gate r, z][unit][32 k] f16
constexpr int GRU0_PLANES_BYTES = 2 * 2 * GRU_NB * GRU_PSB;              // [buffer][plane][clip row]
constexpr int GRU0_LDS_BYTES = 2 * GRU0_PLANES_BYTES + GRU0_WLO_BYTES;   // h planes, x planes, W_ih lo (r, z)

__global__ __launch_bounds__(512, 1) void gru0_fused_kernel(
    const float *__restrict__ x,       // [clip][T][H]: the conv stack's output
    const uint4 *__restrict__ wih,     // packed split [H/16][2 planes][6H][16] f16 (columns [dir][gate][unit]), times 2^Si
    float wih_oscale,                  // 2^-Si
    const float *__restrict__ bi,      // [2][3H]: b_ih (+ b_hh for r, z)
    const uint4 *__restrict__ whh,     // packed split [2][H/16][2 planes][3H][16] f16, times 2^Sh
    float whh_oscale,                  // 2^-Sh
    const float *__restrict__ bhn,     // [2][H]
    float *__restrict__ out,           // [clip][T][2H]
    long n_clips, int T) {
    extern __shared__ __attribute__((aligned(16))) char g0_lds[];
    char *hpl = g0_lds, *xpl = g0_lds + GRU0_PLANES_BYTES, *wlo = g0_lds + 2 * GRU0_PLANES_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, kg = lane >> 4;
    const int dir = blockIdx.y;
    const long b0 = static_cast<long>(blockIdx.x) * GRU_NB;
    const int j = 16 * wave + n16;                    // hidden unit owned by this lane
    if (T <= 0) return;

    // B fragments of k-step ks (32 indices): this lane's 8 indices 32 ks + 8 kg .. + 7 of column (gate, j)
    f16x8 wh[3][GRU_H / 32], wl[3][GRU_H / 32], vh[3][GRU_H / 32], vln[GRU_H / 32];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int ks = 0; ks < GRU_H / 32; ++ks) {
            const int ks16 = 2 * ks + (kg >> 1);
            const uint4 *p = whh + ((static_cast<long>(dir) * (GRU_H / 16) + ks16) * 2 * (3 * GRU_H) + g * GRU_H + j) * 2 + (kg & 1);
            wh[g][ks] = __builtin_bit_cast(f16x8, p[0]);
            wl[g][ks] = __builtin_bit_cast(f16x8, p[2 * 3 * GRU_H]);
            const uint4 *q = wih + (static_cast<long>(ks16) * 2 * (6 * GRU_H) + dir * 3 * GRU_H + g * GRU_H + j) * 2 + (kg & 1);
            vh[g][ks] = __builtin_bit_cast(f16x8, q[0]);
            if (g == 2) vln[ks] = __builtin_bit_cast(f16x8, q[2 * 6 * GRU_H]);
        }
    // lo halves of W_ih's r and z columns of this direction into LDS, in the order the waves read them back:
    // 16-byte piece number ((ks * 2 + g) * H + unit) * 4 + kg
    for (int i = tid; i < GRU0_WLO_BYTES / 16; i += 512) {
        const int kq = i & 3, unit = (i >> 2) % GRU_H, g = (i >> 2) / GRU_H % 2, ks = (i >> 2) / (2 * GRU_H);
        const int ks16 = 2 * ks + (kq >> 1);
        reinterpret_cast<uint4 *>(wlo)[i] =
            wih[((static_cast<long>(ks16) * 2 + 1) * (6 * GRU_H) + dir * 3 * GRU_H + g * GRU_H + unit) * 2 + (kq & 1)];
    }
    // the gates' exponentials are bare exp2: -log2(e) (sigmoid) and -2 log2(e) (tanh as 2 sigmoid(2 a) - 1) are
    // folded into the output scales and biases of the two products
    constexpr float kC1 = -1.4426950408889634f, kC2 = -2.8853900817779268f;
    const float si1 = wih_oscale * kC1, sh1 = whh_oscale * kC1, si2 = wih_oscale * kC2, sh2 = whh_oscale * kC2;
    const float b_r = bi[dir * 3 * GRU_H + j] * kC1, b_z = bi[dir * 3 * GRU_H + GRU_H + j] * kC1;
    const float b_in = bi[dir * 3 * GRU_H + 2 * GRU_H + j] * kC2, b_hn = bhn[dir * GRU_H + j] * kC2;
    for (int i = tid; i < 2 * GRU0_PLANES_BYTES / 4; i += 512) reinterpret_cast<unsigned *>(g0_lds)[i] = 0u;

    // accumulator register i of tile mt is clip 16 mt + 4 kg + i
    float hprev[2][4];
    float *outp[2][4];
    bool live[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            hprev[mt][i] = 0.f;
            const long b = b0 + 16 * mt + 4 * kg + i;
            live[mt][i] = b < n_clips;
            const long bc = live[mt][i] ? b : n_clips - 1;
            outp[mt][i] = out + bc * T * (2 * GRU_H) + dir * GRU_H + j;
        }
    // staging: thread -> clip tid / 16, eight consecutive inputs starting at 8 (tid % 16)
    const int sc = tid >> 4, sf = (tid & 15) * 8;
    const long sclip = b0 + sc < n_clips ? b0 + sc : n_clips - 1;
    const float *xrow = x + sclip * T * GRU_H + sf;
    f32x4v xa, xb;
    auto load_x = [&](int s) {
        const long t = dir ? T - 1 - s : s;
        xa = *reinterpret_cast<const f32x4v *>(xrow + t * GRU_H);
        xb = *reinterpret_cast<const f32x4v *>(xrow + t * GRU_H + 4);
    };
    auto store_x = [&](int buf) {
        uint4 hi, lo;
        split2_f16(xa[0], xa[1], hi.x, lo.x);
        split2_f16(xa[2], xa[3], hi.y, lo.y);
        split2_f16(xb[0], xb[1], hi.z, lo.z);
        split2_f16(xb[2], xb[3], hi.w, lo.w);
        char *base = xpl + buf * 2 * GRU_NB * GRU_PSB + sc * GRU_PSB + sf * 2;
        *reinterpret_cast<uint4 *>(base) = hi;
        *reinterpret_cast<uint4 *>(base + GRU_NB * GRU_PSB) = lo;
    };
    load_x(0);
    __syncthreads();                                   // planes zeroed, W_ih lo in place
    store_x(0);
    load_x(T > 1 ? 1 : 0);
    __syncthreads();

    // planes of step s: x in buffer s & 1 (written during step s - 1), h_{s-1} in buffer (s - 1) & 1
    auto chain = [&](int mt, int s, f32x4v (&ax)[3], f32x4v (&ah)[3]) {
        const int row = (16 * mt + n16) * GRU_PSB + 16 * kg;
        const char *xh = xpl + (s & 1) * 2 * GRU_NB * GRU_PSB + row, *xl = xh + GRU_NB * GRU_PSB;
        const char *hh = hpl + ((s & 1) ^ 1) * 2 * GRU_NB * GRU_PSB + row, *hl = hh + GRU_NB * GRU_PSB;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            ax[q] = f32x4v{0.f, 0.f, 0.f, 0.f};
            ah[q] = f32x4v{0.f, 0.f, 0.f, 0.f};
        }
        f16x8 fxh[GRU_H / 32], fxl[GRU_H / 32];
#pragma unroll
        for (int ks = 0; ks < GRU_H / 32; ++ks) {
            fxh[ks] = *reinterpret_cast<const f16x8 *>(xh + ks * 64);
            fxl[ks] = *reinterpret_cast<const f16x8 *>(xl + ks * 64);
        }
#pragma unroll
        for (int ks = 0; ks < GRU_H / 32; ++ks) {
            f16x8 vl[3];
#pragma unroll
            for (int q = 0; q < 2; ++q)
                vl[q] = *reinterpret_cast<const f16x8 *>(wlo + (((ks * 2 + q) * GRU_H + j) * 4 + kg) * 16);
            vl[2] = vln[ks];
#pragma unroll
            for (int q = 0; q < 3; ++q) ax[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fxl[ks], vh[q][ks], ax[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 3; ++q) ax[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fxh[ks], vl[q], ax[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 3; ++q) ax[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fxh[ks], vh[q][ks], ax[q], 0, 0, 0);
        }
        f16x8 fhh[GRU_H / 32], fhl[GRU_H / 32];
#pragma unroll
        for (int ks = 0; ks < GRU_H / 32; ++ks) {
            fhh[ks] = *reinterpret_cast<const f16x8 *>(hh + ks * 64);
            fhl[ks] = *reinterpret_cast<const f16x8 *>(hl + ks * 64);
        }
#pragma unroll
        for (int ks = 0; ks < GRU_H / 32; ++ks) {
#pragma unroll
            for (int q = 0; q < 3; ++q) ah[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fhl[ks], wh[q][ks], ah[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 3; ++q) ah[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fhh[ks], wl[q][ks], ah[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 3; ++q) ah[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fhh[ks], wh[q][ks], ah[q], 0, 0, 0);
        }
    };
    auto gates = [&](int mt, int s, const f32x4v (&ax)[3], const f32x4v (&ah)[3]) {
        const long t = dir ? T - 1 - s : s;
        char *nh = hpl + (s & 1) * 2 * GRU_NB * GRU_PSB, *nl = nh + GRU_NB * GRU_PSB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float rg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(ah[0][i], sh1, fmaf(ax[0][i], si1, b_r))));
            const float zg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(ah[1][i], sh1, fmaf(ax[1][i], si1, b_z))));
            const float na = fmaf(rg, fmaf(ah[2][i], sh2, b_hn), fmaf(ax[2][i], si2, b_in));     // -2 log2(e) * argument
            const float ng = fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(na)), -1.0f);
            const float hn = fmaf(zg, hprev[mt][i] - ng, ng);      // (1 - z) * n + z * h
            hprev[mt][i] = hn;
            const int row = 16 * mt + 4 * kg + i;
            const _Float16 hi = static_cast<_Float16>(hn);         // |h| < 1: no clamp needed
            const _Float16 lo = static_cast<_Float16>(hn - static_cast<float>(hi));
            *reinterpret_cast<_Float16 *>(nh + row * GRU_PSB + j * 2) = hi;
            *reinterpret_cast<_Float16 *>(nl + row * GRU_PSB + j * 2) = lo;
            if (live[mt][i]) outp[mt][i][t * (2 * GRU_H)] = hn;
        }
    };
#pragma unroll 1
    for (int s = 0; s < T; ++s) {
        // the rows of step s + 1 (requested during step s - 1) go into the other pair of x planes, and the rows of
        // step s + 2 are requested
        store_x((s + 1) & 1);
        load_x(s + 2 < T ? s + 2 : T - 1);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x4v ax[3], ah[3];
            chain(mt, s, ax, ah);
            gates(mt, s, ax, ah);
        }
        __syncthreads();
    }
}

// Orders one wave's LDS traffic (DS operations of a wave execute in issue order: the compiler only has to
// keep that order).
__device__ __forceinline__ void wave_fence_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------
// strong head: probs = sigmoid(x . Wd^T + bd), x [rows][256]
// ---------------------------------------------------------------------------
constexpr int HEAD_K = 2 * GRU_H;
constexpr int HEAD_ROWS = 64;

__global__ __launch_bounds__(256) void head_kernel(const float *__restrict__ x, const float *__restrict__ wd,
                                                   const float *__restrict__ bd, float *__restrict__ probs,
                                                   long n_rows, int n_cls) {
    extern __shared__ __attribute__((aligned(16))) float head_lds[];
    float (*xs)[HEAD_K + 1] = reinterpret_cast<float (*)[HEAD_K + 1]>(head_lds);
    const int tid = threadIdx.x;
    const long r0 = static_cast<long>(blockIdx.x) * HEAD_ROWS;
    for (int i = tid; i < HEAD_ROWS * HEAD_K; i += 256) {
        const int r = i / HEAD_K, k = i - r * HEAD_K;
        xs[r][k] = r0 + r < n_rows ? x[(r0 + r) * HEAD_K + k] : 0.f;
    }
    __syncthreads();
    const int r = tid & 63;
    const int cg = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: weights become scalar loads
    for (int c = cg; c < n_cls; c += 4) {
        const float *wrow = wd + c * HEAD_K;
        float a0 = bd[c], a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 8
        for (int k = 0; k < HEAD_K; k += 4) {
            a0 = fmaf(xs[r][k], wrow[k], a0);
            a1 = fmaf(xs[r][k + 1], wrow[k + 1], a1);
            a2 = fmaf(xs[r][k + 2], wrow[k + 2], a2);
            a3 = fmaf(xs[r][k + 3], wrow[k + 3], a3);
        }
        if (r0 + r < n_rows) probs[(r0 + r) * n_cls + c] = sigmoidf_fast((a0 + a1) + (a2 + a3));
    }
}

// The same head for n_cls <= 32 (every configuration the reference trains) on the f32 matrix instruction
// (v_mfma_f32_32x32x2_f32: bit-for-bit an fmaf chain).  A wave is on its own: it walks over 32-row tiles,
// stages a quarter of a tile (32 rows x 64 features) at a time in its private LDS region with fully
// coalesced loads, requested one quarter ahead into registers, and keeps the whole weight matrix as B
// operands in registers (lane = class: 128 VGPRs).  The contraction order is permuted so that one 16-byte
// LDS read feeds four MFMAs: in MFMA i of step j lanes 0-31 carry feature 8 j + i, lanes 32-63 feature
// 8 j + 4 + i.
constexpr int HEAD_MAX_SMALL = 32;
constexpr int HEAD_NQ = 4;                   // staged pieces per tile
constexpr int HEAD_KQ = HEAD_K / HEAD_NQ;    // features per piece
constexpr int HEAD_XS = HEAD_KQ + 4;         // LDS row stride (floats): conflict-free ds_read_b128 down a column of rows

__global__ __launch_bounds__(256, 2) void head_small_kernel(const float *__restrict__ x, const float *__restrict__ wd,
                                                            const float *__restrict__ bd, float *__restrict__ probs,
                                                            long n_rows, int n_cls) {
    __shared__ __attribute__((aligned(16))) float head_lds[4][32 * HEAD_XS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform by construction; keeps the tile's buffer descriptor scalar
    const int half = lane >> 5, n = lane & 31;
    float *xs = head_lds[wave];
    // B operands: W[n][8 j + 4 half + i], zero for the padding classes
    float bw[HEAD_K / 8][4];
#pragma unroll
    for (int j = 0; j < HEAD_K / 8; ++j) {
        float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < n_cls) w = *reinterpret_cast<const float4 *>(wd + n * HEAD_K + 8 * j + 4 * half);
        bw[j][0] = w.x; bw[j][1] = w.y; bw[j][2] = w.z; bw[j][3] = w.w;
    }
    const float bias = n < n_cls ? bd[n] : 0.f;
    const long n_tiles = (n_rows + 31) / 32;
    const long gw = static_cast<long>(blockIdx.x) * 4 + wave, gstride = static_cast<long>(gridDim.x) * 4;
    // piece c of this wave = quarter (c % 4) of its (c / 4)-th tile: 32 rows x 64 floats = 8 float4 per lane,
    // float4 i = lane + 64 q: row i / 16, columns 4 (i % 16) .. + 3 (four whole 256-byte row pieces per load
    // instruction), through a buffer descriptor that ends at the tile's last valid row (no bounds branch)
    constexpr int NLD = 32 * (HEAD_KQ / 4) / 64;
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    f32x4 v[NLD];
    const int ld_off = (lane >> 4) * (HEAD_K * 4) + (lane & 15) * 16;
    auto load_piece = [&](long c) {
        const long tile = gw + (c / HEAD_NQ) * gstride;
        const long left = n_rows - tile * 32;
        const int rows = tile < n_tiles ? static_cast<int>(left < 32 ? left : 32) : 0;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(x) + (rows ? tile * 32 * HEAD_K : 0), 0, rows * HEAD_K * 4, 0x00020000);
        const int lo = ld_off + static_cast<int>(c % HEAD_NQ) * HEAD_KQ * 4;
#pragma unroll
        for (int q = 0; q < NLD; ++q)     // float4 lane + 64 q: row 4 q + lane / 16
            v[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lo + q * 4 * HEAD_K * 4, 0, 0));
    };
    load_piece(0);
    f32x16 acc;
    // one piece: registers -> LDS, request the next piece, 8 steps of one LDS read + four MFMAs
    auto piece = [&](long c, auto q_tag) {
        constexpr int J0 = decltype(q_tag)::value * (HEAD_KQ / 8);
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int i = lane + 64 * q;
            *reinterpret_cast<f32x4 *>(xs + (i >> 4) * HEAD_XS + 4 * (i & 15)) = v[q];
        }
        wave_fence_lds();
        load_piece(c + 1);
        const float *xr = xs + n * HEAD_XS + 4 * half;
#pragma unroll
        for (int j = 0; j < HEAD_KQ / 8; ++j) {
            const float4 a4 = *reinterpret_cast<const float4 *>(xr + 8 * j);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, bw[J0 + j][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, bw[J0 + j][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, bw[J0 + j][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, bw[J0 + j][3], acc, 0, 0, 0);
        }
        wave_fence_lds();                              // the region is rewritten by the next piece
    };
#pragma unroll 1
    for (long it = 0; gw + it * gstride < n_tiles; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        piece(HEAD_NQ * it, std::integral_constant<int, 0>{});
        piece(HEAD_NQ * it + 1, std::integral_constant<int, 1>{});
        piece(HEAD_NQ * it + 2, std::integral_constant<int, 2>{});
        piece(HEAD_NQ * it + 3, std::integral_constant<int, 3>{});
        const long row0 = (gw + it * gstride) * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long row = row0 + acc_row(r, half);
            if (n < n_cls && row < n_rows) probs[row * n_cls + n] = sigmoidf_fast(acc[r] + bias);
        }
    }
}

// Clips per pass through the conv stack (bounds the scratch), and clips per inner pass through blocks 2-4.  Measured
// on 4096 clips: the blocks with the largest activations per FLOP (2, 3) run fastest when their input was written
// just before (512-clip passes: part of it is still in the memory-side cache; with 1024 and more they lose 2-4 %),
// the light kernels (blocks 0, 1, 5, 6) when a launch is long (fewer launches, smaller tails: 4-6 % at 1024).
#ifndef SEDMI_CNN_CHUNK
#define SEDMI_CNN_CHUNK 1024
#endif
#ifndef SEDMI_CNN_SUB
#define SEDMI_CNN_SUB 512
#endif
constexpr int kCnnChunk = SEDMI_CNN_CHUNK, kCnnSub = SEDMI_CNN_SUB;
static_assert(kCnnSub > 0 && kCnnChunk % kCnnSub == 0, "inner passes tile the outer one");

}  // namespace

struct sedmi_crnn {
    int n_class = 0, device = 0, num_cus = 0;
    Conv0Params *d_c0 = nullptr;
    uint16_t *d_c0s = nullptr;         // block 0 for the matrix cores: split-f16 folded weights [2 dx][2][32][16]
    float *d_c0bias = nullptr;         // and the scaled row biases
    float c0_zscale = 1.f, c0_lscale = 1.f;
    float *d_wp[SEDMI_CRNN_LAYERS] = {};   // packed conv weights (blocks 1..6)
    float *d_s[SEDMI_CRNN_LAYERS] = {}, *d_t[SEDMI_CRNN_LAYERS] = {};
    float *d_gp[SEDMI_CRNN_LAYERS] = {}, *d_gb[SEDMI_CRNN_LAYERS] = {};
    // split-f16 variants (blocks 1..6): packed hi/lo weights, scales with the power-of-two weight scale undone
    uint16_t *d_wps[SEDMI_CRNN_LAYERS] = {}, *d_gps[SEDMI_CRNN_LAYERS] = {};
    float *d_ss[SEDMI_CRNN_LAYERS] = {}, *d_ts[SEDMI_CRNN_LAYERS] = {}, *d_gs[SEDMI_CRNN_LAYERS] = {}, *d_gbh[SEDMI_CRNN_LAYERS] = {};
    uint16_t *d_wihs[2] = {};          // split-f16 GRU input-projection weights
    uint16_t *d_wds = nullptr;         // split-f16 head weights per direction half (only for <= 16 classes)
    float wd_oscale = 1.f;
    float wih_oscale[2] = {1.f, 1.f};
    uint16_t *d_whhs[2] = {};          // split-f16 GRU recurrent weights, both directions
    float whh_oscale[2] = {1.f, 1.f};
    int precision = SEDMI_PRECISION_SPLIT_F16;
    float *d_wih[2] = {}, *d_bi[2] = {}, *d_whh[2] = {}, *d_bhn[2] = {};
    float *d_wd = nullptr, *d_bd = nullptr;
};

namespace {

constexpr int kFilters[SEDMI_CRNN_LAYERS] = {16, 32, 64, 128, 128, 128, 128};

template <int CIN, int COUT, int W>
int launch_conv(const sedmi_crnn *m, int layer, const float *x, float *y, long n_clips, int H, hipStream_t s) {
    using G = ConvGeom<CIN, COUT, W>;
    const int tiles = static_cast<int>(sedmi::ceil_div(H, G::TH));
    const size_t lds = sizeof(float) * G::LDS_FLOATS;
    auto kern = conv_mfma_kernel<CIN, COUT, W>;
    if (int rc_lds = sedmi::set_max_dynamic_lds(reinterpret_cast<const void *>(kern), lds)) return rc_lds;
    static const std::string name = "conv_mfma_kernel<" + std::to_string(CIN) + "," + std::to_string(COUT) + "," +
                                    std::to_string(W) + ">";
    const double pos = static_cast<double>(n_clips) * H * W;
    sedmi::ProfileScope prof(name.c_str(), s, pos * (2.0 * 9 * CIN * COUT + 2.0 * COUT * COUT),
                             pos * 4.0 * (CIN + COUT / 2.0));
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(n_clips * tiles)), dim3(256), lds, s, x,
                       reinterpret_cast<const float4 *>(m->d_wp[layer]), m->d_s[layer], m->d_t[layer],
                       reinterpret_cast<const float4 *>(m->d_gp[layer]), m->d_gb[layer], y, H, tiles);
    SEDMI_CHECK_LAUNCH();
    return SEDMI_OK;
}

template <int CIN, int COUT, int W, int WAVES_M, int MT, int CCH, int XT = 1, int XI = 0>
int launch_conv_split(const sedmi_crnn *m, int layer, const float *x, float *y, long n_clips, int H, hipStream_t s) {
    using G = SplitGeom<CIN, COUT, W, WAVES_M, MT, CCH, XT>;
    const int tiles = static_cast<int>(sedmi::ceil_div(H, G::TH));
    const size_t lds = G::LDS_BYTES;
    auto kern = conv_split_kernel<CIN, COUT, W, WAVES_M, MT, CCH, XT, XI>;
    if (int rc_lds = sedmi::set_max_dynamic_lds(reinterpret_cast<const void *>(kern), lds)) return rc_lds;
    static const std::string name = "conv_split_kernel<" + std::to_string(CIN) + "," + std::to_string(COUT) + "," +
                                    std::to_string(W) + ">";
    {
        const double pos = static_cast<double>(n_clips) * H * W / XT;       // a launch covers one of the XT tiles across the width
        sedmi::ProfileScope prof(name.c_str(), s, pos * (2.0 * 9 * CIN * COUT + 2.0 * COUT * COUT),
                                 pos * 4.0 * (CIN + COUT / 2.0));
        const unsigned n_tiles = static_cast<unsigned>(n_clips * tiles);
        hipLaunchKernelGGL(kern, dim3((n_tiles + 7) / 8 * 8), dim3(256), lds, s, x,
                           reinterpret_cast<const uint4 *>(m->d_wps[layer]), m->d_ss[layer], m->d_ts[layer],
                           reinterpret_cast<const uint4 *>(m->d_gps[layer]), m->d_gs[layer], m->d_gbh[layer], y, H, tiles,
                           n_tiles);
        SEDMI_CHECK_LAUNCH();
    }
    if constexpr (XI + 1 < XT) return launch_conv_split<CIN, COUT, W, WAVES_M, MT, CCH, XT, XI + 1>(m, layer, x, y, n_clips, H, s);
    return SEDMI_OK;
}

struct Scratch {
    float *actA, *actB, *seq, *gi, *rec0, *rec1;
    size_t bytes;
};

Scratch carve(void *ws, long n_clips, long Tp, bool with_cnn, bool with_gru) {
    Scratch sc{};
    size_t off = 0;
    auto take = [&](size_t n_floats) {
        float *p = reinterpret_cast<float *>(static_cast<char *>(ws) + off);
        off += sedmi::align_up(n_floats * sizeof(float), 256);
        return p;
    };
    const long ch = std::min<long>(n_clips, kCnnChunk);
    if (with_cnn) {
        sc.actA = take(static_cast<size_t>(ch) * Tp * 1024);
        sc.actB = take(static_cast<size_t>(ch) * Tp * 1024);
    }
    if (with_gru) {
        sc.seq = take(static_cast<size_t>(n_clips) * Tp * GRU_H);
        sc.gi = take(static_cast<size_t>(n_clips) * Tp * 6 * GRU_H);
        sc.rec0 = take(static_cast<size_t>(n_clips) * Tp * 2 * GRU_H);
        sc.rec1 = take(static_cast<size_t>(n_clips) * Tp * 2 * GRU_H);
    }
    sc.bytes = off;
    return sc;
}

int run_cnn(const sedmi_crnn *m, const float *feats, long n_clips, int T, float *seq, float *actA, float *actB,
            hipStream_t s) {
    const int Tp = T / 2;
    for (long c0 = 0; c0 < n_clips; c0 += kCnnChunk) {
        const long nc = std::min<long>(kCnnChunk, n_clips - c0);
        const float *x = feats + c0 * static_cast<long>(T) * 128;
        const long total0 = nc * Tp * 64;
        const unsigned grid0 = static_cast<unsigned>(sedmi::ceil_div(total0, 256));
        {
            const double pos = static_cast<double>(nc) * (2.0 * Tp) * 128;
            sedmi::ProfileScope prof(m->precision == SEDMI_PRECISION_SPLIT_F16 ? "conv0_split_kernel" : "conv0_kernel", s, pos * (2.0 * 9 * 16 + 2.0 * 16 * 16),
                                     4.0 * nc * (static_cast<double>(T) * 128 + Tp * 64.0 * 16));
            if (m->precision == SEDMI_PRECISION_SPLIT_F16) {
                const int tiles = static_cast<int>(sedmi::ceil_div(Tp, C0_TH));
                hipLaunchKernelGGL(conv0_split_kernel, dim3(static_cast<unsigned>(nc * tiles)), dim3(256), 0, s, x,
                                   reinterpret_cast<const uint4 *>(m->d_c0s), m->d_c0bias, m->c0_zscale, m->c0_lscale,
                                   actA, T, Tp, tiles);
            } else {
                hipLaunchKernelGGL(conv0_kernel, dim3(grid0), dim3(256), 0, s, x, m->d_c0, actA, nc, T, 128, Tp);
            }
        }
        SEDMI_CHECK_LAUNCH();
        int rc;
        if (m->precision == SEDMI_PRECISION_SPLIT_F16) {
            if ((rc = launch_conv_split<16, 32, 64, 4, 2, 16>(m, 1, actA, actB, nc, Tp, s))) return rc;
            // blocks 2-4 one inner pass at a time, each reading what the one before it has just written (a clip's
            // activations are Tp x 1024 floats behind blocks 1-3 and Tp x 512 behind block 4)
            const long sub = kCnnSub;
            for (long s0 = 0; s0 < nc; s0 += sub) {
                const long ns = std::min<long>(sub, nc - s0);
                if ((rc = launch_conv_split<32, 64, 32, 4, 2, 32>(m, 2, actB + s0 * Tp * 1024, actA + s0 * Tp * 1024, ns, Tp, s))) return rc;
                if ((rc = launch_conv_split<64, 128, 16, B3_WM, B3_MT, 32, B3_XT>(m, 3, actA + s0 * Tp * 1024, actB + s0 * Tp * 1024, ns, Tp, s))) return rc;
                if ((rc = launch_conv_split<128, 128, 8, B4_WM, B4_MT, 32>(m, 4, actB + s0 * Tp * 1024, actA + s0 * Tp * 512, ns, Tp, s))) return rc;
            }
            if ((rc = launch_conv_split<128, 128, 4, B56_WM, B56_MT, 32>(m, 5, actA, actB, nc, Tp, s))) return rc;
            if ((rc = launch_conv_split<128, 128, 2, B56_WM, B56_MT, 16>(m, 6, actB, seq + c0 * Tp * GRU_H, nc, Tp, s))) return rc;
            continue;
        }
        if ((rc = launch_conv<16, 32, 64>(m, 1, actA, actB, nc, Tp, s))) return rc;
        if ((rc = launch_conv<32, 64, 32>(m, 2, actB, actA, nc, Tp, s))) return rc;
        if ((rc = launch_conv<64, 128, 16>(m, 3, actA, actB, nc, Tp, s))) return rc;
        if ((rc = launch_conv<128, 128, 8>(m, 4, actB, actA, nc, Tp, s))) return rc;
        if ((rc = launch_conv<128, 128, 4>(m, 5, actA, actB, nc, Tp, s))) return rc;
        if ((rc = launch_conv<128, 128, 2>(m, 6, actB, seq + c0 * Tp * GRU_H, nc, Tp, s))) return rc;
    }
    return SEDMI_OK;
}

// SEDMI_HEAD_FUSED=0 keeps the head as a kernel of its own behind layer 1 (the comparison path)
bool head_fused_wanted() {
    const char *e = std::getenv("SEDMI_HEAD_FUSED");
    return !(e && e[0] == '0');
}

// SEDMI_GRU0_FUSED=0 selects the separate projection GEMM + recurrence for layer 0 (the comparison path)
bool gru0_fused_wanted() {
    const char *e = std::getenv("SEDMI_GRU0_FUSED");
    return !(e && e[0] == '0');
}

// plog != nullptr (split precision, at most 16 classes): layer 1 leaves its partial logits there
// ([2][clip][Tp][16]) and writes no rec1
int run_gru(const sedmi_crnn *m, const float *seq, long n_clips, int Tp, float *gi, float *rec0, float *rec1,
            hipStream_t s, float *plog = nullptr) {
    const long rows = n_clips * Tp;
    const size_t lds = sizeof(float) * 128 * 132;
    if (int rc_lds = sedmi::set_max_dynamic_lds(reinterpret_cast<const void *>(linear_kernel), lds)) return rc_lds;
    const float *in = seq;
    float *outs[2] = {rec0, rec1};
    for (int layer = 0; layer < 2; ++layer) {
        const int K = layer == 0 ? GRU_H : 2 * GRU_H;
        if (layer == 0 && m->precision == SEDMI_PRECISION_SPLIT_F16 && gru0_fused_wanted()) {
            // flops: projection + recurrence; bytes: the conv output in, h out
            sedmi::ProfileScope prof("gru0_fused_kernel", s, 2.0 * rows * 6 * GRU_H * K + 2.0 * rows * 2 * 3 * GRU_H * GRU_H,
                                     4.0 * rows * (GRU_H + 2.0 * GRU_H));
            if (int rc_lds = sedmi::set_max_dynamic_lds(reinterpret_cast<const void *>(gru0_fused_kernel), GRU0_LDS_BYTES)) return rc_lds;
            const dim3 ggrid(static_cast<unsigned>(sedmi::ceil_div(n_clips, GRU_NB)), 2);
            hipLaunchKernelGGL(gru0_fused_kernel, ggrid, dim3(512), GRU0_LDS_BYTES, s, in,
                               reinterpret_cast<const uint4 *>(m->d_wihs[0]), m->wih_oscale[0], m->d_bi[0],
                               reinterpret_cast<const uint4 *>(m->d_whhs[0]), m->whh_oscale[0], m->d_bhn[0], outs[0],
                               n_clips, Tp);
            SEDMI_CHECK_LAUNCH();
            in = outs[0];
            continue;
        }
        if (m->precision == SEDMI_PRECISION_SPLIT_F16) {
            sedmi::ProfileScope prof(layer == 0 ? "linear_split_kernel<768,128,2>" : "linear_split_kernel<768,256,4>", s,
                                     2.0 * rows * 6 * GRU_H * K, 4.0 * rows * (K + 6.0 * GRU_H));
            // persistent workgroups: as many as fit on the chip at once (two per CU at K = 128, one at K = 256)
            const long row_tiles = sedmi::ceil_div(rows, 128);
            const dim3 grid(static_cast<unsigned>(std::min<long>(row_tiles, static_cast<long>(m->num_cus) * (layer == 0 ? 2 : 1))));
            const uint4 *wsp = reinterpret_cast<const uint4 *>(m->d_wihs[layer]);
            if (layer == 0) {
                const size_t lds2 = 2 * 128 * (128 * 2 + 16);
                if (int rc_lds = sedmi::set_max_dynamic_lds(reinterpret_cast<const void *>(linear_split_kernel<768, 128, 2>), lds2)) return rc_lds;
                hipLaunchKernelGGL((linear_split_kernel<768, 128, 2>), grid, dim3(256), lds2, s, in, wsp, m->d_bi[layer],
                                   m->wih_oscale[layer], gi, rows);
            } else {
                const size_t lds2 = 2 * 128 * (256 * 2 + 16);
                if (int rc_lds = sedmi::set_max_dynamic_lds(reinterpret_cast<const void *>(linear_split_kernel<768, 256, 4>), lds2)) return rc_lds;
                hipLaunchKernelGGL((linear_split_kernel<768, 256, 4>), grid, dim3(512), lds2, s, in, wsp, m->d_bi[layer],
                                   m->wih_oscale[layer], gi, rows);
            }
        } else {
            sedmi::ProfileScope prof(layer == 0 ? "linear_kernel(K=128)" : "linear_kernel(K=256)", s,
                                     2.0 * rows * 6 * GRU_H * K, 4.0 * rows * (K + 6.0 * GRU_H));
            hipLaunchKernelGGL(linear_kernel, dim3(static_cast<unsigned>(sedmi::ceil_div(rows, 128)), 6), dim3(256),
                               lds, s, in, reinterpret_cast<const float4 *>(m->d_wih[layer]), m->d_bi[layer], gi,
                               rows, 6 * GRU_H, K);
        }
        SEDMI_CHECK_LAUNCH();
        {
            sedmi::ProfileScope prof(m->precision != SEDMI_PRECISION_SPLIT_F16 ? "gru_kernel"
                                     : layer == 1 && plog             ? "gru_split_kernel<true>"
                                                                      : "gru_split_kernel<false>", s, 2.0 * rows * 2 * 3 * GRU_H * GRU_H,
                                     4.0 * rows * (6.0 * GRU_H + 2.0 * GRU_H));
            const dim3 ggrid(static_cast<unsigned>(sedmi::ceil_div(n_clips, GRU_NB)), 2);
            if (m->precision == SEDMI_PRECISION_SPLIT_F16)
                if (layer == 1 && plog)
                    hipLaunchKernelGGL(gru_split_kernel<true>, ggrid, dim3(512), 0, s, gi,
                                       reinterpret_cast<const uint4 *>(m->d_whhs[layer]), m->whh_oscale[layer],
                                       m->d_bhn[layer], static_cast<float *>(nullptr), n_clips, Tp,
                                       reinterpret_cast<const uint4 *>(m->d_wds), m->wd_oscale, plog);
                else
                    hipLaunchKernelGGL(gru_split_kernel<false>, ggrid, dim3(512), 0, s, gi,
                                       reinterpret_cast<const uint4 *>(m->d_whhs[layer]), m->whh_oscale[layer],
                                       m->d_bhn[layer], outs[layer], n_clips, Tp, static_cast<const uint4 *>(nullptr), 0.f,
                                       static_cast<float *>(nullptr));
            else
                hipLaunchKernelGGL(gru_kernel, ggrid, dim3(256), 0, s, gi,
                                   reinterpret_cast<const float4 *>(m->d_whh[layer]), m->d_bhn[layer], outs[layer],
                                   n_clips, Tp);
        }
        SEDMI_CHECK_LAUNCH();
        in = outs[layer];
    }
    return SEDMI_OK;
}

int check_shape(const sedmi_crnn *m, int64_t n_clips, int64_t T) {
    SEDMI_REQUIRE(m != nullptr, "model is null");
    SEDMI_REQUIRE(n_clips >= 0 && T >= 0, "negative size");
    // per-clip activations are addressed with 32-bit byte offsets (4096 B per frame pair in blocks 0 and 1)
    SEDMI_REQUIRE(T <= 1000000 && n_clips < (1LL << 31), "shape too large: at most 1 000 000 frames per clip");
    return SEDMI_OK;
}

}  // namespace

extern "C" {

int sedmi_crnn_create(sedmi_crnn **out, const sedmi_crnn_weights *w) {
    SEDMI_REQUIRE(out && w, "null argument");
    *out = nullptr;
    SEDMI_REQUIRE(w->n_freq == 128, "only n_freq = 128 is supported (got %d)", w->n_freq);
    SEDMI_REQUIRE(w->hidden == GRU_H, "only GRU hidden size 128 is supported (got %d)", w->hidden);
    SEDMI_REQUIRE(w->n_class > 0 && w->n_class <= 4096, "n_class %d out of range", w->n_class);
    for (int i = 0; i < SEDMI_CRNN_LAYERS; ++i) {
        SEDMI_REQUIRE(w->filters[i] == kFilters[i], "filters[%d] = %d: only 16,32,64,128,128,128,128 is supported",
                      i, w->filters[i]);
        SEDMI_REQUIRE(w->pool_f[i] == 2 && w->pool_t[i] == (i == 0 ? 2 : 1),
                      "pooling of block %d must be (%d,2)", i, i == 0 ? 2 : 1);
        SEDMI_REQUIRE(w->conv_w[i] && w->conv_b[i] && w->bn_w[i] && w->bn_b[i] && w->bn_mean[i] && w->bn_var[i] &&
                          w->glu_w[i] && w->glu_b[i], "null weight pointer in block %d", i);
    }
    for (int l = 0; l < 2; ++l)
        for (int d = 0; d < 2; ++d)
            SEDMI_REQUIRE(w->gru_w_ih[l][d] && w->gru_w_hh[l][d] && w->gru_b_ih[l][d] && w->gru_b_hh[l][d],
                          "null GRU weight pointer (layer %d, direction %d)", l, d);
    SEDMI_REQUIRE(w->dense_w && w->dense_b, "null head weight pointer");

    auto *m = new sedmi_crnn();
    m->n_class = w->n_class;
    int rc = SEDMI_OK;
    auto fail = [&](int code) { sedmi_crnn_destroy(m); return code; };
    if (hipGetDevice(&m->device) != hipSuccess) return fail(sedmi::set_error(SEDMI_EHIP, "hipGetDevice failed"));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, m->device) != hipSuccess)
        return fail(sedmi::set_error(SEDMI_EHIP, "hipGetDeviceProperties failed"));
    m->num_cus = prop.multiProcessorCount;

    // BatchNorm (eval) as z = conv_nobias * s + t, s = gamma / sqrt(var + eps), t = (bias - mean) * s + beta
    std::vector<float> s[SEDMI_CRNN_LAYERS], t[SEDMI_CRNN_LAYERS];
    for (int i = 0; i < SEDMI_CRNN_LAYERS; ++i) {
        const int co = kFilters[i];
        s[i].resize(co); t[i].resize(co);
        for (int c = 0; c < co; ++c) {
            const float sc = w->bn_w[i][c] / std::sqrt(w->bn_var[i][c] + w->bn_eps);
            s[i][c] = sc;
            t[i][c] = (w->conv_b[i][c] - w->bn_mean[i][c]) * sc + w->bn_b[i][c];
        }
    }
    {
        Conv0Params p;
        for (int co = 0; co < 16; ++co) {
            double b2 = w->glu_b[0][co];
            for (int ci = 0; ci < 16; ++ci) b2 += static_cast<double>(w->glu_w[0][co * 16 + ci]) * t[0][ci];
            p.t[co] = t[0][co];
            p.b2[co] = static_cast<float>(b2);
            for (int k = 0; k < 9; ++k) {
                p.wz[co][k] = s[0][co] * w->conv_w[0][co * 9 + k];
                double acc = 0.0;
                for (int ci = 0; ci < 16; ++ci)
                    acc += static_cast<double>(w->glu_w[0][co * 16 + ci]) * s[0][ci] * w->conv_w[0][ci * 9 + k];
                p.wl[co][k] = static_cast<float>(acc);
            }
        }
        if (hipMalloc(reinterpret_cast<void **>(&m->d_c0), sizeof(p)) != hipSuccess ||
            hipMemcpy(m->d_c0, &p, sizeof(p), hipMemcpyHostToDevice) != hipSuccess)
            return fail(sedmi::set_error(SEDMI_EHIP, "uploading block 0 parameters failed"));
        // matrix-core variant: row acc_row(r, half) of the folded 32 x 9 matrix is z (r < 8) or lin (r >= 8) of
        // channel 8 * half + (r & 7); z rows times -log2(e), lin rows times the pooling factor 1/4
        const double kNegLog2e = -1.4426950408889634;
        float rows[32][9], rbias[32];
        for (int h = 0; h < 2; ++h)
            for (int r = 0; r < 16; ++r) {
                const int i = acc_row(r, h), c = 8 * h + (r & 7);
                for (int k = 0; k < 9; ++k)
                    rows[i][k] = static_cast<float>(r < 8 ? kNegLog2e * p.wz[c][k] : 0.25 * p.wl[c][k]);
                rbias[i] = static_cast<float>(r < 8 ? kNegLog2e * p.t[c] : 0.25 * p.b2[c]);
            }
        auto is_z = [](int i) { return (i & 16) == 0; };   // acc_row(r, h) < 16 exactly for r < 8
        float zw[16 * 9], lw[16 * 9];
        int nz = 0, nl = 0;
        for (int i = 0; i < 32; ++i)
            for (int k = 0; k < 9; ++k) (is_z(i) ? zw[nz++] : lw[nl++]) = rows[i][k];
        const int Sz = split_scale_exp(zw, nz), Sl = split_scale_exp(lw, nl);
        // contraction index k16 = 8 * lane half + 4 * e + c: patch row ky = e (lanes 0-31) or 2 (lanes 32-63, e = 0
        // only), patch column c; horizontal member dx sees kernel column kx = c - dx
        auto c0s = pack_split(2, 32, 0, [&](int dx, int k16, int n) {
            const int hf = k16 >> 3, e = (k16 >> 2) & 1, c = k16 & 3, kx = c - dx;
            if ((hf && e) || kx < 0 || kx > 2) return 0.f;
            const int ky = hf ? 2 : e;
            return std::ldexp(rows[n][ky * 3 + kx], is_z(n) ? Sz : Sl);
        });
        std::vector<float> c0bias(32);
        for (int i = 0; i < 32; ++i) c0bias[i] = std::ldexp(rbias[i], is_z(i) ? Sz : Sl);
        m->c0_zscale = std::ldexp(1.0f, -Sz);
        m->c0_lscale = std::ldexp(1.0f, -Sl);
        if ((rc = sedmi::upload(&m->d_c0s, c0s)) || (rc = sedmi::upload(&m->d_c0bias, c0bias))) return fail(rc);
    }
    for (int i = 1; i < SEDMI_CRNN_LAYERS; ++i) {
        const int ci = kFilters[i - 1], co = kFilters[i];
        std::vector<float> wp;
        wp.reserve(static_cast<size_t>(9) * ci * co);
        for (int tap = 0; tap < 9; ++tap) {
            const float *src = w->conv_w[i];
            auto blk = pack_kn8(ci, co, [&](int k, int n) { return src[(static_cast<size_t>(n) * ci + k) * 9 + tap]; });
            wp.insert(wp.end(), blk.begin(), blk.end());
        }
        const float *gsrc = w->glu_w[i];
        auto gp = pack_kn8(co, co, [&](int k, int n) { return gsrc[static_cast<size_t>(n) * co + k]; });
        std::vector<float> gb(w->glu_b[i], w->glu_b[i] + co);
        {
            // split-f16 copies: k-step order [chunk][tap][k-step in chunk], channels 16 per k-step
            static const int kChunk[SEDMI_CRNN_LAYERS] = {0, 16, 32, 32, 32, 32, 16};
            const int cch = kChunk[i], ks_per = cch / 16, nchunk = ci / cch;
            const float *src = w->conv_w[i];
            const int S = split_scale_exp(src, static_cast<size_t>(co) * ci * 9);
            auto wps = pack_split(9 * ci / 16, co, S, [&](int kidx, int k16, int n) {
                const int ks = kidx % ks_per, tap = (kidx / ks_per) % 9, c = kidx / (ks_per * 9);
                (void)nchunk;
                const int ch = c * cch + ks * 16 + k16;
                return src[(static_cast<size_t>(n) * ci + ch) * 9 + tap];
            });
            const int Sg = split_scale_exp(gsrc, static_cast<size_t>(co) * co);
            auto gps = pack_split(co / 16, co, Sg, [&](int kidx, int k16, int n) {
                return gsrc[static_cast<size_t>(n) * co + kidx * 16 + k16];
            });
            // The split path keeps z' = -log2(e) * z (so that the sigmoid's exponential is a bare exp2 of it):
            // the factor goes into the BatchNorm affine here and comes back out through the GLU scale, which
            // also carries the average pool's factor 1/2 (as does the GLU bias).
            const float kNegLog2e = -1.4426950408889634f, kNegLn2 = -0.6931471805599453f;
            std::vector<float> ss(co), ts(co), gs(co, std::ldexp(kNegLn2, -Sg - 1)), gbh(co);
            for (int c = 0; c < co; ++c) {
                ss[c] = std::ldexp(s[i][c], -S) * kNegLog2e;
                ts[c] = t[i][c] * kNegLog2e;
                gbh[c] = 0.5f * w->glu_b[i][c];
            }
            if ((rc = sedmi::upload(&m->d_wps[i], wps)) || (rc = sedmi::upload(&m->d_gps[i], gps)) ||
                (rc = sedmi::upload(&m->d_ss[i], ss)) || (rc = sedmi::upload(&m->d_ts[i], ts)) ||
                (rc = sedmi::upload(&m->d_gs[i], gs)) ||
                (rc = sedmi::upload(&m->d_gbh[i], gbh)))
                return fail(rc);
        }
        if ((rc = sedmi::upload(&m->d_wp[i], wp)) || (rc = sedmi::upload(&m->d_s[i], s[i])) ||
            (rc = sedmi::upload(&m->d_t[i], t[i])) || (rc = sedmi::upload(&m->d_gp[i], gp)) ||
            (rc = sedmi::upload(&m->d_gb[i], gb)))
            return fail(rc);
    }
    for (int l = 0; l < 2; ++l) {
        const int K = l == 0 ? GRU_H : 2 * GRU_H;
        // input projection for both directions: columns [dir][gate][unit]
        auto wih = pack_kn8(K, 6 * GRU_H, [&](int k, int n) {
            return w->gru_w_ih[l][n / (3 * GRU_H)][static_cast<size_t>(n % (3 * GRU_H)) * K + k];
        });
        std::vector<float> bi(6 * GRU_H), bhn(2 * GRU_H), whh;
        for (int d = 0; d < 2; ++d) {
            for (int n = 0; n < 3 * GRU_H; ++n)
                bi[d * 3 * GRU_H + n] = w->gru_b_ih[l][d][n] + (n < 2 * GRU_H ? w->gru_b_hh[l][d][n] : 0.f);
            for (int n = 0; n < GRU_H; ++n) bhn[d * GRU_H + n] = w->gru_b_hh[l][d][2 * GRU_H + n];
            const float *src = w->gru_w_hh[l][d];
            auto blk = pack_kn8(GRU_H, 3 * GRU_H, [&](int k, int n) { return src[static_cast<size_t>(n) * GRU_H + k]; });
            whh.insert(whh.end(), blk.begin(), blk.end());
        }
        {
            int S = 100;
            for (int d = 0; d < 2; ++d) S = std::min(S, split_scale_exp(w->gru_w_ih[l][d], static_cast<size_t>(3) * GRU_H * K));
            auto wihs = pack_split(K / 16, 6 * GRU_H, S, [&](int ks, int k16, int n) {
                return w->gru_w_ih[l][n / (3 * GRU_H)][static_cast<size_t>(n % (3 * GRU_H)) * K + ks * 16 + k16];
            });
            m->wih_oscale[l] = std::ldexp(1.0f, -S);
            if ((rc = sedmi::upload(&m->d_wihs[l], wihs))) return fail(rc);
            int Sh = 100;
            for (int d = 0; d < 2; ++d) Sh = std::min(Sh, split_scale_exp(w->gru_w_hh[l][d], static_cast<size_t>(3) * GRU_H * GRU_H));
            std::vector<uint16_t> whhs;
            for (int d = 0; d < 2; ++d) {
                const float *src = w->gru_w_hh[l][d];
                auto blk = pack_split(GRU_H / 16, 3 * GRU_H, Sh, [&](int ks, int k16, int n) {
                    return src[static_cast<size_t>(n) * GRU_H + ks * 16 + k16];
                });
                whhs.insert(whhs.end(), blk.begin(), blk.end());
            }
            m->whh_oscale[l] = std::ldexp(1.0f, -Sh);
            if ((rc = sedmi::upload(&m->d_whhs[l], whhs))) return fail(rc);
        }
        if ((rc = sedmi::upload(&m->d_wih[l], wih)) || (rc = sedmi::upload(&m->d_bi[l], bi)) ||
            (rc = sedmi::upload(&m->d_whh[l], whh)) || (rc = sedmi::upload(&m->d_bhn[l], bhn)))
            return fail(rc);
    }
    {
        std::vector<float> wd(w->dense_w, w->dense_w + static_cast<size_t>(w->n_class) * HEAD_K);
        std::vector<float> bd(w->dense_b, w->dense_b + w->n_class);
        if ((rc = sedmi::upload(&m->d_wd, wd)) || (rc = sedmi::upload(&m->d_bd, bd))) return fail(rc);
        if (w->n_class <= 16) {
            // [dir][k-step of 16][plane][16 classes][16]: column n = class n (zero beyond n_class), rows = the
            // direction's 128 inputs of the head
            const int Sd = split_scale_exp(w->dense_w, static_cast<size_t>(w->n_class) * HEAD_K);
            std::vector<uint16_t> wds;
            for (int d = 0; d < 2; ++d) {
                auto blk = pack_split(GRU_H / 16, 16, Sd, [&](int ks, int k16, int n) {
                    return n < w->n_class ? w->dense_w[static_cast<size_t>(n) * HEAD_K + d * GRU_H + ks * 16 + k16] : 0.f;
                });
                wds.insert(wds.end(), blk.begin(), blk.end());
            }
            m->wd_oscale = std::ldexp(1.0f, -Sd);
            if ((rc = sedmi::upload(&m->d_wds, wds))) return fail(rc);
        }
    }
    *out = m;
    return SEDMI_OK;
}

int sedmi_crnn_destroy(sedmi_crnn *m) {
    if (!m) return SEDMI_OK;
    (void)hipFree(m->d_c0);
    (void)hipFree(m->d_c0s);
    (void)hipFree(m->d_c0bias);
    for (int i = 0; i < SEDMI_CRNN_LAYERS; ++i) {
        (void)hipFree(m->d_wp[i]); (void)hipFree(m->d_s[i]); (void)hipFree(m->d_t[i]);
        (void)hipFree(m->d_gp[i]); (void)hipFree(m->d_gb[i]);
        (void)hipFree(m->d_wps[i]); (void)hipFree(m->d_gps[i]); (void)hipFree(m->d_ss[i]); (void)hipFree(m->d_ts[i]); (void)hipFree(m->d_gs[i]);
        (void)hipFree(m->d_gbh[i]);
    }
    for (int l = 0; l < 2; ++l) {
        (void)hipFree(m->d_wihs[l]);
        (void)hipFree(m->d_whhs[l]);
        (void)hipFree(m->d_wih[l]); (void)hipFree(m->d_bi[l]); (void)hipFree(m->d_whh[l]); (void)hipFree(m->d_bhn[l]);
    }
    (void)hipFree(m->d_wd); (void)hipFree(m->d_bd); (void)hipFree(m->d_wds);
    delete m;
    return SEDMI_OK;
}

int sedmi_crnn_set_precision(sedmi_crnn *m, int precision) {
    SEDMI_REQUIRE(m != nullptr, "model is null");
    SEDMI_REQUIRE(precision == SEDMI_PRECISION_F32 || precision == SEDMI_PRECISION_SPLIT_F16,
                  "unknown precision %d", precision);
    m->precision = precision;
    return SEDMI_OK;
}

int64_t sedmi_crnn_out_frames(const sedmi_crnn *m, int64_t T) {
    if (!m || T < 0) return sedmi::set_error(SEDMI_EINVAL, "bad model or T");
    return T / 2;
}

size_t sedmi_crnn_workspace_bytes(const sedmi_crnn *m, int64_t n_clips, int64_t T) {
    if (!m || n_clips <= 0 || T < 2) return 0;
    return carve(nullptr, n_clips, T / 2, true, true).bytes;
}

int sedmi_crnn_cnn_f32(const sedmi_crnn *m, const float *feats, int64_t n_clips, int64_t T, float *seq, void *ws,
                       size_t ws_bytes, void *stream) {
    int rc = check_shape(m, n_clips, T);
    if (rc) return rc;
    if (n_clips == 0 || T < 2) return SEDMI_OK;
    SEDMI_REQUIRE(feats && seq && ws, "null data pointer");
    Scratch sc = carve(ws, n_clips, T / 2, true, false);
    SEDMI_REQUIRE(ws_bytes >= sc.bytes, "workspace too small: need %zu bytes, got %zu", sc.bytes, ws_bytes);
    return run_cnn(m, feats, n_clips, static_cast<int>(T), seq, sc.actA, sc.actB,
                   static_cast<hipStream_t>(stream));
}

int sedmi_crnn_gru_f32(const sedmi_crnn *m, const float *seq, int64_t n_clips, int64_t Tp, float *rec, void *ws,
                       size_t ws_bytes, void *stream) {
    int rc = check_shape(m, n_clips, Tp);
    if (rc) return rc;
    if (n_clips == 0 || Tp == 0) return SEDMI_OK;
    SEDMI_REQUIRE(seq && rec && ws, "null data pointer");
    Scratch sc = carve(ws, n_clips, Tp, false, true);
    SEDMI_REQUIRE(ws_bytes >= sc.bytes, "workspace too small: need %zu bytes, got %zu", sc.bytes, ws_bytes);
    return run_gru(m, seq, n_clips, static_cast<int>(Tp), sc.gi, sc.rec0, rec, static_cast<hipStream_t>(stream));
}

int sedmi_crnn_head_f32(const sedmi_crnn *m, const float *rec, int64_t n_rows, float *probs, void *stream) {
    SEDMI_REQUIRE(m != nullptr && n_rows >= 0, "bad argument");
    if (n_rows == 0) return SEDMI_OK;
    SEDMI_REQUIRE(rec && probs, "null data pointer");
    sedmi::ProfileScope prof(m->n_class <= HEAD_MAX_SMALL ? "head_small_kernel" : "head_kernel", static_cast<hipStream_t>(stream), 2.0 * n_rows * HEAD_K * m->n_class,
                             4.0 * n_rows * (HEAD_K + m->n_class));
    if (m->n_class <= HEAD_MAX_SMALL) {
        // persistent waves over 32-row tiles: two workgroups of four waves per CU
        const long wgs = std::min<long>(sedmi::ceil_div(n_rows, 128), 2L * m->num_cus);
        hipLaunchKernelGGL(head_small_kernel, dim3(static_cast<unsigned>(wgs)), dim3(256), 0,
                           static_cast<hipStream_t>(stream), rec, m->d_wd, m->d_bd, probs, static_cast<long>(n_rows),
                           m->n_class);
        SEDMI_CHECK_LAUNCH();
        return SEDMI_OK;
    }
    const size_t lds = sizeof(float) * HEAD_ROWS * (HEAD_K + 1);
    if (int rc_lds = sedmi::set_max_dynamic_lds(reinterpret_cast<const void *>(head_kernel), lds)) return rc_lds;
    hipLaunchKernelGGL(head_kernel, dim3(static_cast<unsigned>(sedmi::ceil_div(n_rows, HEAD_ROWS))), dim3(256), lds,
                       static_cast<hipStream_t>(stream), rec, m->d_wd, m->d_bd, probs, static_cast<long>(n_rows),
                       m->n_class);
    SEDMI_CHECK_LAUNCH();
    return SEDMI_OK;
}

int sedmi_crnn_forward_f32(const sedmi_crnn *m, const float *feats, int64_t n_clips, int64_t T, float *probs,
                           void *ws, size_t ws_bytes, void *stream) {
    int rc = check_shape(m, n_clips, T);
    if (rc) return rc;
    if (n_clips == 0 || T < 2) return SEDMI_OK;
    SEDMI_REQUIRE(feats && probs && ws, "null data pointer");
    const int Tp = static_cast<int>(T / 2);
    Scratch sc = carve(ws, n_clips, Tp, true, true);
    SEDMI_REQUIRE(ws_bytes >= sc.bytes, "workspace too small: need %zu bytes, got %zu", sc.bytes, ws_bytes);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((rc = run_cnn(m, feats, n_clips, static_cast<int>(T), sc.seq, sc.actA, sc.actB, s))) return rc;
    if (m->precision == SEDMI_PRECISION_SPLIT_F16 && m->d_wds != nullptr && head_fused_wanted()) {
        // the head rides in layer 1's recurrence: partial logits go where rec1 would have gone
        if ((rc = run_gru(m, sc.seq, n_clips, Tp, sc.gi, sc.rec0, sc.rec1, s, sc.rec1))) return rc;
        const long rows = n_clips * Tp;
        sedmi::ProfileScope prof("head_combine_kernel", s, 0.0, 4.0 * rows * (32.0 + m->n_class));
        const unsigned grid = static_cast<unsigned>(std::min<long>(sedmi::ceil_div(rows * m->n_class, 256), 8L * m->num_cus));
        hipLaunchKernelGGL(head_combine_kernel, dim3(grid), dim3(256), 0, s, sc.rec1, m->d_bd, probs, rows, m->n_class);
        SEDMI_CHECK_LAUNCH();
        return SEDMI_OK;
    }
    if ((rc = run_gru(m, sc.seq, n_clips, Tp, sc.gi, sc.rec0, sc.rec1, s))) return rc;
    return sedmi_crnn_head_f32(m, sc.rec1, n_clips * Tp, probs, stream);
}

}  // extern "C"

#ifdef SPLIT_TIMING
// diagnostic build only: copies out and clears the per-section tick sums of conv_split_kernel
extern "C" int sedmi_debug_split_ticks(unsigned long long *out72) {
    SEDMI_CHECK_HIP(hipDeviceSynchronize());
    SEDMI_CHECK_HIP(hipMemcpyFromSymbol(out72, HIP_SYMBOL(g_split_ticks), sizeof(unsigned long long) * 72));
    unsigned long long init[72] = {};
    for (int l = 0; l < 6; ++l) init[l * 12 + 10] = ~0ull;
    SEDMI_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_split_ticks), init, sizeof(init)));
    return SEDMI_OK;
}
#endif
