// ShakeRCNN eval-mode forward for gfx950 (MI355X).
//
// Stands in for ShakeRCNN.forward (models.py:124-142) and its blocks:
//   ShakeCNNBlock  (models.py:155-180): Conv(k, in -> 2*out, 'same') -> BatchNorm(eps 1e-5) ->
//                  GLU(dim=1) [a * sigmoid(b), channels halved] -> ShakeShakeBlock -> BatchNorm ->
//                  ReLU -> AvgPool
//   ShakeShakeBlock (models.py:70-100): x * sigmoid(a*conv3x3_1(x) + (1-a)*conv3x3_2(x)); in eval
//                  mode the reference draws a = torch.rand(1) per block per forward, so `a` is an
//                  argument here (three values per forward)
//   GRU(6144, 1024, bidirectional) -> LayerNorm(2048) -> LeakyReLU(0.01) -> Linear(2048, n_cls)
//
// Layout: activations channels-last [clip][time][freq][chan] float32.  The reference flattens
// [B, 64, T', 96] as (chan, freq) per time step; here the GRU input is (freq, chan) and the
// columns of W_ih are permuted once at model creation instead of moving data.
//
// Kernels
//   shake_conv1_split_kernel  block 1 conv (5x5, 1 -> 32) + BN + GLU on the matrix cores (K = 25 padded to 32).
//   conv2d_split_kernel  every other conv (3x3): implicit GEMM on the f16 matrix cores with the
//                        split-f16 arithmetic of conv_split.h, 8 x 32 position tiles with halo.
//                        MODE_GLU : BN affine + GLU halving (the two halves of a channel pair sit in
//                                   two accumulator tiles of the same lane).
//                        MODE_GATE: both shake branches as one GEMM with N = 2*C, mixed by `a` in the
//                                   epilogue, gate with the centre input, BN affine, ReLU, average
//                                   pool over positions ordered so that a pooling window is a run
//                                   of accumulator registers of one lane.
//                        MODE_POOL_W5 / MODE_POOL_NCHW (BasicRCNN): plain conv + affine with the max-pool
//                                   + ReLU that follows it in the epilogue.
//   gemm_split_kernel    C = A.B + bias for the GRU input projection (M = clips*T', N = K = 6144).
//   gru_big_persistent_kernel  the whole H = 1024 recurrence in one launch (up to 192 clips): W_hh
//                        (25 MB as f16 hi/lo planes) resident in the registers of 128 or 256
//                        workgroups, h exchanged between them through device memory every step.
//   gru_big_step_kernel  one recurrence step per launch for larger batches: every step streams
//                        W_hh from L2 / Infinity Cache; a workgroup owns 32 clips x 32 hidden units
//                        x 3 gates, its four waves split K and reduce through LDS, gate math included.
//   ln_head_kernel       LayerNorm + LeakyReLU + Linear, one wave per frame.
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>

namespace {

#include "mfma_common.h"
#include "conv_split.h"

// ---------------------------------------------------------------------------
// block 1: 5x5 conv, 1 -> 32 channels, BN affine, GLU halving -> u [B][H][W][16]
// ---------------------------------------------------------------------------
struct ShakeConv1Params {   // host-side staging of the block's parameters
    float w[32][28];     // [co][25 taps + 3 pad]
    float s[32], t[32];  // z = conv * s + t
};

// On the f16 matrix cores (split-f16): out[32 rows][32 positions] = Wfold[32][32] .
// patches[32][32] (25 taps, zero-weighted padding), two k-steps of 16.  BatchNorm's scale is folded into
// the weights, its shift is the initial accumulator.  The 32 rows are ordered so that a lane's 16
// accumulator registers hold the value half (registers 0-7) and the gate half (8-15) of the SAME eight
// output channels 8*half .. 8*half+7: the GLU is lane-local.  Gate rows carry -log2(e) * z * 2^Sb
// (sigmoid = 1 / (1 + exp2(acc * 2^-Sb))), value rows z * 2^Sa.  A lane supplies contraction indices
// 16 ks + 8 half + 0..7 of its position's 5x5 patch through a per-lane table of LDS offsets; the seven
// padding indices point at tap 0 (inside the same patch) and meet zero weights.
constexpr int C1_TR = 4;                     // time rows per workgroup
constexpr int C1_HO = 4;                     // LDS column of input column 0 (keeps the float4 stores aligned)
__global__ __launch_bounds__(256, 2) void shake_conv1_split_kernel(const float *__restrict__ x,
                                                                   const uint4 *__restrict__ wpk,     // [2 ks][2 planes][32 rows][16] f16
                                                                   const float *__restrict__ bias32,  // [32] scaled row shifts
                                                                   float ascale, float bscale,        // 2^-Sa, 2^-Sb
                                                                   float *__restrict__ u, int H, int W, int tiles_per_clip) {
    extern __shared__ __attribute__((aligned(16))) float c1_lds[];   // [C1_TR + 4][W + 2 * C1_HO]
    const int RS = W + 2 * C1_HO;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, col32 = lane & 31;
    const long clip = blockIdx.x / tiles_per_clip;
    const int y0 = static_cast<int>(blockIdx.x % tiles_per_clip) * C1_TR;
    const float *xc = x + clip * static_cast<long>(H) * W;
    const int row_f4 = W / 4;
    for (int i = tid; i < (C1_TR + 4) * row_f4; i += 256) {
        const int rr = i / row_f4, c4 = (i - rr * row_f4) * 4, yy = y0 - 2 + rr;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (yy >= 0 && yy < H) v = *reinterpret_cast<const float4 *>(xc + static_cast<long>(yy) * W + c4);
        *reinterpret_cast<float4 *>(&c1_lds[rr * RS + C1_HO + c4]) = v;
    }
    for (int i = tid; i < (C1_TR + 4) * 4; i += 256) {           // two halo columns on either side
        const int rr = i >> 2, j = i & 3;
        c1_lds[rr * RS + (j < 2 ? C1_HO - 2 + j : C1_HO + W + j - 2)] = 0.f;
    }
    f16x8 a_hi[2], a_lo[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        a_hi[ks] = __builtin_bit_cast(f16x8, wpk[(ks * 2 + 0) * 64 + col32 * 2 + half]);
        a_lo[ks] = __builtin_bit_cast(f16x8, wpk[(ks * 2 + 1) * 64 + col32 * 2 + half]);
    }
    f32x16 cinit;
#pragma unroll
    for (int r = 0; r < 16; ++r) cinit[r] = bias32[acc_row(r, half)];
    // LDS offset (floats, relative to the patch's top-left element) of this lane's 16 contraction indices
    int toff[2][8];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * ks + 8 * half + j, tap = k < 25 ? k : 0;
            toff[ks][j] = (tap / 5) * RS + tap % 5;
        }
    __syncthreads();

    const int groups_per_row = W / 32;
#pragma unroll 1
    for (int gi = wave; gi < C1_TR * groups_per_row; gi += 4) {
        const int rr = gi / groups_per_row, xq = (gi - rr * groups_per_row) * 32 + col32, yy = y0 + rr;
        // patch[ky][kx] = input (yy - 2 + ky, xq - 2 + kx) = c1_lds[(rr + ky) * RS + C1_HO - 2 + xq + kx]
        const float *pb = c1_lds + rr * RS + C1_HO - 2 + xq;
        f32x16 acc = cinit;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            unsigned bh[4], bl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split2_f16(pb[toff[ks][2 * j]], pb[toff[ks][2 * j + 1]], bh[j], bl[j]);
            const f16x8 b_hi = __builtin_bit_cast(f16x8, make_uint4(bh[0], bh[1], bh[2], bh[3]));
            const f16x8 b_lo = __builtin_bit_cast(f16x8, make_uint4(bl[0], bl[1], bl[2], bl[3]));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo[ks], b_hi, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[ks], b_lo, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[ks], b_hi, acc, 0, 0, 0);
        }
        float o[8];
#pragma unroll
        for (int r = 0; r < 8; ++r)
            o[r] = acc[r] * ascale * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[r + 8] * bscale));
        if (yy < H) {
            float4 *dst = reinterpret_cast<float4 *>(u + ((clip * H + yy) * static_cast<long>(W) + xq) * 16 + 8 * half);
            dst[0] = make_float4(o[0], o[1], o[2], o[3]);
            dst[1] = make_float4(o[4], o[5], o[6], o[7]);
        }
    }
}

// ---------------------------------------------------------------------------
// 3x3 conv as split-f16 implicit GEMM over 8 x 32 position tiles
// ---------------------------------------------------------------------------
constexpr int MODE_GLU = 0, MODE_GATE = 1, MODE_BIAS = 2, MODE_POOL_NCHW = 3, MODE_POOL_W5 = 4;
constexpr int C2_TR = 8, C2_TC = 32;
// columns a tile advances by: a (1,5) pooling window needs 4 more conv columns than it yields outputs
__host__ __device__ constexpr int c2_xstride(int mode) { return mode == MODE_POOL_W5 ? C2_TC - 4 : C2_TC; }

// TR: rows of a workgroup's tile (8, or 16 where 16-channel chunks leave room for two such workgroups per CU)
template <int CIN, int N, int CCH, int TR = C2_TR>
struct Conv2dGeom {
    static constexpr int TW = C2_TC + 2;
    static constexpr int PSB = CCH * 2 + 16;
    static constexpr int PLANE = (TR + 2) * TW * PSB;
    static constexpr int LDS_BYTES = 2 * PLANE;
    static constexpr int NCHUNK = CIN / CCH;
    static constexpr int KS = CCH / 16;
    static constexpr int NT = N / 32;
    static_assert(CIN % CCH == 0 && CCH % 16 == 0 && N % 32 == 0 && LDS_BYTES <= 80 * 1024, "bad conv2d geometry");
};

// v_permlane16_swap: rows 1 and 3 (lanes 16-31, 48-63) of `a` change places with rows 0 and 2 of `b`.  Through inline
// assembly: this toolchain's __builtin_amdgcn_permlane16_swap returns its first result twice.  The waits on either
// side cover the instruction's read-after-write distances, which the compiler does not track through an asm.
__device__ __forceinline__ void swap_rows16(float &a, float &b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}

// position p (0..255) of a tile -> (row, col) inside the tile; a pooling window is a run of
// consecutive p, hence of consecutive accumulator registers of one lane
template <int PH, int PW>
__device__ __forceinline__ void tile_pos(int p, int &ty, int &tx) {
    if (PH == 2 && PW == 2) {
        const int blk = p >> 2;
        ty = 2 * (blk / (C2_TC / 2)) + ((p >> 1) & 1);
        tx = 2 * (blk % (C2_TC / 2)) + (p & 1);
    } else {
        ty = p / C2_TC;
        tx = p % C2_TC;
    }
}

// WN: waves across the N columns (1: every wave has all of them and 64 positions; 2: a wave has every second
// 32-column tile and 128 positions, so that the weights it reads from L2 feed twice the MFMAs).
template <int CIN, int N, int CCH, int MODE, int PH, int PW, int WN, int TR = C2_TR>
__global__ __launch_bounds__(256, 2) void conv2d_split_kernel(
    const float *__restrict__ x,      // [clip][H][W][CIN]
    const uint4 *__restrict__ wp,     // packed split weights [9*CIN/16][2][N][16], k-step order chunk/tap/k
    const float *__restrict__ es, const float *__restrict__ et,     // z = acc * es + et per GEMM column
    float alpha,                                                     // MODE_GATE: branch mix
    const float *__restrict__ s2, const float *__restrict__ t2,     // MODE_GATE: second BatchNorm affine
    float *__restrict__ y,            // MODE_GLU: [clip][H][W][N/2]; MODE_GATE: [clip][H/PH][W/PW][N/2];
                                      // MODE_BIAS: [clip][H][W][N] (plain conv + per-column affine);
                                      // MODE_POOL_NCHW: [clip][N][H/2][W/2] (the same, max-pooled, ReLU);
                                      // MODE_POOL_W5: [clip][H][W-4][N] (the same, (1,5) max-pooled, ReLU)
    int H, int W, int tiles_x, int tiles_y, unsigned n_tiles) {
    using G = Conv2dGeom<CIN, N, CCH, TR>;
    constexpr int NT = G::NT / WN, MT = TR * WN / 4, CO = N / 2, PWV = 32 * MT;   // NT: column tiles per wave
    static_assert(TR % 8 == 0 && (MODE != MODE_POOL_W5 || TR == 8), "tile rows");
    static_assert(G::NT % (2 * WN) == 0 || WN == 1, "a wave needs both halves of the GEMM columns");
    static_assert(MODE != MODE_GATE || CIN == CO, "the gate multiplies the conv input channel by channel");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *plane_h = smem, *plane_l = smem + G::PLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, col32 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    auto colof = [&](int nt) { return col32 + 32 * (wn + WN * nt); };     // GEMM column of the wave's tile nt
    // workgroup b runs on XCD b % 8 and every XCD has its own L2: hand each XCD a contiguous range of
    // tiles so that the tiles it runs back to back are neighbours and share their halos through L2
    const unsigned per_xcd = (gridDim.x + 7) / 8;
    int bid = static_cast<int>((blockIdx.x % 8) * per_xcd + blockIdx.x / 8);
    if (static_cast<unsigned>(bid) >= n_tiles) return;
    const int tx0 = (bid % tiles_x) * c2_xstride(MODE);
    bid /= tiles_x;
    const int ty0 = (bid % tiles_y) * TR;
    const long clip = bid / tiles_y;
    const float *xc = x + clip * static_cast<long>(H) * W * CIN;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    const char *ah[MT], *al[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int ty, tx;
        tile_pos<PH, PW>(PWV * wm + 32 * mt + col32, ty, tx);
        const int off = (ty * G::TW + tx) * G::PSB + 16 * half;
        ah[mt] = plane_h + off;
        al[mt] = plane_l + off;
    }
    const uint4 *bp = wp + (col32 + 32 * wn) * 2 + half;

    // Staging map.  A 16-byte piece of the tile's (TR + 2) x 34 input positions is fetched by the same thread in every
    // pass and every chunk: columns 0..31 of RSTEP rows per pass ("main"), then the two rightmost halo columns of all
    // rows in one more pass ("edge").  Everything that depends on the thread is computed once per tile -- a load costs
    // one add, an LDS store none (its offset is an immediate).  The clip's input sits behind a buffer descriptor:
    // rows above and below the image come out of its range check as zero (the conv's padding), columns left and right
    // of the image get an offset that stays out of range whatever is added to it.
    constexpr int Q = CCH / 4;                          // pieces per position and chunk
    constexpr int ROWS = TR + 2;
    constexpr int RSTEP = 256 / (C2_TC * Q);            // rows per main pass
    constexpr int NMAIN = ROWS / RSTEP, NEDGE = 2 * Q * ROWS, NLD = NMAIN + 1;
    static_assert((Q == 4 || Q == 8) && ROWS % RSTEP == 0 && NEDGE <= 256, "staging map");
    constexpr unsigned kOutOfRange = 0x80000000u;       // + (rows + chunks) of a tile stays below 2^32 (checked by the host)
    using f32x4v = __attribute__((ext_vector_type(4))) float;
    f32x4v v[NLD];
    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xc), 0, H * W * CIN * 4, 0x00020000);
    const unsigned rowb = static_cast<unsigned>(W) * CIN * 4;
    unsigned g_main, g_edge;
    int l_main, l_edge;
    {
        const int q = tid % Q, col = (tid / Q) % C2_TC, rsel = tid / (Q * C2_TC);
        const int xx = tx0 - 1 + col;
        g_main = (xx >= 0 && xx < W) ? static_cast<unsigned>((ty0 - 1 + rsel) * W + xx) * (CIN * 4) + q * 16 : kOutOfRange;
        l_main = (rsel * G::TW + col) * G::PSB + q * 8;
        const int er = tid / (2 * Q), ec = (tid / Q) & 1, ex = tx0 + C2_TC - 1 + ec;
        g_edge = (tid < NEDGE && ex < W) ? static_cast<unsigned>((ty0 - 1 + er) * W + ex) * (CIN * 4) + q * 16 : kOutOfRange;
        l_edge = (er * G::TW + C2_TC + ec) * G::PSB + q * 8;
    }
    auto load_chunk = [&](int c) {
        const unsigned co = static_cast<unsigned>(c) * (CCH * 4);
#pragma unroll
        for (int j = 0; j < NMAIN; ++j)
            v[j] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(
                                                  xrs, static_cast<int>(g_main + (co + j * RSTEP * rowb)), 0, 0));
        v[NMAIN] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(xrs, static_cast<int>(g_edge + co), 0, 0));
    };
    // MODE_GATE with one staged chunk: the conv input at the lane's own output positions (the factor of the gate) is
    // requested here, behind the tile's staging loads and from the same cache lines, and waits in registers for the
    // epilogue: one load per value instead of two 16-bit LDS reads, two conversions and an add.  A position outside
    // the image reads zero (rows) or a neighbour's value (columns); neither reaches a stored output.
    // position of register r relative to the lane's register 0 (tile_pos of consecutive positions), and the pooled
    // column of the window that ends in register r relative to the lane's first window
    auto d_ty = [](int r) { return PH == 2 ? (r & 3) >> 1 : 0; };
    auto d_tx = [](int r) { return PH == 2 ? 4 * (r >> 2) + (r & 1) : 8 * (r >> 2) + (r & 3); };
    auto d_xo = [](int r) { return PH == 2 ? 2 * (r >> 2) : 4 * (r >> 2) + ((r & 3) >> 1); };
    constexpr bool UC_EARLY = MODE == MODE_GATE && G::NCHUNK == 1;
    constexpr int UC_NT = CO >= 32 ? CO / 32 / WN : 1, UC_R = CO == 16 ? 8 : 16;
    float ucv[UC_EARLY ? MT * UC_NT * UC_R : 1];
    auto load_uc = [&]() {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < UC_NT; ++nt) {
                int ty, tx;
                tile_pos<PH, PW>(PWV * wm + 32 * mt + 4 * half + (CO == 16 ? 16 * (col32 >> 4) : 0), ty, tx);
                const int ch = CO == 16 ? (col32 & 15) : colof(nt);
                const unsigned gbase = static_cast<unsigned>(((ty0 + ty) * W + tx0 + tx) * CIN + ch) * 4u;
#pragma unroll
                for (int r = 0; r < UC_R; ++r)
                    ucv[(mt * UC_NT + nt) * UC_R + r] = __builtin_bit_cast(
                        float, __builtin_amdgcn_raw_buffer_load_b32(
                                   xrs, static_cast<int>(gbase + d_ty(r) * rowb + d_tx(r) * CIN * 4), 0, 0));
            }
    };
    for (int c = 0; c < G::NCHUNK; ++c) {
        load_chunk(c);
        if constexpr (UC_EARLY) load_uc();
        if (c) __syncthreads();
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            uint2 hi, lo;
            split2_f16(v[j][0], v[j][1], hi.x, lo.x);
            split2_f16(v[j][2], v[j][3], hi.y, lo.y);
            if (j < NMAIN) {
                *reinterpret_cast<uint2 *>(plane_h + l_main + j * RSTEP * G::TW * G::PSB) = hi;
                *reinterpret_cast<uint2 *>(plane_l + l_main + j * RSTEP * G::TW * G::PSB) = lo;
            } else if (tid < NEDGE) {
                *reinterpret_cast<uint2 *>(plane_h + l_edge) = hi;
                *reinterpret_cast<uint2 *>(plane_l + l_edge) = lo;
            }
        }
        __syncthreads();
        split_gemm<MT, NT, 9 * G::KS, SPLIT_BDIST, WN>(acc, ah, al, bp + static_cast<long>(c) * (9 * G::KS) * (N * 4), N, [](int i) {
            const int tap = i / G::KS, ks = i % G::KS;
            return ((tap / 3) * G::TW + (tap % 3)) * G::PSB + ks * 32;
        });
    }

    // ---- epilogue
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        float sc = es[colof(nt)], sh = et[colof(nt)];
        if constexpr (MODE == MODE_GATE) {
            // the gate is sigmoid(a x1 + (1 - a) x2) = 1 / (1 + exp2(-log2(e) (a x1 + (1 - a) x2))): the branch weight
            // and the exponent factor go into the column's affine, the epilogue adds the two and calls exp2
            const bool first = CO >= 32 ? nt < NT / 2 : col32 < CO;
            const float f = -1.4426950408889634f * (first ? alpha : 1.0f - alpha);
            sc *= f;
            sh *= f;
        } else if constexpr (MODE == MODE_GLU) {
            if (nt >= NT / 2) {          // the GLU's gate half: sigmoid(x) = 1 / (1 + exp2(-log2(e) x))
                sc *= -1.4426950408889634f;
                sh *= -1.4426950408889634f;
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = fmaf(acc[mt][nt][r], sc, sh);
    }
    if constexpr (MODE == MODE_BIAS) {
        float *yc = y + clip * static_cast<long>(H) * W * N;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int ty, tx;
                tile_pos<PH, PW>(PWV * wm + 32 * mt + acc_row(r, half), ty, tx);
                const int yy = ty0 + ty, xx = tx0 + tx;
                if (yy < H && xx < W) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        yc[(static_cast<long>(yy) * W + xx) * N + colof(nt)] = acc[mt][nt][r];
                }
            }
    } else if constexpr (MODE == MODE_POOL_W5) {
        // conv + per-column affine, then MaxPool2d((1, 5), stride 1) + ReLU (models.py:32,52-53): y [clip][H][W - 4][N].
        // A tile yields 28 pooled columns from its 32 conv columns.  An m-tile is one row of 32 columns; a lane
        // holds 4 groups of 4 consecutive columns (8j + 4 * half + q), its partner (lane ^ 32) the groups in
        // between.  A lane finishes the windows that START in its own groups: they end in the partner's next group,
        // which is the partner's group j (half 0) or j + 1 (half 1) -- the partner sends exactly that.
        static_assert(PH == 1 && PW == 1, "row-wise pooling");
        const int Wo = W - 4;
        float *yc = y + clip * static_cast<long>(H) * Wo * N;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int ty, tx;
            tile_pos<PH, PW>(PWV * wm + 32 * mt, ty, tx);          // tx = 0: an m-tile is one tile row
            const int yy = ty0 + ty;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float *dst = yc + (static_cast<long>(yy) * Wo + tx0 + 4 * half) * N + colof(nt);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float own[4], nxt[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        own[q] = acc[mt][nt][4 * j + q];
                        // what the partner needs from this lane: its group j if this lane is half 1, else group j + 1
                        const float send = half ? own[q] : acc[mt][nt][j < 3 ? 4 * (j + 1) + q : 4 * j + q];
                        nxt[q] = __shfl_xor(send, 32);
                    }
                    const float s3 = own[3], s2 = fmaxf(own[2], s3), s1 = fmaxf(own[1], s2), s0 = fmaxf(own[0], s1);
                    const float p0 = nxt[0], p1 = fmaxf(p0, nxt[1]), p2 = fmaxf(p1, nxt[2]), p3 = fmaxf(p2, nxt[3]);
                    const float o[4] = {fmaxf(fmaxf(s0, p0), 0.f), fmaxf(fmaxf(s1, p1), 0.f), fmaxf(fmaxf(s2, p2), 0.f),
                                        fmaxf(fmaxf(s3, p3), 0.f)};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int txo = 8 * j + 4 * half + q;          // pooled column inside the tile
                        if (yy < H && txo < C2_TC - 4 && tx0 + txo < Wo) dst[static_cast<long>(8 * j + q) * N] = o[q];
                    }
                }
            }
        }
    } else if constexpr (MODE == MODE_POOL_NCHW) {
        // conv + per-column affine, then MaxPool2d(2, 2) + ReLU (models.py:34,55-56) written in the reference's
        // memory order [clip][chan][H/2][W/2], which the model re-reads as [clip][H/2][chan * W/2] (the reshape at
        // models.py:59-61 is a reinterpretation, not a permute).  A pooling window is four consecutive accumulator
        // registers of a lane.
        static_assert(PH == 2 && PW == 2, "2x2 pooling");
        const int Ho = H / 2, Wo = W / 2;
        float *yc = y + clip * static_cast<long>(N) * Ho * Wo;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                int ty, tx;
                tile_pos<PH, PW>(PWV * wm + 32 * mt + acc_row(4 * w, half), ty, tx);
                const int yo = (ty0 + ty) / 2, xo = (tx0 + tx) / 2;
                if (yo < Ho && xo < Wo) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float mx = fmaxf(fmaxf(acc[mt][nt][4 * w], acc[mt][nt][4 * w + 1]),
                                               fmaxf(acc[mt][nt][4 * w + 2], acc[mt][nt][4 * w + 3]));
                        yc[(static_cast<long>(colof(nt)) * Ho + yo) * Wo + xo] = fmaxf(mx, 0.f);
                    }
                }
            }
    } else if constexpr (MODE == MODE_GLU) {
        // An m-tile is one row of the tile and a lane's 16 registers are 16 of its columns: one offset per m-tile and
        // lane, the registers' distances are immediates.  Rows below the image fall out of the descriptor's range; the
        // width is a whole number of tiles (checked by the host).
        static_assert(PH == 1 && PW == 1, "the GLU kernels do not pool");
        float *yc = y + clip * static_cast<long>(H) * W * CO;
        const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(yc, 0, H * W * CO * 4, 0x00020000);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const unsigned base =
                static_cast<unsigned>(((ty0 + MT * wm + mt) * W + tx0 + 4 * half) * CO + col32 + 32 * wn) * 4u;
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int nt = 0; nt < NT / 2; ++nt) {
                    const float val =
                        acc[mt][nt][r] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[mt][nt + NT / 2][r]));
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), yrs,
                                                          static_cast<int>(base + (acc_row(r, 0) * CO + 32 * WN * nt) * 4), 0, 0);
                }
        }
    } else {
        static_assert(CO == 16 || CO % (32 * WN) == 0, "channels per wave");
        constexpr int NTC = CO >= 32 ? CO / 32 / WN : 1;   // channel tiles per wave after mixing the branches
        constexpr int WIN = PH * PW;
        static_assert((PH == 2 && PW == 2) || (PH == 1 && PW == 2), "pooling windows are runs of 4 or 2 registers");
        const int Ho = H / PH, Wo = W / PW;
        float *yc = y + clip * static_cast<long>(Ho) * Wo * CO;
        const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(yc, 0, Ho * Wo * CO * 4, 0x00020000);
        if constexpr (CO == 16) {
            // One 32-column tile holds both branches: channel c is column c (branch 1) and column c + 16 (branch
            // 2), i.e. lanes L and L + 16 hold the two branch values of the same 16 positions.  The pair shares the
            // work: the low lane finishes the positions of registers 0-7 (two pooling windows), the high lane those of
            // registers 8-15.  One row swap per register pair (k, k + 8) leaves branch 1 of the lane's position in
            // the first register and branch 2 in the second, in both lanes.
            static_assert(G::NCHUNK == 1 && WN == 1 && PH == 2 && PW == 2, "the 16-channel gate assumes one staged chunk and 2x2 pooling");
            const int hi16 = col32 >> 4, ch = col32 & 15;
            const float b2s = s2[ch], b2t = t2[ch];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float z[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float b1 = acc[mt][0][k], b2 = acc[mt][0][k + 8];
                    swap_rows16(b1, b2);
                    z[k] = b1 + b2;
                }
                // position of value k: pb + (k & 3) + 8 * (k >> 2) with pb a multiple of 4 (acc_row(k + 8) = acc_row(k) + 16)
                int ty, tx;
                tile_pos<PH, PW>(PWV * wm + 32 * mt + 4 * half + 16 * hi16, ty, tx);
                const unsigned obase = static_cast<unsigned>((((ty0 + ty) / PH) * Wo + (tx0 + tx) / PW) * CO + ch) * 4u;
#pragma unroll
                for (int win = 0; win < 2; ++win) {
                    float sum = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float gate = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z[4 * win + q]));
                        sum += fmaxf(fmaf(ucv[mt * 8 + 4 * win + q] * gate, b2s, b2t), 0.f);
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sum * 0.25f), yrs,
                                                          static_cast<int>(obase + d_xo(4 * win) * CO * 4), 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < NTC; ++nt) {
                const int ch = colof(nt);
                const float b2s = s2[ch], b2t = t2[ch];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    int ty, tx;                                   // the lane's register 0
                    tile_pos<PH, PW>(PWV * wm + 32 * mt + 4 * half, ty, tx);
                    const unsigned obase = static_cast<unsigned>((((ty0 + ty) / PH) * Wo + (tx0 + tx) / PW) * CO + ch) * 4u;
                    // the centre input: requested before the main loop when the kernel stages one chunk, else here
                    float ucl[16];
                    if constexpr (!UC_EARLY) {
                        const unsigned gbase = static_cast<unsigned>(((ty0 + ty) * W + tx0 + tx) * CIN + ch) * 4u;
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            ucl[r] = __builtin_bit_cast(
                                float, __builtin_amdgcn_raw_buffer_load_b32(
                                           xrs, static_cast<int>(gbase + d_ty(r) * rowb + d_tx(r) * CIN * 4), 0, 0));
                    }
                    float sum = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float gate = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[mt][nt][r] + acc[mt][nt + NTC][r]));
                        const float uc = UC_EARLY ? ucv[(mt * UC_NT + nt) * 16 + r] : ucl[r];
                        sum += fmaxf(fmaf(uc * gate, b2s, b2t), 0.f);
                        if ((r % WIN) == WIN - 1) {
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sum * (1.0f / WIN)), yrs,
                                                                  static_cast<int>(obase + d_xo(r) * CO * 4), 0, 0);
                            sum = 0.f;
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// C[M][N] = act((A[M][K] . B) * oscale + bias), act = identity or ReLU, split-f16; B packed [K/16][2][N][16]; N % 128 == 0, K % KC == 0.
// A workgroup owns 64*MT rows x 128 columns; DB = two sets of A planes in LDS (one barrier per K chunk).
// ---------------------------------------------------------------------------
template <int MT, int KC, bool DB>
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(const float *__restrict__ A, long lda,
                                                           const uint4 *__restrict__ bpk, int N, int K,
                                                           const float *__restrict__ bias, float oscale,
                                                           float *__restrict__ C, long ldc, long M, int relu,
                                                           unsigned m_tiles, unsigned n_tiles) {
    constexpr int TM = 64 * MT, PSB = KC * 2 + 16, PLANE = TM * PSB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *plane_h = smem, *plane_l = smem + PLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, col32 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    // Tile order.  Workgroup b runs on XCD b % 8, so XCD x takes the contiguous logical range
    // [x * per_xcd, (x + 1) * per_xcd); inside it, tiles go in groups of 8 row blocks x all column
    // blocks, row block fastest: the ~64 tiles an XCD works on at one time form an 8 x 8 patch whose
    // A and B panels of the current K chunk (1 MB) are shared through its L2 instead of each tile
    // streaming its own panels from HBM.
    const unsigned per_xcd = (gridDim.x + 7) / 8;
    const unsigned logical = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (logical >= m_tiles * n_tiles) return;
    constexpr unsigned GM = 8;
    const unsigned group = logical / (GM * n_tiles), in_group = logical % (GM * n_tiles);
    const unsigned rows_in_group = min(GM, m_tiles - group * GM);
    const unsigned tile_m = group * GM + in_group % rows_in_group, tile_n = in_group / rows_in_group;
    const long m0 = static_cast<long>(tile_m) * TM;
    const int ncol = tile_n * 128 + wn * 64 + col32;
    f32x16 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    const char *ah[MT], *al[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int off = (32 * MT * wm + 32 * mt + col32) * PSB + 16 * half;
        ah[mt] = plane_h + off;
        al[mt] = plane_l + off;
    }
    const uint4 *bp = bpk + ncol * 2 + half;
    constexpr int NLD = TM * (KC / 4) / 256, RPP = 256 / (KC / 4);     // loads per thread and chunk; rows per pass
    // Rows are addressed through buffer descriptors that span exactly this tile's valid rows of A and C: rows past
    // M read as zero and their stores are dropped by the hardware.  Only the per-lane offset is range-checked, so
    // everything that selects a row stays in it (the asm keeps the compiler from moving a uniform part out).
    const int rows_valid = static_cast<int>(min(static_cast<long>(TM), M - m0));
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(A + m0 * lda), 0, rows_valid * static_cast<int>(lda) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t crs =
        __builtin_amdgcn_make_buffer_rsrc(C + m0 * ldc, 0, rows_valid * static_cast<int>(ldc) * 4, 0x00020000);
    using f32x4v = __attribute__((ext_vector_type(4))) float;
    const unsigned a0 = ((tid / (KC / 4)) * static_cast<unsigned>(lda) + (tid % (KC / 4)) * 4) * 4u;
    const unsigned astep = RPP * static_cast<unsigned>(lda) * 4u;
    f32x4v v[NLD];
    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            unsigned o = a0 + j * astep;
            asm volatile("" : "+v"(o));
            v[j] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(ars, static_cast<int>(o), k0 * 4, 0));
        }
    };
    auto stage_chunk = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + 256 * j, r = i / (KC / 4), c = (i - r * (KC / 4)) * 4;
            uint2 hi, lo;
            split2_f16(v[j][0], v[j][1], hi.x, lo.x);
            split2_f16(v[j][2], v[j][3], hi.y, lo.y);
            *reinterpret_cast<uint2 *>(plane_h + buf * 2 * PLANE + r * PSB + c * 2) = hi;
            *reinterpret_cast<uint2 *>(plane_l + buf * 2 * PLANE + r * PSB + c * 2) = lo;
        }
    };
    if constexpr (DB) {
        // Two sets of planes: the rows of chunk k+1 are requested before the MFMAs of chunk k and written to the
        // other set after them, so a chunk costs one barrier and its load latency hides behind the GEMM.
        load_chunk(0);
        stage_chunk(0);
        lds_barrier();
        for (int k0 = 0; k0 < K; k0 += 2 * KC) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int kk = k0 + b * KC;
                if (kk < K) {
                    if (kk + KC < K) load_chunk(kk + KC);
                    split_gemm<MT, 2, KC / 16>(acc, ah, al, bp + static_cast<long>(kk / 16) * (N * 4), N,
                                                [b](int i) { return b * 2 * PLANE + i * 32; });
                    if (kk + KC < K) stage_chunk(b ^ 1);
                    lds_barrier();
                }
            }
        }
    } else {
        for (int k0 = 0; k0 < K; k0 += KC) {
            load_chunk(k0);
            if (k0) __syncthreads();
            stage_chunk(0);
            __syncthreads();
            split_gemm<MT, 2, KC / 16>(acc, ah, al, bp + static_cast<long>(k0 / 16) * (N * 4), N, [](int i) { return i * 32; });
        }
    }
    float bv[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) bv[nt] = bias ? bias[ncol + 32 * nt] : 0.f;
    const unsigned c0 = ((32 * MT * wm + 4 * half) * static_cast<unsigned>(ldc) + ncol) * 4u;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // acc_row(r, half) = (r & 3) + 8 * (r >> 2) + 4 * half
            unsigned so = c0 + (32 * mt + (r & 3) + 8 * (r >> 2)) * static_cast<unsigned>(ldc) * 4u;
            asm volatile("" : "+v"(so));
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float o = fmaf(acc[mt][nt][r], oscale, bv[nt]);
                if (relu) o = fmaxf(o, 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), crs, static_cast<int>(so + nt * 128), 0, 0);
            }
        }
}

// ---------------------------------------------------------------------------
// one GRU step, H = 1024.  grid (32 unit groups, clip tiles of 32, 2 directions)
// ---------------------------------------------------------------------------
constexpr int GB_H = 1024;

__global__ __launch_bounds__(256) void gru_big_step_kernel(
    const float *__restrict__ gi,       // [clip][T][2][3][1024]: W_ih x + b_ih (+ b_hh for r, z)
    const uint4 *__restrict__ whh,      // packed split [2][1024/16][2][3072][16], columns [unit group][gate][32]
    float whh_oscale, const float *__restrict__ bhn,   // [2][1024]
    const float *__restrict__ h_prev, float *__restrict__ h_next,   // [2][clips][1024]
    float *__restrict__ out,            // [clip][T][2048]
    long n_clips, int T, int step) {
    constexpr int KW = GB_H / 4;                      // K slice per wave
    constexpr int PSB = KW * 2 + 16, PLANE = 32 * PSB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, col32 = lane & 31;
    const int ug = blockIdx.x, dir = blockIdx.z;
    const long b0 = static_cast<long>(blockIdx.y) * 32;
    const int t = dir ? T - 1 - step : step;
    char *plane_h = smem + wave * 2 * PLANE, *plane_l = plane_h + PLANE;
    const float *hp = h_prev + (static_cast<long>(dir) * n_clips) * GB_H;

    // The gate inputs of this thread's four (clip, unit) elements are requested first: gi streams from HBM, and
    // asked for only after the GEMM its latency would sit at the end of every step.
    float pg[4][3], ph_prev[4], pb;
    {
        const int jj = tid & 31, j = ug * 32 + jj;
        pb = bhn[dir * GB_H + j];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long b = b0 + (tid >> 5) + 8 * q;
            const bool ok = b < n_clips;
            const float *gp = gi + (((ok ? b : 0) * T + t) * 2 + dir) * (3L * GB_H) + j;
#pragma unroll
            for (int g = 0; g < 3; ++g) pg[q][g] = ok ? gp[g * GB_H] : 0.f;
            ph_prev[q] = ok ? hp[b * GB_H + j] : 0.f;
        }
    }

    const uint4 *bp = whh + (static_cast<long>(dir) * (GB_H / 16) + wave * (KW / 16)) * (3 * GB_H * 4) +
                      (ug * 96 + col32) * 2 + half;
    // each wave stages its own K slice of h for the 32 clips
    constexpr int NLD = 32 * (KW / 4) / 64;
    float4 v[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int i = lane + 64 * j, r = i / (KW / 4), c = (i - r * (KW / 4)) * 4;
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b0 + r < n_clips) v[j] = *reinterpret_cast<const float4 *>(hp + (b0 + r) * GB_H + wave * KW + c);
    }
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int i = lane + 64 * j, r = i / (KW / 4), c = (i - r * (KW / 4)) * 4;
        uint2 hi, lo;
        split2_f16(v[j].x, v[j].y, hi.x, lo.x);
        split2_f16(v[j].z, v[j].w, hi.y, lo.y);
        *reinterpret_cast<uint2 *>(plane_h + r * PSB + c * 2) = hi;
        *reinterpret_cast<uint2 *>(plane_l + r * PSB + c * 2) = lo;
    }
    __syncthreads();

    f32x16 acc[1][3];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][g][r] = 0.f;
    const char *ah[1] = {plane_h + col32 * PSB + 16 * half}, *al[1] = {plane_l + col32 * PSB + 16 * half};
    split_gemm<1, 3, KW / 16>(acc, ah, al, bp, 3 * GB_H, [](int i) { return i * 32; });

    // reduce the four K slices through LDS, then the gate math: thread -> (clip row, unit)
    __syncthreads();
    float *red = reinterpret_cast<float *>(smem);     // [wave][gate][row][unit] = 4*3*32*32 floats = 48 KB
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * 3 + g) * 32 + acc_row(r, half)) * 32 + col32] = acc[0][g][r];
    __syncthreads();
    float *hn = h_next + (static_cast<long>(dir) * n_clips) * GB_H;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = (tid >> 5) + 8 * q, jj = tid & 31;
        const long b = b0 + row;
        const int j = ug * 32 + jj;
        float gh[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) s += red[((w * 3 + g) * 32 + row) * 32 + jj];
            gh[g] = s * whh_oscale;
        }
        const float rg = sigmoidf_fast(pg[q][0] + gh[0]);
        const float zg = sigmoidf_fast(pg[q][1] + gh[1]);
        const float ng = tanhf_fast(fmaf(rg, gh[2] + pb, pg[q][2]));
        const float hv = fmaf(zg, ph_prev[q] - ng, ng);
        if (b < n_clips) {
            hn[b * GB_H + j] = hv;
            out[(b * T + t) * (2L * GB_H) + dir * GB_H + j] = hv;
        }
    }
}

// ---------------------------------------------------------------------------
// The whole H = 1024 recurrence in ONE launch (small batches, long sequences: minutes-long recordings).
//
// grid (64 unit groups, 2 directions, n_sets clip sets), 512 threads, one workgroup per CU.  A workgroup owns 16
// hidden units x 3 gates of one direction over the whole K = 1024 and keeps that slice of W_hh in registers for the
// lifetime of the kernel as f16 (hi, lo) B fragments: wave w holds K slice [128 w, 128 w + 128) (96 VGPRs).  Clips
// come in groups of 16 (= the M of v_mfma_f32_16x16x32_f16); a workgroup serves the groups of its clip set one
// after the other inside every step, so that one group's exchange latency passes while the next group computes.
//
// Per step and group: every workgroup of the direction needs ALL of h_{t-1} (16 clips x 1024) and produces 16
// units of h_t.  h travels between workgroups through a double-buffered exchange array in device memory, already
// split into f16 hi / lo planes and laid out [producer workgroup][clip][unit] so that a consumer lane reads its
// MFMA A fragment (8 consecutive k of one clip) straight into registers with one 16-byte load: no LDS staging and
// no workgroup barrier in front of the matrix product.  The hand-off is flag based, independent of dispatch
// order, timing and XCD placement:
//   producer: wave 7 alone stores the workgroup's 1 KB (16-byte write-through sc1 stores), drains them
//             (s_waitcnt vmcnt(0)), then its lane 0 stores flag = step + 1 (sc1);
//   consumer: each wave polls the 8 flags of the 8 producers of ITS K slice (relaxed sc1 loads, s_sleep between
//             polls), and only after its own poll has matched issues its sc1 payload loads.
// Buffer parity: step s writes buffer s & 1 and reads (s - 1) & 1; a workgroup can only write buffer s & 1 at step s
// after every flag has reached s, i.e. after every reader of step s - 1 (which read that buffer) has published.
// The K slices are reduced across the eight waves through LDS; threads 0..255 then do the gate arithmetic for
// one (clip, unit) each, keep their h in a register for the next step, write the output row and stage the f16
// halves for wave 7.  gi of the next step is requested a step ahead (after the A fragment loads, so that its HBM
// latency is never in front of them in the in-order return queue).
// Every spin is bounded: a wave that gives up raises *status and stops waiting, so the grid always drains; the
// head kernel that follows turns a raised status into NaN logits.
// ---------------------------------------------------------------------------
constexpr int GP_U = 16;                       // hidden units per workgroup
constexpr int GP_WG = GB_H / GP_U;             // 64 workgroups per direction and clip set
constexpr int GP_MAX_G = 3;                    // groups per workgroup (four 32-clip groups would spill registers)
constexpr int GP_MAX_MT = 2;                   // 16-clip MFMA tiles per group
constexpr unsigned GP_SPIN_LIMIT = 3000000u;   // polls a wave makes before it gives up (~ seconds); default of the spin_limit argument
constexpr int GP_LDS_BYTES = 96 * 1024;        // more than half of a CU's LDS: never two workgroups on one CU
constexpr int GP_STATUS_WORDS = 64;            // status word + padding in front of the flags

using gu32 = __attribute__((address_space(1))) unsigned;
using u32x4v = __attribute__((ext_vector_type(4))) unsigned;
using f32x4g = __attribute__((ext_vector_type(4))) float;

// exchange array, in uint4 (16-byte) units: [parity][dir][group][plane][producer workgroup][16 mt clips][16 units] f16
__host__ __device__ constexpr long gp_hx_index(int mt_tiles, int par, int dir, int n_groups, int gg, int plane, int wg) {
    return ((((static_cast<long>(par) * 2 + dir) * n_groups + gg) * 2 + plane) * GP_WG + wg) * (32 * mt_tiles);
}

// G: groups a workgroup serves per step; MT: 16-clip tiles per group (a group = 16 MT clips)
template <int G, int MT>
__global__ __launch_bounds__(512) void gru_big_persistent_kernel(
    const float *__restrict__ gi,       // [clip][T][2][3][1024]: W_ih x + b_ih (+ b_hh for r, z)
    const uint4 *__restrict__ whh,      // packed split [2][1024/16][2][3072][16], columns [unit group of 32][gate][32]
    float whh_oscale, const float *__restrict__ bhn,   // [2][1024]
    float *__restrict__ out,            // [clip][T][2048]
    uint4 *hx,                          // exchange array (gp_hx_index)
    unsigned *sync_words,               // [GP_STATUS_WORDS] status, then flags [2 dir][n_groups][64]; zeroed before the launch
    long n_clips, int T, int n_groups, int n_sets, unsigned spin_limit) {
    constexpr int CLIPS = 16 * MT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *red = reinterpret_cast<float *>(smem);                       // [wave][gate][mt][clip 16][unit 16]
    char *stage = smem + 8 * 3 * MT * 256 * 4;                          // [plane][clip 16 MT][unit 16] f16
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, kg = lane >> 4;
    const int wg = blockIdx.x, dir = blockIdx.y, set = blockIdx.z;
    gu32 *status = (gu32 *)sync_words;
    gu32 *flags = (gu32 *)(sync_words + GP_STATUS_WORDS) + static_cast<long>(dir) * n_groups * GP_WG;

    // this wave's slice of W_hh: B fragment of k-step 4 wave + s, column = unit 16 wg + n16 of gate q
    f16x8 wh[3][4], wl[3][4];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int ks16 = 2 * (4 * wave + s4) + (kg >> 1);
            const int col = (wg >> 1) * 96 + q * 32 + (wg & 1) * 16 + n16;
            const uint4 *p = whh + ((static_cast<long>(dir) * (GB_H / 16) + ks16) * 2 * (3 * GB_H) + col) * 2 + (kg & 1);
            wh[q][s4] = __builtin_bit_cast(f16x8, p[0]);
            wl[q][s4] = __builtin_bit_cast(f16x8, p[2 * 3 * GB_H]);
        }

    // gate role of threads 0..255: unit n of the workgroup, clips c + 16 mt of the group
    const bool gate_thread = tid < 256;
    const int gc = (tid >> 4) & 15, gn = tid & 15, j = wg * GP_U + gn;
    const float b_hn = bhn[dir * GB_H + j];
    float hprev[G][MT], gcur[G][MT][3], gnext[G][MT][3];
    int clip_of[G][MT];                  // -1: no such clip (addresses are formed per use: pointers kept per
    bool live[G][MT];                    // (group, tile) would cost 8 registers each)
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int gg = set + n_sets * g;
            const long b = static_cast<long>(gg) * CLIPS + 16 * mt + gc;
            live[g][mt] = gate_thread && gg < n_groups && b < n_clips;
            clip_of[g][mt] = live[g][mt] ? static_cast<int>(b) : -1;
            hprev[g][mt] = 0.f;
            const long t0 = dir ? T - 1 : 0;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                gnext[g][mt][q] = live[g][mt] ? gi[((b * T + t0) * 2 + dir) * (3L * GB_H) + q * GB_H + j] : 0.f;
                gcur[g][mt][q] = 0.f;
            }
        }
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(
        hx, 0, static_cast<int>(gp_hx_index(MT, 2, 0, n_groups, 0, 0, 0) * 16), 0x00020000);
    bool gave_up = false;

#pragma unroll 1
    for (int s = 0; s < T; ++s) {
        const long t = dir ? T - 1 - s : s, tn = dir ? T - 2 - s : s + 1;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int gg = set + n_sets * g;
            if (gg >= n_groups) continue;                     // uniform over the workgroup
            // this step's gate inputs were requested a step ago (behind that step's A fragments); taking them over
            // here, in front of the poll, is where waiting for them costs nothing
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int q = 0; q < 3; ++q) gcur[g][mt][q] = gnext[g][mt][q];
            f32x4g acc[MT][3];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int q = 0; q < 3; ++q) acc[mt][q] = f32x4g{0.f, 0.f, 0.f, 0.f};
            if (s > 0) {
                // wait for the 8 producers of this wave's K slice
                gu32 *fp = flags + static_cast<long>(gg) * GP_WG + 8 * wave + (lane & 7);
                if (!gave_up) {
                    unsigned spins = 0;
                    while (true) {
                        const unsigned f = __hip_atomic_load(fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (__all(f >= static_cast<unsigned>(s))) break;
                        if (++spins > spin_limit) {
                            gave_up = true;
                            if (lane == 0) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");       // no instruction: keeps the loads below the poll
                // A fragments of h_{t-1}: row = clip 16 mt + n16, k = 128 wave + 32 s4 + 8 kg .. + 7 -> producer
                // workgroup 8 wave + 2 s4 + kg / 2, its units 8 (kg & 1) .. + 7
                f16x8 ah[MT][4], al[MT][4];
                const int par = (s - 1) & 1;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const int wgp = 8 * wave + 2 * s4 + (kg >> 1);
                        const int in_block = (16 * mt + n16) * 2 + (kg & 1);
                        const int o_h = static_cast<int>((gp_hx_index(MT, par, dir, n_groups, gg, 0, wgp) + in_block) * 16);
                        const int o_l = static_cast<int>((gp_hx_index(MT, par, dir, n_groups, gg, 1, wgp) + in_block) * 16);
                        ah[mt][s4] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(hrs, o_h, 0, 16));
                        al[mt][s4] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(hrs, o_l, 0, 16));
                    }
                // all requests go out before the first product: left alone, the scheduler reuses one register
                // quad for every fragment and pays the round trips one after the other
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
                        for (int q = 0; q < 3; ++q)
                            acc[mt][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt][s4], wh[q][s4], acc[mt][q], 0, 0, 0);
#pragma unroll
                        for (int q = 0; q < 3; ++q)
                            acc[mt][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt][s4], wl[q][s4], acc[mt][q], 0, 0, 0);
#pragma unroll
                        for (int q = 0; q < 3; ++q)
                            acc[mt][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt][s4], wh[q][s4], acc[mt][q], 0, 0, 0);
                    }
            }
            // the next step's gate inputs, requested behind the A fragments
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                if (live[g][mt] && s + 1 < T) {
                    const float *gp = gi + ((static_cast<long>(clip_of[g][mt]) * T + tn) * 2 + dir) * (3L * GB_H) + j;
#pragma unroll
                    for (int q = 0; q < 3; ++q) gnext[g][mt][q] = gp[q * GB_H];
                }
            // partial sums of this wave's K slice: accumulator register i = clip 16 mt + 4 kg + i, unit n16
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        red[(((wave * 3 + q) * MT + mt) * 16 + 4 * kg + i) * 16 + n16] = acc[mt][q][i];
            __syncthreads();
            if (gate_thread) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    float gh[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        float sum = 0.f;
#pragma unroll
                        for (int w = 0; w < 8; ++w) sum += red[((w * 3 + q) * MT + mt) * 256 + tid];
                        gh[q] = sum * whh_oscale;
                    }
                    const float rg = sigmoidf_fast(gcur[g][mt][0] + gh[0]);
                    const float zg = sigmoidf_fast(gcur[g][mt][1] + gh[1]);
                    const float ng = tanhf_fast(fmaf(rg, gh[2] + b_hn, gcur[g][mt][2]));
                    const float hv = fmaf(zg, hprev[g][mt] - ng, ng);             // (1 - z) * n + z * h
                    hprev[g][mt] = hv;
                    if (live[g][mt]) out[(static_cast<long>(clip_of[g][mt]) * T + t) * (2L * GB_H) + dir * GB_H + j] = hv;
                    _Float16 hi, lo;
                    split_f16(hv, hi, lo);
                    *reinterpret_cast<_Float16 *>(stage + (mt * 256 + tid) * 2) = hi;
                    *reinterpret_cast<_Float16 *>(stage + (MT * 256 + mt * 256 + tid) * 2) = lo;
                }
            }
            __syncthreads();
            if (wave == 7 && s + 1 < T) {
                // publish: both planes of this workgroup's [16 MT clips][16 units], 16 bytes per lane and store
#pragma unroll
                for (int r = 0; r < MT; ++r) {
                    const int u = lane + 64 * r, plane = u / (32 * MT), idx = u % (32 * MT);
                    const uint4 v = *reinterpret_cast<const uint4 *>(stage + u * 16);
                    const int o = static_cast<int>((gp_hx_index(MT, s & 1, dir, n_groups, gg, plane, wg) + idx) * 16);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), hrs, o, 0, 16);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0)
                    __hip_atomic_store(flags + static_cast<long>(gg) * GP_WG + wg, static_cast<unsigned>(s + 1), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// The same recurrence with the step number carried INSIDE the exchanged data instead of in separate flags.
// An element of h travels as one 32-bit word: f16 hi half in the low 16 bits, f16 lo half in the high 16 bits,
// whose least significant bit (2^-21 of h: below what the hi/lo pair resolves) is replaced by a tag = bit 1 of the
// step number.  A buffer (parity s & 1) is rewritten every second step, so consecutive contents of a word carry
// alternating tags; the array starts out as all ones (tag 1: the first write, step 0 or 1, carries tag 0).  Every
// 32-bit word carries its own tag, so a reader can never take a half-updated 16-byte piece for a complete one.
// A consumer simply loads its A-fragment words (sc1) until every tag shows the step it waits for: the payload
// loads ARE the poll, and the producer side has no drain and no flag store (one memory round trip less on either
// side of the hand-off).  The gate threads publish directly: a quad of lanes gathers its four words (DPP) and
// its first lane stores them as one 16-byte write-through store, so there is one barrier per step and group (the
// reduction buffer in LDS is double buffered).
// exchange array, in uint4 units: [parity][dir][group][producer workgroup][clip 16 MT][4 quads of units]
__host__ __device__ constexpr long gt_hx_index(int mt_tiles, int par, int dir, int n_groups, int gg, int wg) {
    return (((static_cast<long>(par) * 2 + dir) * n_groups + gg) * GP_WG + wg) * (64 * mt_tiles);
}

template <int G, int MT>
__global__ __launch_bounds__(512) void gru_big_tagged_kernel(
    const float *__restrict__ gi, const uint4 *__restrict__ whh, float whh_oscale, const float *__restrict__ bhn,
    float *__restrict__ out,
    uint4 *hx,                          // exchange array (gt_hx_index), every byte 0xFF before the launch
    unsigned *sync_words,               // [0]: status, zero before the launch
    long n_clips, int T, int n_groups, int n_sets, unsigned spin_limit) {
    constexpr int CLIPS = 16 * MT;
    constexpr int RED_FLOATS = 8 * 3 * MT * 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *red_all = reinterpret_cast<float *>(smem);                   // [2][wave][gate][mt][clip 16][unit 16]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, kg = lane >> 4;
    const int wg = blockIdx.x, dir = blockIdx.y, set = blockIdx.z;
    gu32 *status = (gu32 *)sync_words;

    f16x8 wh[3][4], wl[3][4];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int ks16 = 2 * (4 * wave + s4) + (kg >> 1);
            const int col = (wg >> 1) * 96 + q * 32 + (wg & 1) * 16 + n16;
            const uint4 *p = whh + ((static_cast<long>(dir) * (GB_H / 16) + ks16) * 2 * (3 * GB_H) + col) * 2 + (kg & 1);
            wh[q][s4] = __builtin_bit_cast(f16x8, p[0]);
            wl[q][s4] = __builtin_bit_cast(f16x8, p[2 * 3 * GB_H]);
        }

    const bool gate_thread = tid < 256;
    const int gc = (tid >> 4) & 15, gn = tid & 15, j = wg * GP_U + gn;
    const float b_hn = bhn[dir * GB_H + j];
    float hprev[G][MT], gcur[G][MT][3], gnext[G][MT][3];
    int clip_of[G][MT];
    bool live[G][MT];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int gg = set + n_sets * g;
            const long b = static_cast<long>(gg) * CLIPS + 16 * mt + gc;
            live[g][mt] = gate_thread && gg < n_groups && b < n_clips;
            clip_of[g][mt] = live[g][mt] ? static_cast<int>(b) : -1;
            hprev[g][mt] = 0.f;
            const long t0 = dir ? T - 1 : 0;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                gnext[g][mt][q] = live[g][mt] ? gi[((b * T + t0) * 2 + dir) * (3L * GB_H) + q * GB_H + j] : 0.f;
                gcur[g][mt][q] = 0.f;
            }
        }
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(
        hx, 0, static_cast<int>(gt_hx_index(MT, 2, 0, n_groups, 0, 0) * 16), 0x00020000);
    bool gave_up = false;
    int it = 0;                                               // (step, group) iterations done: selects the LDS buffer

#pragma unroll 1
    for (int s = 0; s < T; ++s) {
        const long t = dir ? T - 1 - s : s, tn = dir ? T - 2 - s : s + 1;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int gg = set + n_sets * g;
            if (gg >= n_groups) continue;                     // uniform over the workgroup
            float *red = red_all + (it & 1) * RED_FLOATS;
            ++it;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int q = 0; q < 3; ++q) gcur[g][mt][q] = gnext[g][mt][q];
            f32x4g acc[MT][3];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int q = 0; q < 3; ++q) acc[mt][q] = f32x4g{0.f, 0.f, 0.f, 0.f};
            if (s > 0) {
                // this lane's words of h_{t-1}: row = clip 16 mt + n16, k = 128 wave + 32 s4 + 8 kg .. + 7 -> producer
                // workgroup 8 wave + 2 s4 + kg / 2, its quads 2 (kg & 1) and 2 (kg & 1) + 1: 32 contiguous bytes
                const int par = (s - 1) & 1;
                const unsigned want = ((static_cast<unsigned>(s - 1) >> 1) & 1u) ? 0x00010000u : 0u;
                u32x4v wa[MT][4], wb[MT][4];
                unsigned spins = 0;
                if (MT > 1) {
                    // large groups: wait on ONE piece per lane first (the last one its producers write), so that the
                    // waiting costs the memory system 16 bytes per lane and poll instead of 256
                    const int wgp = 8 * wave + 2 * 3 + (kg >> 1);
                    const int o = static_cast<int>((gt_hx_index(MT, par, dir, n_groups, gg, wgp) + (16 * (MT - 1) + n16) * 4 +
                                                    2 * (kg & 1) + 1) * 16);
                    while (!gave_up) {
                        const u32x4v probe = __builtin_amdgcn_raw_buffer_load_b128(hrs, o, 0, 16);
                        const unsigned pa = probe[0] & probe[1] & probe[2] & probe[3], po = probe[0] | probe[1] | probe[2] | probe[3];
                        const bool ok = want ? (pa & 0x00010000u) != 0u : (po & 0x00010000u) == 0u;
                        if (__all(ok)) break;
                        if (++spins > spin_limit) {
                            gave_up = true;
                            if (lane == 0) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                while (true) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) {
                            const int wgp = 8 * wave + 2 * s4 + (kg >> 1);
                            const int o = static_cast<int>((gt_hx_index(MT, par, dir, n_groups, gg, wgp) + (16 * mt + n16) * 4 +
                                                            2 * (kg & 1)) * 16);
                            wa[mt][s4] = __builtin_amdgcn_raw_buffer_load_b128(hrs, o, 0, 16);
                            wb[mt][s4] = __builtin_amdgcn_raw_buffer_load_b128(hrs, o + 16, 0, 16);
                        }
                    unsigned all_and = 0xffffffffu, all_or = 0u;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                all_and &= wa[mt][s4][e] & wb[mt][s4][e];
                                all_or |= wa[mt][s4][e] | wb[mt][s4][e];
                            }
                    const bool ok = want ? (all_and & 0x00010000u) != 0u : (all_or & 0x00010000u) == 0u;
                    if (__all(ok) || gave_up) break;
                    if (++spins > spin_limit) {
                        gave_up = true;
                        if (lane == 0) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        // de-interleave: word = hi | lo << 16 -> fragment words (k, k + 1) of the hi and lo planes
                        u32x4v fh, fl;
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            fh[e] = __builtin_amdgcn_perm(wa[mt][s4][2 * e + 1], wa[mt][s4][2 * e], 0x05040100u);
                            fl[e] = __builtin_amdgcn_perm(wa[mt][s4][2 * e + 1], wa[mt][s4][2 * e], 0x07060302u) & 0xfffefffeu;
                            fh[2 + e] = __builtin_amdgcn_perm(wb[mt][s4][2 * e + 1], wb[mt][s4][2 * e], 0x05040100u);
                            fl[2 + e] = __builtin_amdgcn_perm(wb[mt][s4][2 * e + 1], wb[mt][s4][2 * e], 0x07060302u) & 0xfffefffeu;
                        }
                        const f16x8 ah = __builtin_bit_cast(f16x8, fh), al = __builtin_bit_cast(f16x8, fl);
#pragma unroll
                        for (int q = 0; q < 3; ++q)
                            acc[mt][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh[q][s4], acc[mt][q], 0, 0, 0);
#pragma unroll
                        for (int q = 0; q < 3; ++q)
                            acc[mt][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl[q][s4], acc[mt][q], 0, 0, 0);
#pragma unroll
                        for (int q = 0; q < 3; ++q)
                            acc[mt][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh[q][s4], acc[mt][q], 0, 0, 0);
                    }
            }
            // the next step's gate inputs, requested behind the exchange loads
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                if (live[g][mt] && s + 1 < T) {
                    const float *gp = gi + ((static_cast<long>(clip_of[g][mt]) * T + tn) * 2 + dir) * (3L * GB_H) + j;
#pragma unroll
                    for (int q = 0; q < 3; ++q) gnext[g][mt][q] = gp[q * GB_H];
                }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        red[(((wave * 3 + q) * MT + mt) * 16 + 4 * kg + i) * 16 + n16] = acc[mt][q][i];
            __syncthreads();
            if (gate_thread) {
                const unsigned tag = ((static_cast<unsigned>(s) >> 1) & 1u) << 16;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    float gh[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        float sum = 0.f;
#pragma unroll
                        for (int w = 0; w < 8; ++w) sum += red[((w * 3 + q) * MT + mt) * 256 + tid];
                        gh[q] = sum * whh_oscale;
                    }
                    const float rg = sigmoidf_fast(gcur[g][mt][0] + gh[0]);
                    const float zg = sigmoidf_fast(gcur[g][mt][1] + gh[1]);
                    const float ng = tanhf_fast(fmaf(rg, gh[2] + b_hn, gcur[g][mt][2]));
                    const float hv = fmaf(zg, hprev[g][mt] - ng, ng);             // (1 - z) * n + z * h
                    hprev[g][mt] = hv;
                    if (live[g][mt]) out[(static_cast<long>(clip_of[g][mt]) * T + t) * (2L * GB_H) + dir * GB_H + j] = hv;
                    if (s + 1 < T) {
                        _Float16 hi, lo;
                        split_f16(hv, hi, lo);
                        const unsigned word = static_cast<unsigned>(__builtin_bit_cast(unsigned short, hi)) |
                                              ((static_cast<unsigned>(__builtin_bit_cast(unsigned short, lo)) << 16) & 0xfffe0000u) | tag;
                        // lanes 4 k .. 4 k + 3 hold units 4 k .. 4 k + 3 of one clip: the first lane of the quad stores all four
                        u32x4v v;
                        v[0] = word;
                        v[1] = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(word), 0x55, 0xf, 0xf, false));
                        v[2] = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(word), 0xaa, 0xf, 0xf, false));
                        v[3] = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(word), 0xff, 0xf, 0xf, false));
                        if ((gn & 3) == 0) {
                            const int o = static_cast<int>((gt_hx_index(MT, s & 1, dir, n_groups, gg, wg) + (16 * mt + gc) * 4 + (gn >> 2)) * 16);
                            __builtin_amdgcn_raw_buffer_store_b128(v, hrs, o, 0, 16);
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// [LayerNorm(D) -> LeakyReLU(0.01) ->] Linear(D, n_cls): one wave per row
// ---------------------------------------------------------------------------
template <int D, bool DO_LN>
__global__ __launch_bounds__(256) void ln_head_kernel(const float *__restrict__ x, const float *__restrict__ lw,
                                                      const float *__restrict__ lb, const float *__restrict__ fw,
                                                      const float *__restrict__ fb, float *__restrict__ out,
                                                      long n_rows, int n_cls, const unsigned *status) {
    // status: word raised by a kernel upstream that could not finish its work (nullptr: none); the logits are
    // then NaN rather than plausible numbers
    const bool upstream_failed = status != nullptr && *status != 0u;
    constexpr int PER = D / 64;
    const int lane = threadIdx.x & 63;
    const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const float *xr = x + row * D;
    float v[PER];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        v[i] = xr[lane + 64 * i];
        s += v[i];
    }
    if (DO_LN) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const float d = v[i] - mean;
            q = fmaf(d, d, q);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = rsqrtf(q * (1.0f / D) + 1e-5f);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const float n = (v[i] - mean) * rstd * lw[lane + 64 * i] + lb[lane + 64 * i];
            v[i] = n > 0.f ? n : 0.01f * n;
        }
    }
    for (int c = 0; c < n_cls; ++c) {
        const float *wr = fw + static_cast<long>(c) * D;
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < PER; ++i) a = fmaf(v[i], wr[lane + 64 * i], a);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        if (lane == 0) out[row * n_cls + c] = upstream_failed ? __builtin_nanf("") : a + fb[c];
    }
}

// [clip][T'][96][64] (freq, chan) -> [clip][T'][64*96] in the reference's (chan, freq) order
__global__ __launch_bounds__(256) void to_reference_order_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                                 long n_rows) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= n_rows * 6144) return;
    const long row = i / 6144;
    const int e = static_cast<int>(i - row * 6144), c = e / 96, f = e - c * 96;
    dst[i] = src[row * 6144 + f * 64 + c];
}

// ---------------------------------------------------------------------------
// multi-head self-attention, head_dim 128, no mask (TransformerEncoderLayer, batch_first):
// one workgroup = one (clip, head) x 128 queries, a wave = 32 queries; keys/values stream through
// LDS 32 at a time with the running-max softmax.
// ---------------------------------------------------------------------------
constexpr int AT_D = 128, AT_KB = 32;

// Multi-head attention (8 heads x 128, no mask) as a streaming softmax over blocks of 32 keys, on the f16 matrix
// cores with the split-f16 arithmetic of conv_split.h (both operands of both products are activations: all four
// are split on the fly).  Per block of 32 keys a workgroup stages K as hi/lo
// f16 planes [key][dim] and V TRANSPOSED, [dim][key slot], so that both products read their LDS operand with one
// 16-byte load per fragment:  S^T = K . Q^T (A = K from LDS, B = this lane's query from registers), then
// O^T += V^T . P^T (A = V^T from LDS, B = the probabilities straight out of the accumulator registers of S^T --
// the key slots of V^T are numbered in the order in which a lane holds its keys there).  A query's scores,
// maximum, sum and output column all sit in one lane pair (one shfl_xor 32 per reduction); 48 MFMAs of 32 cycles
// per key block and wave (the exact-f32 form on v_mfma_f32_32x32x2_f32 took 128 of 64).
constexpr int AT_KSB = AT_D * 2 + 16;      // bytes per key row of a K plane
constexpr int AT_VSB = AT_KB * 2 + 16;     // bytes per dim row of a V^T plane

// One key block of one (clip, head) in the form the attention kernel contracts: K as f16 hi / lo planes [key][dim] and
// V transposed, [dim][key slot], each row padded as in LDS -- 37 888 contiguous bytes, built once per layer instead of
// by every one of the T / 128 workgroups that walk the keys.  Keys past the end of the clip are zero rows.
constexpr int AT_BLOB = 2 * AT_KB * AT_KSB + 2 * AT_D * AT_VSB;
static_assert(AT_BLOB % 16 == 0, "blocks are copied in 16-byte pieces");
// Tokens per clip from which the packing pass pays: a block is then walked by 8 or more query tiles.  At 250 tokens
// (two query tiles) the pass costs more than it saves: 1.7 ms against 1.0 ms per layer-step of 128 clips.
constexpr int kAttentionPackTokens = 1024;
__global__ __launch_bounds__(256) void attention_pack_kernel(const float *__restrict__ qkv, uint4 *__restrict__ kv, int T) {
    __shared__ __attribute__((aligned(16))) char blob[AT_BLOB];
    char *kh = blob, *kl = kh + AT_KB * AT_KSB, *vth = kl + AT_KB * AT_KSB, *vtl = vth + AT_D * AT_VSB;
    const int tid = threadIdx.x, head = blockIdx.y, kb0 = blockIdx.x * AT_KB;
    const float *base = qkv + static_cast<long>(blockIdx.z) * T * 3072 + head * AT_D;
    for (int i = tid; i < AT_BLOB / 16; i += 256) reinterpret_cast<uint4 *>(blob)[i] = make_uint4(0, 0, 0, 0);   // the padding too
    __syncthreads();
    // K: thread -> (key i >> 5, dims 4 (i & 31) .. + 3)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = tid + 256 * j, key = i >> 5, c4 = (i & 31) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kb0 + key < T) v = *reinterpret_cast<const float4 *>(base + static_cast<long>(kb0 + key) * 3072 + 1024 + c4);
        uint2 hi, lo;
        split2_f16(v.x, v.y, hi.x, lo.x);
        split2_f16(v.z, v.w, hi.y, lo.y);
        *reinterpret_cast<uint2 *>(kh + key * AT_KSB + c4 * 2) = hi;
        *reinterpret_cast<uint2 *>(kl + key * AT_KSB + c4 * 2) = lo;
    }
    // V: thread -> (key pair (tid >> 5) + 8 j, dims (tid & 31) + 32 e); key = acc_row(r, h) = (r & 3) + 8 (r >> 2) + 4 h
    // sits in slot 16 (r >> 3) + 8 h + (r & 7) of V^T's rows (the order in which a lane of the attention kernel holds
    // its probabilities), an even key and its successor in an even slot and its successor
    const int vlq = tid & 31, vpair = tid >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int key = 2 * (vpair + 8 * j);
        const int h = (key >> 2) & 1, r = (key & 3) + 4 * (key >> 3);
        const int slot = 16 * (r >> 3) + 8 * h + (r & 7);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v0 = 0.f, v1 = 0.f;
            if (kb0 + key < T) v0 = base[static_cast<long>(kb0 + key) * 3072 + 2048 + vlq + 32 * e];
            if (kb0 + key + 1 < T) v1 = base[static_cast<long>(kb0 + key + 1) * 3072 + 2048 + vlq + 32 * e];
            unsigned hi, lo;
            split2_f16(v0, v1, hi, lo);
            *reinterpret_cast<unsigned *>(vth + (vlq + 32 * e) * AT_VSB + slot * 2) = hi;
            *reinterpret_cast<unsigned *>(vtl + (vlq + 32 * e) * AT_VSB + slot * 2) = lo;
        }
    }
    __syncthreads();
    uint4 *dst = kv + ((static_cast<long>(blockIdx.z) * 8 + head) * gridDim.x + blockIdx.x) * (AT_BLOB / 16);
    for (int i = tid; i < AT_BLOB / 16; i += 256) dst[i] = reinterpret_cast<const uint4 *>(blob)[i];
}

// n_splits > 1: the keys are cut into n_splits ranges of whole 32-key blocks and blockIdx.x = query tile * n_splits +
// range; a workgroup then leaves its normalised partial output in part_o [range][row][1024] and its running maximum
// and sum in part_ml [range][row][head][2] for attention_merge_kernel (used when the grid would otherwise fill the
// chip's workgroup slots badly: few long recordings).
// PACKED: the keys and values come from attention_pack_kernel's blocks (kv) -- staging is a copy into one of two LDS
// buffers behind the products of the block before, one barrier per block; worth the packing pass when every block is
// walked by many query tiles (long recordings).  Otherwise every workgroup splits the K / V rows of qkv itself into one
// LDS buffer, two barriers per block.  The products, their order and their operand bits are the same in both forms.
template <bool PACKED>
__global__ __launch_bounds__(256, 2) void attention_split_kernel(const float *__restrict__ qkv,   // [clip*T + t][3*1024]
                                                              const uint4 *__restrict__ kv,    // PACKED: the blocks
                                                              float *__restrict__ out,         // [clip*T + t][1024]
                                                              int T, float scale, int n_splits, float *__restrict__ part_o,
                                                              float *__restrict__ part_ml, long total_rows) {
    extern __shared__ __attribute__((aligned(16))) char at_lds[];       // [kh | kl | vth | vtl], twice if PACKED
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, col32 = lane & 31;
    const int head = blockIdx.y;
    const long row0 = static_cast<long>(blockIdx.z) * T;
    const int qtile = blockIdx.x / n_splits, split = blockIdx.x - qtile * n_splits;
    const int q = qtile * 128 + wave * 32 + col32;
    const int kb_per = ((T + AT_KB - 1) / AT_KB + n_splits - 1) / n_splits;          // key blocks per range
    const int k_begin = split * kb_per * AT_KB, k_end = min(T, k_begin + kb_per * AT_KB);
    const float *base = qkv + row0 * 3072 + head * AT_D;

    // B operand of S^T: this lane's query times scale * log2(e) -- the scores come out in the exponent's own unit and
    // the softmax takes exp2 of differences -- dims 16 ks + 8 half + 0..7 of k-step ks
    const float qs = scale * 1.4426950408889634f;
    f16x8 qh[AT_D / 16], ql[AT_D / 16];
#pragma unroll
    for (int ks = 0; ks < AT_D / 16; ++ks) {
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        if (q < T) {
            const float *p = base + static_cast<long>(q) * 3072 + 16 * ks + 8 * half;
            v0 = *reinterpret_cast<const float4 *>(p);
            v1 = *reinterpret_cast<const float4 *>(p + 4);
        }
        uint4 h, l;
        split2_f16(v0.x * qs, v0.y * qs, h.x, l.x);
        split2_f16(v0.z * qs, v0.w * qs, h.y, l.y);
        split2_f16(v1.x * qs, v1.y * qs, h.z, l.z);
        split2_f16(v1.z * qs, v1.w * qs, h.w, l.w);
        qh[ks] = __builtin_bit_cast(f16x8, h);
        ql[ks] = __builtin_bit_cast(f16x8, l);
    }
    f32x16 o[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[nt][r] = 0.f;
    float m = -INFINITY, l = 0.f;

    // ---- staging, PACKED: a key block arrives as attention_pack_kernel left it, the four planes in their LDS layout,
    // 37 888 contiguous bytes behind a buffer descriptor over the (clip, head)'s blocks.  The block after the current
    // one is requested into registers before the products of the current one and copied into the other LDS buffer
    // behind them.
    using f32x4v = __attribute__((ext_vector_type(4))) float;
    constexpr int NPC = (AT_BLOB / 16 + 255) / 256;            // 16-byte pieces per thread and block
    constexpr int LAST = AT_BLOB / 16 - 256 * (NPC - 1);       // threads of the last, partial pass
    f32x4v st[NPC];                                            // (the form not instantiated leaves its registers unused)
    const long n_kb = (T + AT_KB - 1) / AT_KB;
    const __amdgpu_buffer_rsrc_t kvrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4 *>(kv + (PACKED ? (static_cast<long>(blockIdx.z) * 8 + head) * n_kb * (AT_BLOB / 16) : 0)), 0,
        PACKED ? static_cast<int>(n_kb) * AT_BLOB : 0, 0x00020000);
    auto load_block = [&](int kb0) {
        const int off = (kb0 / AT_KB) * AT_BLOB + tid * 16;
#pragma unroll
        for (int j = 0; j < NPC; ++j)          // the idle threads of the last pass read past the block (never stored)
            st[j] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(kvrs, off + j * 4096, 0, 0));
    };
    auto store_block = [&](int b) {
#pragma unroll
        for (int j = 0; j < NPC - 1; ++j) *reinterpret_cast<f32x4v *>(at_lds + b * AT_BLOB + (tid + 256 * j) * 16) = st[j];
        if (tid < LAST) *reinterpret_cast<f32x4v *>(at_lds + b * AT_BLOB + (tid + 256 * (NPC - 1)) * 16) = st[NPC - 1];
    };
    // ---- staging, not PACKED: the K / V rows of a key block are requested one block ahead (into registers) and split
    // into the LDS planes at the top of the next iteration: their latency passes under the current block's products.
    // K: thread -> (key i >> 5, dims 4 (i & 31) .. + 3), one 8-byte LDS write per plane, lanes contiguous.
    // V: thread -> (key PAIR (tid >> 5) + 8 j, dims (tid & 31) + 32 e): a pair of neighbouring keys is a pair of
    // neighbouring slots of V^T, so one packed split gives one 32-bit LDS write per dim and plane, and the 32 lanes of
    // a half wave write rows 80 bytes apart (16 banks, two lanes each) -- with four consecutive dims per thread the
    // rows were 320 bytes apart and every 2-byte write was a 16-way bank conflict.
    float4 kreg[4];
    float vreg[2][4][2];                // [j][e][key of the pair]
    const int vlq = tid & 31, vpair = tid >> 5;
    auto load_kv = [&](int kb0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + 256 * j, key = i >> 5, c4 = (i & 31) * 4;
            kreg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kb0 + key < T) kreg[j] = *reinterpret_cast<const float4 *>(base + static_cast<long>(kb0 + key) * 3072 + 1024 + c4);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int key = 2 * (vpair + 8 * j) + kk;
                const bool ok = kb0 + key < T;
                const float *p = base + static_cast<long>(ok ? kb0 + key : 0) * 3072 + 2048 + vlq;
#pragma unroll
                for (int e = 0; e < 4; ++e) vreg[j][e][kk] = ok ? p[32 * e] : 0.f;
            }
    };
    auto split_kv = [&]() {
        char *kh = at_lds, *kl = kh + AT_KB * AT_KSB, *vth = kl + AT_KB * AT_KSB, *vtl = vth + AT_D * AT_VSB;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + 256 * j, key = i >> 5, c4 = (i & 31) * 4;
            uint2 hi, lo;
            split2_f16(kreg[j].x, kreg[j].y, hi.x, lo.x);
            split2_f16(kreg[j].z, kreg[j].w, hi.y, lo.y);
            *reinterpret_cast<uint2 *>(kh + key * AT_KSB + c4 * 2) = hi;
            *reinterpret_cast<uint2 *>(kl + key * AT_KSB + c4 * 2) = lo;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            // key = acc_row(r, h) = (r & 3) + 8 (r >> 2) + 4 h sits in slot 16 (r >> 3) + 8 h + (r & 7) of V^T's rows;
            // an even key and its successor are an even slot and its successor
            const int key = 2 * (vpair + 8 * j);
            const int h = (key >> 2) & 1, r = (key & 3) + 4 * (key >> 3);
            const int slot = 16 * (r >> 3) + 8 * h + (r & 7);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned hi, lo;
                split2_f16(vreg[j][e][0], vreg[j][e][1], hi, lo);
                *reinterpret_cast<unsigned *>(vth + (vlq + 32 * e) * AT_VSB + slot * 2) = hi;
                *reinterpret_cast<unsigned *>(vtl + (vlq + 32 * e) * AT_VSB + slot * 2) = lo;
            }
        }
    };
    if constexpr (PACKED) {
        if (k_begin < k_end) {
            load_block(k_begin);
            store_block(0);
        }
        __syncthreads();
    } else {
        load_kv(k_begin);
    }
    int buf = 0;
    for (int kb0 = k_begin; kb0 < k_end; kb0 += AT_KB) {
        if constexpr (PACKED) {
            if (kb0 + AT_KB < k_end) load_block(kb0 + AT_KB);
        } else {
            __syncthreads();
            split_kv();
            __syncthreads();
            if (kb0 + AT_KB < k_end) load_kv(kb0 + AT_KB);
        }
        const char *kh = at_lds + buf * AT_BLOB, *kl = kh + AT_KB * AT_KSB;
        const char *vth = kl + AT_KB * AT_KSB, *vtl = vth + AT_D * AT_VSB;
        // S^T[key][query]: A row = key col32, B column = query col32
        f32x16 sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f;
        // (one accumulator: a second one for the odd k-steps, so that no product waits for the one issued just before
        // it, measured 3 % slower -- 16 more additions, and the other wave of the SIMD fills the gaps anyway)
#pragma unroll
        for (int ks = 0; ks < AT_D / 16; ++ks) {
            const f16x8 ah = *reinterpret_cast<const f16x8 *>(kh + col32 * AT_KSB + ks * 32 + 16 * half);
            const f16x8 al = *reinterpret_cast<const f16x8 *>(kl + col32 * AT_KSB + ks * 32 + 16 * half);
            sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[ks], sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[ks], sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[ks], sc, 0, 0, 0);
        }
        if (kb0 + AT_KB > T) {                       // only the clip's last block has keys past its end
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kb0 + acc_row(r, half) >= T) sc[r] = -INFINITY;
        }
        float mx = sc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m, mx);            // finite: key kb0 < T is in every block
        const float corr = __builtin_amdgcn_exp2f(m - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sc[r] = __builtin_amdgcn_exp2f(sc[r] - m_new);
            psum += sc[r];
        }
        psum += __shfl_xor(psum, 32);
        l = fmaf(l, corr, psum);
        const bool moved = m_new != m;
        m = m_new;
        if (__any(moved)) {                          // the running maximum moved for some query of the wave: rescale
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[nt][r] *= corr;
        }
        // B operand of O^T: this lane's probabilities, registers 8 ks .. 8 ks + 7 = key slots 16 ks + 8 half + 0..7
        f16x8 pph[2], ppl[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 h, lw;
            split2_f16(sc[8 * ks], sc[8 * ks + 1], h.x, lw.x);
            split2_f16(sc[8 * ks + 2], sc[8 * ks + 3], h.y, lw.y);
            split2_f16(sc[8 * ks + 4], sc[8 * ks + 5], h.z, lw.z);
            split2_f16(sc[8 * ks + 6], sc[8 * ks + 7], h.w, lw.w);
            pph[ks] = __builtin_bit_cast(f16x8, h);
            ppl[ks] = __builtin_bit_cast(f16x8, lw);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const f16x8 ah = *reinterpret_cast<const f16x8 *>(vth + (32 * nt + col32) * AT_VSB + (16 * ks + 8 * half) * 2);
                const f16x8 al = *reinterpret_cast<const f16x8 *>(vtl + (32 * nt + col32) * AT_VSB + (16 * ks + 8 * half) * 2);
                o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, pph[ks], o[nt], 0, 0, 0);
                o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ppl[ks], o[nt], 0, 0, 0);
                o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, pph[ks], o[nt], 0, 0, 0);
            }
        if constexpr (PACKED) {
            if (kb0 + AT_KB < k_end) store_block(buf ^ 1);
            lds_barrier();
            buf ^= 1;
        }
    }
    if (q < T) {
        const float inv = l > 0.f ? 1.0f / l : 0.f;          // l == 0: an empty key range (its weight in the merge is 0)
        float *op = out + (row0 + q) * 1024 + head * AT_D;
        if (n_splits > 1) {
            op = part_o + (static_cast<long>(split) * total_rows + row0 + q) * 1024 + head * AT_D;
            if (half == 0) {
                // the running maximum in natural-log units, as attention_merge_kernel weighs the ranges
                float *ml = part_ml + ((static_cast<long>(split) * total_rows + row0 + q) * 8 + head) * 2;
                ml[0] = m * 0.6931471805599453f;
                ml[1] = l;
            }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                *reinterpret_cast<float4 *>(op + 32 * nt + 8 * r4 + 4 * half) =
                    make_float4(o[nt][4 * r4] * inv, o[nt][4 * r4 + 1] * inv, o[nt][4 * r4 + 2] * inv, o[nt][4 * r4 + 3] * inv);
    }
}

// Combines the partial outputs of n_splits key ranges: out = sum_s w_s o_s / sum_s w_s with w_s = l_s exp(m_s - max m).
// One thread per (row, head, 4 dims).
__global__ __launch_bounds__(256) void attention_merge_kernel(const float *__restrict__ part_o, const float *__restrict__ part_ml,
                                                              float *__restrict__ out, long total_rows, int n_splits) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= total_rows * 256) return;
    const long row = i >> 8;
    const int d4 = static_cast<int>(i & 255) * 4, head = d4 >> 7;
    float mx = -INFINITY;
    for (int sp = 0; sp < n_splits; ++sp) mx = fmaxf(mx, part_ml[((sp * total_rows + row) * 8 + head) * 2]);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float wsum = 0.f;
    for (int sp = 0; sp < n_splits; ++sp) {
        const float *ml = part_ml + ((sp * total_rows + row) * 8 + head) * 2;
        const float w = ml[1] > 0.f ? ml[1] * __expf(ml[0] - mx) : 0.f;
        const float4 o = *reinterpret_cast<const float4 *>(part_o + (sp * total_rows + row) * 1024 + d4);
        acc.x = fmaf(w, o.x, acc.x); acc.y = fmaf(w, o.y, acc.y); acc.z = fmaf(w, o.z, acc.z); acc.w = fmaf(w, o.w, acc.w);
        wsum += w;
    }
    const float inv = 1.0f / wsum;
    *reinterpret_cast<float4 *>(out + row * 1024 + d4) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
}

// out = LayerNorm(x + y) over rows of 1024 (post-LN residual), one wave per row; out may alias x
__global__ __launch_bounds__(256) void add_ln_kernel(const float *x, const float *__restrict__ y,
                                                     const float *__restrict__ w, const float *__restrict__ b,
                                                     float *out, long n_rows) {
    constexpr int D = 1024, PER = D / 256;
    const int lane = threadIdx.x & 63;
    const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    float4 v[PER];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const float4 a = *reinterpret_cast<const float4 *>(x + row * D + (lane + 64 * i) * 4);
        const float4 c = *reinterpret_cast<const float4 *>(y + row * D + (lane + 64 * i) * 4);
        v[i] = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * (1.0f / D);
    float qq = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
        qq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) qq += __shfl_xor(qq, o);
    const float rstd = rsqrtf(qq * (1.0f / D) + 1e-5f);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const float4 ww = *reinterpret_cast<const float4 *>(w + (lane + 64 * i) * 4);
        const float4 bb = *reinterpret_cast<const float4 *>(b + (lane + 64 * i) * 4);
        *reinterpret_cast<float4 *>(out + row * D + (lane + 64 * i) * 4) =
            make_float4((v[i].x - mean) * rstd * ww.x + bb.x, (v[i].y - mean) * rstd * ww.y + bb.y,
                        (v[i].z - mean) * rstd * ww.z + bb.z, (v[i].w - mean) * rstd * ww.w + bb.w);
    }
}

// ---------------------------------------------------------------------------
// BasicRCNN (models.py:7-66) pieces that are not a 3x3 split-f16 conv
// ---------------------------------------------------------------------------
constexpr int BR_C = 128;            // conv channels (models.py:14)
constexpr int BR_H = 32;             // GRU hidden size (models.py:9)

// conv1 (3x3, 1 -> 128 channels, + bias) with MaxPool2d((1, 5), stride 1) + ReLU (models.py:29-30,47-50) in the
// same pass: x [clip][T][F] -> y [clip][T][F - 4][128].  K = 9 is too thin for the matrix cores.  One thread = 4
// channels x a run of 12 pooled outputs: it evaluates the conv at the 16 positions the run's windows cover (1.33x
// the minimum) and slides the maximum over them in registers; the 32 lanes of a half wave are the 32 channel groups
// of one run, so every store instruction writes 512 contiguous bytes per run position.
constexpr int BC1_RUN = 12;
__global__ __launch_bounds__(256) void basic_conv1_pool_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                               const float *__restrict__ bias, float *__restrict__ y,
                                                               long n_rows, int T, int F) {
    __shared__ float wsm[BR_C][10];
    for (int i = threadIdx.x; i < BR_C * 10; i += 256) {
        const int c = i / 10, k = i - c * 10;
        wsm[c][k] = k < 9 ? w[c * 9 + k] : bias[c];
    }
    __syncthreads();
    const int Wo = F - 4, n_runs = (Wo + BC1_RUN - 1) / BC1_RUN;
    const long idx = blockIdx.x * 256L + threadIdx.x;
    if (idx >= n_rows * n_runs * (BR_C / 4)) return;
    const int c4 = static_cast<int>(idx % (BR_C / 4));
    const long rr = idx / (BR_C / 4);
    const int f0 = static_cast<int>(rr % n_runs) * BC1_RUN;
    const long row = rr / n_runs;                        // clip * T + t
    const int t = static_cast<int>(row % T);
    constexpr int NP = BC1_RUN + 4;                      // conv positions f0 .. f0 + NP - 1
    float cv[4][NP];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < NP; ++i) cv[c][i] = wsm[c4 * 4 + c][9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int tt = t - 1 + ky;
        const bool row_ok = tt >= 0 && tt < T;
        const float *xr = x + (row - 1 + ky) * F;       // rows of one clip are consecutive; row_ok keeps it inside the clip
        float in[NP + 2];
#pragma unroll
        for (int i = 0; i < NP + 2; ++i) {
            const int ff = f0 - 1 + i;
            in[i] = (row_ok && ff >= 0 && ff < F) ? xr[ff] : 0.f;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float w0 = wsm[c4 * 4 + c][ky * 3], w1 = wsm[c4 * 4 + c][ky * 3 + 1], w2 = wsm[c4 * 4 + c][ky * 3 + 2];
#pragma unroll
            for (int i = 0; i < NP; ++i) cv[c][i] = fmaf(in[i + 2], w2, fmaf(in[i + 1], w1, fmaf(in[i], w0, cv[c][i])));
        }
    }
    // positions past the row's end hold the bias-only value of an all-zero patch or partial sums: they only feed
    // outputs past Wo, which are not written
    float *dst = y + (row * Wo + f0) * BR_C + c4 * 4;
#pragma unroll
    for (int o = 0; o < BC1_RUN; ++o) {
        if (f0 + o < Wo) {
            float m[4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                m[c] = fmaxf(fmaxf(fmaxf(fmaxf(cv[c][o], cv[c][o + 1]), fmaxf(cv[c][o + 2], cv[c][o + 3])), cv[c][o + 4]), 0.f);
            *reinterpret_cast<float4 *>(dst + static_cast<long>(o) * BR_C) = make_float4(m[0], m[1], m[2], m[3]);
        }
    }
}

// GRU(K, 32) recurrence, one direction, + ReLU + Linear(32, n_cls) (models.py:62-65): one wave per
// clip.  gi [clip*T2][128] holds W_ih x + b_ih (+ b_hh for r, z) in columns 0..95.
__global__ __launch_bounds__(64) void basic_gru_head_kernel(const float *__restrict__ gi, const float *__restrict__ whh,
                                                            const float *__restrict__ bhn, const float *__restrict__ fw,
                                                            const float *__restrict__ fb, float *__restrict__ logits,
                                                            int T2, int n_cls) {
    __shared__ float hs[BR_H], hrelu[BR_H], gh[3 * BR_H];
    const int lane = threadIdx.x;
    const long b = blockIdx.x;
    // lane l computes the recurrent pre-activations of gate columns l and l + 64 (96 in all)
    float w0[BR_H], w1[BR_H];
#pragma unroll
    for (int k = 0; k < BR_H; ++k) {
        w0[k] = whh[lane * BR_H + k];
        w1[k] = lane < 32 ? whh[(lane + 64) * BR_H + k] : 0.f;
    }
    const float b_hn = lane < BR_H ? bhn[lane] : 0.f;
    if (lane < BR_H) hs[lane] = 0.f;
    float h = 0.f;
    __syncthreads();
    for (int t = 0; t < T2; ++t) {
        const float *g = gi + (b * T2 + t) * 128;
        const float gr = lane < BR_H ? g[lane] : 0.f, gz = lane < BR_H ? g[BR_H + lane] : 0.f,
                    gn = lane < BR_H ? g[2 * BR_H + lane] : 0.f;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int k = 0; k < BR_H; ++k) {
            const float hv = hs[k];
            a0 = fmaf(hv, w0[k], a0);
            a1 = fmaf(hv, w1[k], a1);
        }
        gh[lane] = a0;
        if (lane < 32) gh[lane + 64] = a1;
        __syncthreads();
        if (lane < BR_H) {
            const float r = 1.0f / (1.0f + __expf(-(gr + gh[lane])));
            const float z = 1.0f / (1.0f + __expf(-(gz + gh[BR_H + lane])));
            const float n = tanhf(fmaf(r, gh[2 * BR_H + lane] + b_hn, gn));
            h = fmaf(z, h - n, n);
            hs[lane] = h;
            hrelu[lane] = fmaxf(h, 0.f);
        }
        __syncthreads();
        for (int c = lane; c < n_cls; c += 64) {
            float a = fb[c];
#pragma unroll
            for (int k = 0; k < BR_H; ++k) a = fmaf(hrelu[k], fw[c * BR_H + k], a);
            logits[(b * T2 + t) * n_cls + c] = a;
        }
    }
}

constexpr int kShakeChunk = 16;     // clips per pass through the conv blocks (u1 is 49 MB per clip)
constexpr int kMaxEncoderLayers = SEDMI_SHAKE_MAX_LAYERS;

struct BlockDev {
    uint16_t *wa = nullptr, *wb = nullptr;      // packed split weights: conv A (blocks 2, 3), branches (all)
    float *es_a = nullptr, *et_a = nullptr;     // conv A epilogue affine per GEMM column
    float *es_b = nullptr, *et_b = nullptr;     // branch conv epilogue (scale undo, bias) per GEMM column
    float *s2 = nullptr, *t2 = nullptr;         // second BatchNorm affine
};

struct ShakeCnnDev {
    uint16_t *d_c1s = nullptr;          // block 1 conv: folded split-f16 weights [2][2][32][16]
    float *d_c1bias = nullptr;          // and the scaled row shifts
    float c1_ascale = 1.f, c1_bscale = 1.f;
    BlockDev blk[3];
};

// y = act(x . W^T + b) with W [N][K] row-major (torch Linear), packed for gemm_split_kernel
struct LinearDev {
    uint16_t *w = nullptr;
    float *b = nullptr;
    float oscale = 1.f;
    int N = 0, K = 0;
};

struct EncoderLayerDev {
    LinearDev in_proj, out_proj, lin1, lin2;
    float *n1w = nullptr, *n1b = nullptr, *n2w = nullptr, *n2b = nullptr;
};

}  // namespace

struct sedmi_shake_rcnn {
    int n_class = 0, num_cus = 0;
    int recurrence = SEDMI_RECURRENCE_AUTO;
    ShakeCnnDev cnn;
    LinearDev wih;                       // both directions: N = 2 * 3072
    uint16_t *d_whh = nullptr;
    float whh_oscale = 1.f;
    float *d_bhn = nullptr, *d_lw = nullptr, *d_lb = nullptr, *d_fw = nullptr, *d_fb = nullptr;
};

struct sedmi_shake_transformer {
    int n_class = 0, n_layers = 0, num_cus = 0;
    ShakeCnnDev cnn;
    LinearDev fc1;
    EncoderLayerDev layers[kMaxEncoderLayers];
    float *d_fw = nullptr, *d_fb = nullptr;
};

struct sedmi_basic_rcnn {
    int n_class = 0, F = 0, K = 0;        // K = GRU input width = 128 * (F - 8) / 2
    float *d_w1 = nullptr, *d_b1 = nullptr;
    uint16_t *d_wc[2] = {nullptr, nullptr};               // conv2, conv3 packed split weights
    float *d_es[2] = {nullptr, nullptr}, *d_et[2] = {nullptr, nullptr};
    LinearDev wih;                                        // 96 gate columns padded to 128
    float *d_whh = nullptr, *d_bhn = nullptr, *d_fw = nullptr, *d_fb = nullptr;
};

namespace {

struct ShakeScratch {
    float *u1, *y1, *u2, *y2, *u3, *seq;
    float *gi, *rec, *h0, *h1;               // ShakeRCNN
    uint4 *hx;                               // persistent recurrence: exchange buffers
    unsigned *sync_words;                    // and its status word + flags
    size_t sync_bytes;
    float *x, *qkv, *att, *ff;               // ShakeTransformer
    float *part_o, *part_ml;                 // attention over several key ranges: partial outputs, (max, sum)
    uint4 *kvp;                              // keys and values of the layer in flight as attention_pack_kernel's blocks
    int att_splits;
    size_t bytes;
};

enum { SHAKE_CNN_ONLY = 0, SHAKE_RCNN = 1, SHAKE_TRANSFORMER = 2 };

// Key ranges for the attention kernel.  Its grid is n * 8 heads * ceil(T / 128) equal workgroups on 2 * num_cus
// slots; when that is a small non-integer number of rounds (2 recordings x 6000 tokens: 752 workgroups on 512
// slots = 1.47 -> 2 rounds) cutting the keys into S ranges gives S times as many workgroups of 1/S the length:
// ceil(1.47 S) / S rounds.  Ranges keep at least 512 keys; S is only raised for a gain of 10 % or more.
int attention_splits(long n, long T4, int num_cus) {
    if (num_cus <= 0 || T4 < 1024) return 1;
    const double tiles = static_cast<double>(n) * 8 * sedmi::ceil_div(T4, 128), slots = 2.0 * num_cus;
    int best = 1;
    double best_t = std::ceil(tiles / slots);
    for (int S = 2; S <= 4 && T4 / S >= 512; ++S) {
        const double t = std::ceil(tiles * S / slots) / S;
        if (t < 0.9 * best_t) { best = S; best_t = t; }
    }
    return best;
}

ShakeScratch carve_shake(void *ws, long n, long T, int kind, int num_cus = 0) {
    ShakeScratch sc{};
    size_t off = 0;
    auto take = [&](size_t n_floats) {
        float *p = reinterpret_cast<float *>(static_cast<char *>(ws) + off);
        off += sedmi::align_up(n_floats * sizeof(float), 256);
        return p;
    };
    const long ch = std::min<long>(n, kShakeChunk), T2 = T / 2, T4 = T / 4;
    sc.u1 = take(static_cast<size_t>(ch) * T * 768 * 16);
    sc.y1 = take(static_cast<size_t>(ch) * T2 * 384 * 16);
    sc.u2 = take(static_cast<size_t>(ch) * T2 * 384 * 32);
    sc.y2 = take(static_cast<size_t>(ch) * T4 * 192 * 32);
    sc.u3 = take(static_cast<size_t>(ch) * T4 * 192 * 64);
    sc.seq = take(static_cast<size_t>(n) * T4 * 6144);
    if (kind == SHAKE_RCNN) {
        sc.gi = take(static_cast<size_t>(n) * T4 * 6144);
        sc.rec = take(static_cast<size_t>(n) * T4 * 2048);
        sc.h0 = take(static_cast<size_t>(2) * n * GB_H);
        sc.h1 = take(static_cast<size_t>(2) * n * GB_H);
        // sized for the largest geometry the persistent recurrence takes (2 sets x GP_MAX_G groups of 16 GP_MAX_MT clips)
        // (in units of 16 clips; a batch in 32-clip groups rounds up to whole groups)
        const long n_groups = std::min<long>(GP_MAX_MT * sedmi::ceil_div(n, 16 * GP_MAX_MT), 2 * GP_MAX_G * GP_MAX_MT);
        sc.hx = reinterpret_cast<uint4 *>(take(static_cast<size_t>(gp_hx_index(1, 2, 0, static_cast<int>(n_groups), 0, 0, 0)) * 4));
        sc.sync_bytes = sizeof(unsigned) * (GP_STATUS_WORDS + 2 * n_groups * GP_WG);
        sc.sync_words = reinterpret_cast<unsigned *>(take(sc.sync_bytes / sizeof(float)));
    } else if (kind == SHAKE_TRANSFORMER) {
        sc.x = take(static_cast<size_t>(n) * T4 * 1024);
        sc.qkv = take(static_cast<size_t>(n) * T4 * 3072);
        sc.att = take(static_cast<size_t>(n) * T4 * 1024);
        sc.ff = take(static_cast<size_t>(n) * T4 * 2048);
        sc.kvp = nullptr;
        if (T4 >= kAttentionPackTokens)
            sc.kvp = reinterpret_cast<uint4 *>(take(static_cast<size_t>(n) * 8 * sedmi::ceil_div(T4, AT_KB) * (AT_BLOB / 4)));
        sc.att_splits = attention_splits(n, T4, num_cus);
        if (sc.att_splits > 1) {
            sc.part_o = take(static_cast<size_t>(sc.att_splits) * n * T4 * 1024);
            sc.part_ml = take(static_cast<size_t>(sc.att_splits) * n * T4 * 16);
        }
    }
    sc.bytes = off;
    return sc;
}

// kernels with 128 GEMM columns split them over two waves (gate<64> 3.04 -> 2.74 ms, conv<32,128> 1.56 -> 1.51 ms per
// 128 clips); requesting the next input chunk before the MFMAs of the current one on top of that gained nothing
constexpr int C2_WN = 2;
#ifndef SHAKE_TR16
#define SHAKE_TR16 8          // tile rows of the two kernels that stage 16-channel chunks
#endif
template <int CIN, int N, int CCH, int MODE, int PH, int PW, int WN = 1, int TR = C2_TR>
int launch_conv2d(const char *name, const uint16_t *wp, const float *es, const float *et, float alpha, const float *s2,
                  const float *t2, const float *x, float *y, long n, int H, int W, hipStream_t s) {
    using G = Conv2dGeom<CIN, N, CCH, TR>;
    const int tiles_x = static_cast<int>(MODE == MODE_POOL_W5 ? sedmi::ceil_div(W - 4, C2_TC - 4) : sedmi::ceil_div(W, C2_TC));
    const int tiles_y = static_cast<int>(sedmi::ceil_div(H, TR));
    // one clip's input and output are addressed with 32-bit offsets, and an offset sent out of range (2^31) must stay
    // below 2^32 when the rows of a tile are added to it
    // (an odd last row of a pooling kernel has no output: its stores fall out of the output descriptor's range)
    if ((MODE == MODE_GLU || MODE == MODE_GATE) && W % C2_TC)
        return sedmi::set_error(SEDMI_EINVAL, "%s: a width of %d is not a whole number of %d-column tiles", name, W, C2_TC);
    const double out_ch = MODE == MODE_GLU ? N / 2.0 : MODE == MODE_GATE ? N / 2.0 / (PH * PW) : 0.0;   // the other modes store through pointers
    if (4.0 * (H + TR + 2) * W * std::max<double>(CIN, out_ch) >= 2147483648.0)
        return sedmi::set_error(SEDMI_EINVAL, "%s: a clip of %d x %d positions does not fit 32-bit byte offsets", name, H, W);
    auto kern = conv2d_split_kernel<CIN, N, CCH, MODE, PH, PW, WN, TR>;
    if (int rc_lds = sedmi::set_max_dynamic_lds(reinterpret_cast<const void *>(kern), G::LDS_BYTES)) return rc_lds;
    const double pos = static_cast<double>(n) * H * W;
    sedmi::ProfileScope prof(name, s, pos * 2.0 * 9 * CIN * N,
                             pos * 4.0 * (CIN + (MODE == MODE_BIAS || MODE == MODE_POOL_W5 ? N : MODE == MODE_POOL_NCHW ? N / 4.0 : MODE == MODE_GLU ? N / 2.0
                                                                                        : CIN + N / 2.0 / (PH * PW))));
    const unsigned n_tiles = static_cast<unsigned>(n * tiles_x * tiles_y);
    hipLaunchKernelGGL(kern, dim3((n_tiles + 7) / 8 * 8), dim3(256), G::LDS_BYTES, s, x,
                       reinterpret_cast<const uint4 *>(wp), es, et, alpha, s2, t2, y, H, W, tiles_x, tiles_y, n_tiles);
    SEDMI_CHECK_LAUNCH();
    return SEDMI_OK;
}

int run_shake_cnn(const ShakeCnnDev *m, const float *feats, long n, int T, const float *alphas, float *seq,
                  const ShakeScratch &sc, hipStream_t s) {
    const int T2 = T / 2, T4 = T / 4;
    for (long c0 = 0; c0 < n; c0 += kShakeChunk) {
        const long nc = std::min<long>(kShakeChunk, n - c0);
        const float *x = feats + c0 * static_cast<long>(T) * 768;
        {
            sedmi::ProfileScope prof("shake_conv1", s, 2.0 * 25 * 32 * nc * T * 768.0, 4.0 * nc * T * 768.0 * 17);
            const int tiles = static_cast<int>(sedmi::ceil_div(T, C1_TR));
            const size_t lds = sizeof(float) * (C1_TR + 4) * (768 + 2 * C1_HO);
            hipLaunchKernelGGL(shake_conv1_split_kernel, dim3(static_cast<unsigned>(nc * tiles)), dim3(256), lds, s, x,
                               reinterpret_cast<const uint4 *>(m->d_c1s), m->d_c1bias, m->c1_ascale, m->c1_bscale, sc.u1,
                               T, 768, tiles);
            SEDMI_CHECK_LAUNCH();
        }
        int rc;
        const BlockDev *b = m->blk;
        if ((rc = launch_conv2d<16, 32, 16, MODE_GATE, 2, 2, 1, SHAKE_TR16>("shake_gate<16>", b[0].wb, b[0].es_b, b[0].et_b, alphas[0],
                                                            b[0].s2, b[0].t2, sc.u1, sc.y1, nc, T, 768, s))) return rc;
        if ((rc = launch_conv2d<16, 64, 16, MODE_GLU, 1, 1, 1, SHAKE_TR16>("shake_conv<16,64>", b[1].wa, b[1].es_a, b[1].et_a, 0.f,
                                                           nullptr, nullptr, sc.y1, sc.u2, nc, T2, 384, s))) return rc;
        if ((rc = launch_conv2d<32, 64, 32, MODE_GATE, 2, 2>("shake_gate<32>", b[1].wb, b[1].es_b, b[1].et_b, alphas[1],
                                                            b[1].s2, b[1].t2, sc.u2, sc.y2, nc, T2, 384, s))) return rc;
        if ((rc = launch_conv2d<32, 128, 32, MODE_GLU, 1, 1, C2_WN>("shake_conv<32,128>", b[2].wa, b[2].es_a, b[2].et_a, 0.f,
                                                            nullptr, nullptr, sc.y2, sc.u3, nc, T4, 192, s))) return rc;
        if ((rc = launch_conv2d<64, 128, 32, MODE_GATE, 1, 2, C2_WN>("shake_gate<64>", b[2].wb, b[2].es_b, b[2].et_b, alphas[2],
                                                             b[2].s2, b[2].t2, sc.u3, seq + c0 * T4 * 6144, nc, T4, 192,
                                                             s))) return rc;
    }
    return SEDMI_OK;
}

int launch_linear(const char *name, const LinearDev &L, const float *A, float *Cout, long rows, bool relu, hipStream_t s) {
    // 256 x 128 tiles, 32 columns of K per chunk, two sets of planes: 80 KB of LDS, two workgroups fill a CU exactly.
    // Measured on the GRU input projection (32000 x 6144 x 6144): 128-row tiles with one set of 128-column planes
    // 334 TFLOP/s, 256-row tiles 362, 256-row tiles with two sets 392.
    constexpr int MT = 4, KC = 32, TM = 64 * MT;
    constexpr bool DB = true;
    // the kernel addresses one tile's rows of A and of C with 32-bit byte offsets
    if (L.K % KC || L.N % 128 || 4.0 * TM * std::max(L.K, L.N) >= 2147483648.0) return SEDMI_EINVAL;
    auto kern = gemm_split_kernel<MT, KC, DB>;
    const size_t lds = (DB ? 4 : 2) * TM * (KC * 2 + 16);
    if (int rc_lds = sedmi::set_max_dynamic_lds(reinterpret_cast<const void *>(kern), lds)) return rc_lds;
    sedmi::ProfileScope prof(name, s, 2.0 * rows * L.N * static_cast<double>(L.K), 4.0 * rows * (L.N + static_cast<double>(L.K)));
    const unsigned m_tiles = static_cast<unsigned>(sedmi::ceil_div(rows, TM)), n_tiles = L.N / 128;
    hipLaunchKernelGGL(kern, dim3((m_tiles * n_tiles + 7) / 8 * 8), dim3(256), lds, s,
                       A, static_cast<long>(L.K), reinterpret_cast<const uint4 *>(L.w), L.N, L.K, L.b, L.oscale, Cout,
                       static_cast<long>(L.N), rows, relu ? 1 : 0, m_tiles, n_tiles);
    SEDMI_CHECK_LAUNCH();
    return SEDMI_OK;
}

// 0: one launch per step; 1: the persistent kernel whenever the batch fits it (default); read per call so
// that a benchmark can compare the two
bool gru_persistent_wanted() {
    const char *e = std::getenv("SEDMI_SHAKE_GRU_PERSISTENT");
    return e == nullptr || e[0] != '0';
}

// Polls a waiting wave of the persistent recurrence makes before it gives up and raises the status word.
// SEDMI_SHAKE_GRU_SPIN_LIMIT overrides the default (tests set it to 0 to drive the give-up path: the first poll
// that finds its producers not ready yet then ends the wait).
unsigned gru_spin_limit() {
    const char *e = std::getenv("SEDMI_SHAKE_GRU_SPIN_LIMIT");
    if (e == nullptr || e[0] == '\0') return GP_SPIN_LIMIT;
    char *end = nullptr;
    const unsigned long v = std::strtoul(e, &end, 10);
    return end == e ? GP_SPIN_LIMIT : static_cast<unsigned>(std::min<unsigned long>(v, 0xfffffff0ul));
}

// Which hand-off the persistent recurrence uses: separate flags (gru_big_persistent_kernel) or step tags inside
// the data (gru_big_tagged_kernel).  Measured per step: 4 clips 4.3 us with flags / 3.6 with tags, 64 clips 6.4 /
// 5.9, 128 clips (two groups per workgroup) 6.8 / 8.9 -- tags when a workgroup serves one group per step, flags
// when it serves several.  SEDMI_SHAKE_GRU_EXCHANGE=flag|tag overrides (tests run both).
bool gru_exchange_by_flags(int groups_per_workgroup) {
    const char *e = std::getenv("SEDMI_SHAKE_GRU_EXCHANGE");
    if (e != nullptr && e[0] == 'f') return true;
    if (e != nullptr && e[0] == 't') return false;
    return groups_per_workgroup > 1;
}

// Can `blocks` workgroups of this kernel be resident at the same time on this device?  The runtime's occupancy
// figure (registers, LDS, waves per CU) times the number of CUs; asked once per kernel and device.  It says nothing
// about OTHER work on the GPU -- that case is caught by the bounded spins and reported by sedmi_shake_rcnn_check.
bool persistent_grid_fits(const void *kern, int blocks, int num_cus) {
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, int> per_cu;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(mu);
    auto it = per_cu.find({kern, device});
    if (it == per_cu.end()) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 512, GP_LDS_BYTES) != hipSuccess) {
            (void)hipGetLastError();
            nb = 0;
        }
        it = per_cu.emplace(std::make_pair(kern, device), nb).first;
    }
    return static_cast<long>(it->second) * num_cus >= blocks;
}

constexpr int kPersistentDoesNotFit = 1;      // launch_gru_persistent: use one launch per step instead

template <int G, int MT>
int launch_gru_persistent(const sedmi_shake_rcnn *m, long n, int T4, int n_groups, int n_sets, const ShakeScratch &sc,
                          hipStream_t s) {
    if (!gru_exchange_by_flags(G)) {
        auto kern = gru_big_tagged_kernel<G, MT>;
        if (int rc_lds = sedmi::set_max_dynamic_lds(reinterpret_cast<const void *>(kern), GP_LDS_BYTES)) return rc_lds;
        if (!persistent_grid_fits(reinterpret_cast<const void *>(kern), 2 * GP_WG * n_sets, m->num_cus)) return kPersistentDoesNotFit;
        SEDMI_CHECK_HIP(hipMemsetAsync(sc.hx, 0xff, static_cast<size_t>(gt_hx_index(MT, 2, 0, n_groups, 0, 0)) * 16, s));
        hipLaunchKernelGGL(kern, dim3(GP_WG, 2, static_cast<unsigned>(n_sets)), dim3(512), GP_LDS_BYTES, s, sc.gi,
                           reinterpret_cast<const uint4 *>(m->d_whh), m->whh_oscale, m->d_bhn, sc.rec, sc.hx, sc.sync_words,
                           n, T4, n_groups, n_sets, gru_spin_limit());
        SEDMI_CHECK_LAUNCH();
        return SEDMI_OK;
    }
    auto kern = gru_big_persistent_kernel<G, MT>;
    if (int rc_lds = sedmi::set_max_dynamic_lds(reinterpret_cast<const void *>(kern), GP_LDS_BYTES)) return rc_lds;
    if (!persistent_grid_fits(reinterpret_cast<const void *>(kern), 2 * GP_WG * n_sets, m->num_cus)) return kPersistentDoesNotFit;
    hipLaunchKernelGGL(kern, dim3(GP_WG, 2, static_cast<unsigned>(n_sets)), dim3(512), GP_LDS_BYTES, s, sc.gi,
                       reinterpret_cast<const uint4 *>(m->d_whh), m->whh_oscale, m->d_bhn, sc.rec, sc.hx, sc.sync_words, n,
                       T4, n_groups, n_sets, gru_spin_limit());
    SEDMI_CHECK_LAUNCH();
    return SEDMI_OK;
}

// the status word at sc.sync_words[0] is zeroed either way (the head kernel reads it)
int run_shake_gru(const sedmi_shake_rcnn *m, const float *seq, long n, int T4, const ShakeScratch &sc, hipStream_t s) {
    const long rows = n * T4;
    int rc;
    if ((rc = launch_linear("shake_gru_input_proj", m->wih, seq, sc.gi, rows, false, s))) return rc;
    SEDMI_CHECK_HIP(hipMemsetAsync(sc.sync_words, 0, sc.sync_bytes, s));
    // Up to 192 clips: the whole recurrence in one launch, W_hh resident in registers across 128 or 256 CUs
    // (needs every workgroup resident at once, one per CU).  Clips go in groups of 16 (up to 32 clips) or 32.
    const int MT = n <= 32 ? 1 : 2;
    const int n_groups = static_cast<int>(sedmi::ceil_div(n, 16 * MT));
    const int n_sets = n_groups >= 2 && 2 * 2 * GP_WG <= m->num_cus ? 2 : 1;
    const int G = static_cast<int>(sedmi::ceil_div(n_groups, n_sets));
    if (m->recurrence == SEDMI_RECURRENCE_AUTO && gru_persistent_wanted() && G <= GP_MAX_G &&
        2 * GP_WG * n_sets <= m->num_cus && T4 >= 2) {
        sedmi::ProfileScope prof("shake_gru_persistent", s, 2.0 * rows * 2 * 3 * GB_H * GB_H, 4.0 * rows * (6144.0 + 2048.0));
        int prc;
        if (MT == 1) prc = launch_gru_persistent<1, 1>(m, n, T4, n_groups, n_sets, sc, s);
        else if (G == 1) prc = launch_gru_persistent<1, 2>(m, n, T4, n_groups, n_sets, sc, s);
        else if (G == 2) prc = launch_gru_persistent<2, 2>(m, n, T4, n_groups, n_sets, sc, s);
        else prc = launch_gru_persistent<3, 2>(m, n, T4, n_groups, n_sets, sc, s);
        if (prc != kPersistentDoesNotFit) return prc;
        // the runtime says the grid cannot be resident at once on this device: one launch per step below
    }
    SEDMI_CHECK_HIP(hipMemsetAsync(sc.h0, 0, sizeof(float) * 2 * n * GB_H, s));
    const size_t lds = 4 * 2 * 32 * ((GB_H / 4) * 2 + 16);
    if (int rc_lds = sedmi::set_max_dynamic_lds(reinterpret_cast<const void *>(gru_big_step_kernel), lds)) return rc_lds;
    float *hp = sc.h0, *hn = sc.h1;
    sedmi::ProfileScope prof("shake_gru_steps", s, 2.0 * rows * 2 * 3 * GB_H * GB_H, 4.0 * rows * (6144.0 + 2048.0));
    for (int step = 0; step < T4; ++step) {
        hipLaunchKernelGGL(gru_big_step_kernel, dim3(GB_H / 32, static_cast<unsigned>(sedmi::ceil_div(n, 32)), 2), dim3(256),
                           lds, s, sc.gi, reinterpret_cast<const uint4 *>(m->d_whh), m->whh_oscale, m->d_bhn, hp, hn,
                           sc.rec, n, T4, step);
        std::swap(hp, hn);
    }
    SEDMI_CHECK_LAUNCH();
    return SEDMI_OK;
}

// SEDMI_SHAKE_ATT_PACKED=0: every attention workgroup splits its keys and values itself (the comparison path)
bool attention_packed_wanted() {
    const char *e = std::getenv("SEDMI_SHAKE_ATT_PACKED");
    return e == nullptr || e[0] != '0';
}

int run_encoder(const sedmi_shake_transformer *m, long n, int T4, const ShakeScratch &sc, hipStream_t s) {
    const long rows = n * T4;
    const unsigned ln_grid = static_cast<unsigned>(sedmi::ceil_div(rows, 4));
    int rc;
    if ((rc = launch_linear("shake_tf_fc1", m->fc1, sc.seq, sc.x, rows, false, s))) return rc;
    for (int i = 0; i < m->n_layers; ++i) {
        const EncoderLayerDev &L = m->layers[i];
        if ((rc = launch_linear("shake_tf_in_proj", L.in_proj, sc.x, sc.qkv, rows, false, s))) return rc;
        {
            sedmi::ProfileScope prof("shake_tf_attention", s, 4.0 * n * 8 * static_cast<double>(T4) * T4 * AT_D,
                                     4.0 * rows * 4096.0);
            const int S = sc.att_splits > 1 ? sc.att_splits : 1;
            const dim3 agrid(static_cast<unsigned>(sedmi::ceil_div(T4, 128) * S), 8, static_cast<unsigned>(n));
            const float qscale = 1.0f / std::sqrt(static_cast<float>(AT_D));
            // keys and values once per layer in the kernel's own operand layout when enough query tiles walk each block
            if (attention_packed_wanted() && sc.kvp != nullptr) {
                hipLaunchKernelGGL(attention_pack_kernel, dim3(static_cast<unsigned>(sedmi::ceil_div(T4, AT_KB)), 8, static_cast<unsigned>(n)),
                                   dim3(256), 0, s, sc.qkv, sc.kvp, T4);
                SEDMI_CHECK_LAUNCH();
                auto kern = attention_split_kernel<true>;
                if (int rc_lds = sedmi::set_max_dynamic_lds(reinterpret_cast<const void *>(kern), 2 * AT_BLOB)) return rc_lds;
                hipLaunchKernelGGL(kern, agrid, dim3(256), 2 * AT_BLOB, s, sc.qkv, sc.kvp, sc.att, T4, qscale, S, sc.part_o,
                                   sc.part_ml, rows);
            } else {
                auto kern = attention_split_kernel<false>;
                if (int rc_lds = sedmi::set_max_dynamic_lds(reinterpret_cast<const void *>(kern), AT_BLOB)) return rc_lds;
                hipLaunchKernelGGL(kern, agrid, dim3(256), AT_BLOB, s, sc.qkv, sc.kvp, sc.att, T4, qscale, S, sc.part_o,
                                   sc.part_ml, rows);
            }
            SEDMI_CHECK_LAUNCH();
            if (S > 1) {
                hipLaunchKernelGGL(attention_merge_kernel, dim3(static_cast<unsigned>(sedmi::ceil_div(rows * 256, 256))), dim3(256), 0,
                                   s, sc.part_o, sc.part_ml, sc.att, rows, S);
                SEDMI_CHECK_LAUNCH();
            }
        }
        if ((rc = launch_linear("shake_tf_out_proj", L.out_proj, sc.att, sc.ff, rows, false, s))) return rc;
        {
            sedmi::ProfileScope prof("shake_tf_add_ln", s, 8.0 * rows * 1024, 4.0 * rows * 3 * 1024.0);
            hipLaunchKernelGGL(add_ln_kernel, dim3(ln_grid), dim3(256), 0, s, sc.x, sc.ff, L.n1w, L.n1b, sc.x, rows);
            SEDMI_CHECK_LAUNCH();
        }
        if ((rc = launch_linear("shake_tf_ffn1", L.lin1, sc.x, sc.ff, rows, true, s))) return rc;
        if ((rc = launch_linear("shake_tf_ffn2", L.lin2, sc.ff, sc.att, rows, false, s))) return rc;
        {
            sedmi::ProfileScope prof("shake_tf_add_ln", s, 8.0 * rows * 1024, 4.0 * rows * 3 * 1024.0);
            hipLaunchKernelGGL(add_ln_kernel, dim3(ln_grid), dim3(256), 0, s, sc.x, sc.att, L.n2w, L.n2b, sc.x, rows);
            SEDMI_CHECK_LAUNCH();
        }
    }
    return SEDMI_OK;
}

// ---- host: weight repacking ---------------------------------------------------

int upload_floats(float **dev, const float *src, size_t n) {
    return sedmi::upload(dev, std::vector<float>(src, src + n));
}

// Linear weight [N][K] (row-major) -> split planes; kmap(k) = source column of GEMM k index
template <typename KMap>
int make_linear(LinearDev &L, const float *w, const float *bias, int N, int K, KMap kmap) {
    L.N = N;
    L.K = K;
    const int S = split_scale_exp(w, static_cast<size_t>(N) * K);
    L.oscale = std::ldexp(1.0f, -S);
    auto packed = pack_split(K / 16, N, S, [&](int ks, int k16, int n) {
        return w[static_cast<size_t>(n) * K + kmap(ks * 16 + k16)];
    });
    int rc;
    if ((rc = sedmi::upload(&L.w, packed))) return rc;
    return upload_floats(&L.b, bias, N);
}

void free_linear(LinearDev &L) {
    (void)hipFree(L.w);
    (void)hipFree(L.b);
    L = LinearDev();
}

int check_cnn_weights(const sedmi_shake_cnn_weights &c) {
    for (int i = 0; i < 3; ++i)
        SEDMI_REQUIRE(c.conv_w[i] && c.conv_b[i] && c.bn1_w[i] && c.bn1_b[i] && c.bn1_mean[i] && c.bn1_var[i] &&
                          c.br1_w[i] && c.br1_b[i] && c.br2_w[i] && c.br2_b[i] && c.bn2_w[i] && c.bn2_b[i] &&
                          c.bn2_mean[i] && c.bn2_var[i], "null weight pointer in block %d", i + 1);
    return SEDMI_OK;
}

int create_shake_cnn(ShakeCnnDev &dev, const sedmi_shake_cnn_weights &c) {
    int rc = SEDMI_OK;
    static const int cin[3] = {1, 16, 32}, cout[3] = {16, 32, 64};
    const float eps = 1e-5f;
    for (int i = 0; i < 3; ++i) {
        const int ci = cin[i], co = cout[i], n2 = 2 * co;
        std::vector<float> s1(n2), t1(n2), s2(co), t2(co);
        for (int k = 0; k < n2; ++k) {
            s1[k] = c.bn1_w[i][k] / std::sqrt(c.bn1_var[i][k] + eps);
            t1[k] = (c.conv_b[i][k] - c.bn1_mean[i][k]) * s1[k] + c.bn1_b[i][k];
        }
        for (int k = 0; k < co; ++k) {
            s2[k] = c.bn2_w[i][k] / std::sqrt(c.bn2_var[i][k] + eps);
            t2[k] = c.bn2_b[i][k] - c.bn2_mean[i][k] * s2[k];
        }
        BlockDev &b = dev.blk[i];
        if (i == 0) {
            ShakeConv1Params p{};
            for (int k = 0; k < 32; ++k) {
                for (int tap = 0; tap < 25; ++tap) p.w[k][tap] = c.conv_w[0][k * 25 + tap];
                p.s[k] = s1[k];
                p.t[k] = t1[k];
            }
            // matrix-core variant: row acc_row(r, half) is the value half (r < 8) or the gate half (r >= 8) of output
            // channel 8 * half + (r & 7); BatchNorm's scale folded in, gate rows times -log2(e)
            const double kNegLog2e = -1.4426950408889634;
            float rows[32][25], rshift[32], aw[16 * 25], bw[16 * 25];
            int na = 0, nbw = 0;
            for (int h = 0; h < 2; ++h)
                for (int r = 0; r < 16; ++r) {
                    const int row = acc_row(r, h), k = 8 * h + (r & 7) + (r < 8 ? 0 : 16);
                    const double f = r < 8 ? 1.0 : kNegLog2e;
                    for (int tap = 0; tap < 25; ++tap) {
                        rows[row][tap] = static_cast<float>(f * p.s[k] * p.w[k][tap]);
                        (r < 8 ? aw[na++] : bw[nbw++]) = rows[row][tap];
                    }
                    rshift[row] = static_cast<float>(f * p.t[k]);
                }
            auto is_gate = [](int row) { return (row & 16) != 0; };   // acc_row(r, h) >= 16 exactly for r >= 8
            const int Sa = split_scale_exp(aw, na), Sb = split_scale_exp(bw, nbw);
            auto c1s = pack_split(2, 32, 0, [&](int ks, int k16, int n) {
                const int k = 16 * ks + k16;
                return k < 25 ? std::ldexp(rows[n][k], is_gate(n) ? Sb : Sa) : 0.f;
            });
            std::vector<float> c1bias(32);
            for (int row = 0; row < 32; ++row) c1bias[row] = std::ldexp(rshift[row], is_gate(row) ? Sb : Sa);
            dev.c1_ascale = std::ldexp(1.0f, -Sa);
            dev.c1_bscale = std::ldexp(1.0f, -Sb);
            int rc1;
            if ((rc1 = sedmi::upload(&dev.d_c1s, c1s)) || (rc1 = sedmi::upload(&dev.d_c1bias, c1bias))) return rc1;
        } else {
            const int cch = ci >= 32 ? 32 : 16, ksp = cch / 16;
            const float *src = c.conv_w[i];
            const int S = split_scale_exp(src, static_cast<size_t>(n2) * ci * 9);
            auto wa = pack_split(9 * ci / 16, n2, S, [&](int kidx, int k16, int n) {
                const int ks = kidx % ksp, tap = (kidx / ksp) % 9, ch = (kidx / (ksp * 9)) * cch + ks * 16 + k16;
                return src[(static_cast<size_t>(n) * ci + ch) * 9 + tap];
            });
            std::vector<float> es(n2);
            for (int k = 0; k < n2; ++k) es[k] = std::ldexp(s1[k], -S);
            if ((rc = sedmi::upload(&b.wa, wa)) || (rc = sedmi::upload(&b.es_a, es)) || (rc = sedmi::upload(&b.et_a, t1)))
                return rc;
        }
        {   // both shake branches as one GEMM: columns [branch 1 | branch 2]
            const int cch = co >= 32 ? 32 : 16, ksp = cch / 16;
            const float *w1 = c.br1_w[i], *w2 = c.br2_w[i];
            const int S = std::min(split_scale_exp(w1, static_cast<size_t>(co) * co * 9),
                                   split_scale_exp(w2, static_cast<size_t>(co) * co * 9));
            auto wb = pack_split(9 * co / 16, n2, S, [&](int kidx, int k16, int n) {
                const int ks = kidx % ksp, tap = (kidx / ksp) % 9, ch = (kidx / (ksp * 9)) * cch + ks * 16 + k16;
                const float *src = n < co ? w1 : w2;
                return src[(static_cast<size_t>(n % co) * co + ch) * 9 + tap];
            });
            std::vector<float> es(n2, std::ldexp(1.0f, -S)), et(n2);
            for (int k = 0; k < co; ++k) {
                et[k] = c.br1_b[i][k];
                et[co + k] = c.br2_b[i][k];
            }
            if ((rc = sedmi::upload(&b.wb, wb)) || (rc = sedmi::upload(&b.es_b, es)) || (rc = sedmi::upload(&b.et_b, et)) ||
                (rc = sedmi::upload(&b.s2, s2)) || (rc = sedmi::upload(&b.t2, t2)))
                return rc;
        }
    }
    return SEDMI_OK;
}

void free_shake_cnn(ShakeCnnDev &dev) {
    (void)hipFree(dev.d_c1s);
    (void)hipFree(dev.d_c1bias);
    for (auto &b : dev.blk) {
        (void)hipFree(b.wa); (void)hipFree(b.wb); (void)hipFree(b.es_a); (void)hipFree(b.et_a);
        (void)hipFree(b.es_b); (void)hipFree(b.et_b); (void)hipFree(b.s2); (void)hipFree(b.t2);
    }
    dev = ShakeCnnDev();
}

// GEMM k index of the sequence is (freq, chan); the reference flattens (chan, freq): chan * 96 + freq
inline int seq_kmap(int k) { return (k % 64) * 96 + k / 64; }

int shake_check(const void *m, int64_t n, int64_t T) {
    SEDMI_REQUIRE(m != nullptr, "model is null");
    SEDMI_REQUIRE(n >= 0 && T >= 0 && T < (1 << 24) && n < (1LL << 31), "bad shape");
    return SEDMI_OK;
}

}  // namespace

extern "C" {

int sedmi_shake_rcnn_create(sedmi_shake_rcnn **out, const sedmi_shake_rcnn_weights *w) {
    SEDMI_REQUIRE(out && w, "null argument");
    *out = nullptr;
    SEDMI_REQUIRE(w->n_class > 0 && w->n_class <= 4096, "n_class %d out of range", w->n_class);
    int rc = check_cnn_weights(w->cnn);
    if (rc) return rc;
    for (int d = 0; d < 2; ++d)
        SEDMI_REQUIRE(w->gru_w_ih[d] && w->gru_w_hh[d] && w->gru_b_ih[d] && w->gru_b_hh[d], "null GRU weight pointer");
    SEDMI_REQUIRE(w->ln_w && w->ln_b && w->fc_w && w->fc_b, "null head weight pointer");

    auto *m = new sedmi_shake_rcnn();
    m->n_class = w->n_class;
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&m->num_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            m->num_cus = 0;             // unknown: the persistent recurrence is not used
    }
    auto fail = [&](int code) { sedmi_shake_rcnn_destroy(m); return code; };
    if ((rc = create_shake_cnn(m->cnn, w->cnn))) return fail(rc);
    {
        // input projection of both directions as one [6144][6144] matrix; b_hh of the r and z gates
        // folds into its bias, b_hn stays inside the reset gate product
        std::vector<float> wcat(static_cast<size_t>(6144) * 6144), bi(6144), bhn(2 * GB_H);
        for (int d = 0; d < 2; ++d) {
            std::memcpy(wcat.data() + static_cast<size_t>(d) * 3072 * 6144, w->gru_w_ih[d], sizeof(float) * 3072 * 6144);
            for (int n = 0; n < 3072; ++n) bi[d * 3072 + n] = w->gru_b_ih[d][n] + (n < 2 * GB_H ? w->gru_b_hh[d][n] : 0.f);
            for (int n = 0; n < GB_H; ++n) bhn[d * GB_H + n] = w->gru_b_hh[d][2 * GB_H + n];
        }
        if ((rc = make_linear(m->wih, wcat.data(), bi.data(), 6144, 6144, seq_kmap))) return fail(rc);
        int Sh = 100;
        for (int d = 0; d < 2; ++d) Sh = std::min(Sh, split_scale_exp(w->gru_w_hh[d], static_cast<size_t>(3072) * GB_H));
        std::vector<uint16_t> whh;
        for (int d = 0; d < 2; ++d) {
            const float *src = w->gru_w_hh[d];
            auto blk = pack_split(GB_H / 16, 3 * GB_H, Sh, [&](int ks, int k16, int n) {
                const int ug = n / 96, g = (n % 96) / 32, jj = n % 32;      // columns [unit group][gate][32]
                return src[static_cast<size_t>(g * GB_H + ug * 32 + jj) * GB_H + ks * 16 + k16];
            });
            whh.insert(whh.end(), blk.begin(), blk.end());
        }
        m->whh_oscale = std::ldexp(1.0f, -Sh);
        if ((rc = sedmi::upload(&m->d_whh, whh)) || (rc = sedmi::upload(&m->d_bhn, bhn)) ||
            (rc = upload_floats(&m->d_lw, w->ln_w, 2048)) || (rc = upload_floats(&m->d_lb, w->ln_b, 2048)) ||
            (rc = upload_floats(&m->d_fw, w->fc_w, static_cast<size_t>(w->n_class) * 2048)) ||
            (rc = upload_floats(&m->d_fb, w->fc_b, w->n_class)))
            return fail(rc);
    }
    *out = m;
    return SEDMI_OK;
}

int sedmi_shake_rcnn_destroy(sedmi_shake_rcnn *m) {
    if (!m) return SEDMI_OK;
    free_shake_cnn(m->cnn);
    free_linear(m->wih);
    (void)hipFree(m->d_whh); (void)hipFree(m->d_bhn);
    (void)hipFree(m->d_lw); (void)hipFree(m->d_lb); (void)hipFree(m->d_fw); (void)hipFree(m->d_fb);
    delete m;
    return SEDMI_OK;
}

size_t sedmi_shake_rcnn_workspace_bytes(const sedmi_shake_rcnn *m, int64_t n_clips, int64_t T) {
    if (!m || n_clips <= 0 || T < 4) return 0;
    return carve_shake(nullptr, n_clips, T, SHAKE_RCNN).bytes;
}

int sedmi_shake_cnn_f32(const sedmi_shake_rcnn *m, const float *feats, int64_t n, int64_t T, const float *alphas,
                        float *seq_ref_order, void *ws, size_t ws_bytes, void *stream) {
    int rc = shake_check(m, n, T);
    if (rc) return rc;
    if (n == 0 || T < 4) return SEDMI_OK;
    SEDMI_REQUIRE(feats && alphas && seq_ref_order && ws, "null pointer");
    ShakeScratch sc = carve_shake(ws, n, T, SHAKE_RCNN);
    SEDMI_REQUIRE(ws_bytes >= sc.bytes, "workspace too small: need %zu bytes, got %zu", sc.bytes, ws_bytes);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((rc = run_shake_cnn(&m->cnn, feats, n, static_cast<int>(T), alphas, sc.seq, sc, s))) return rc;
    const long rows = n * (T / 4);
    hipLaunchKernelGGL(to_reference_order_kernel, dim3(static_cast<unsigned>(sedmi::ceil_div(rows * 6144, 256))), dim3(256), 0,
                       s, sc.seq, seq_ref_order, rows);
    SEDMI_CHECK_LAUNCH();
    return SEDMI_OK;
}

int sedmi_shake_rcnn_forward_f32(const sedmi_shake_rcnn *m, const float *feats, int64_t n, int64_t T,
                                 const float *alphas, float *logits, void *ws, size_t ws_bytes, void *stream) {
    int rc = shake_check(m, n, T);
    if (rc) return rc;
    if (n == 0 || T < 4) return SEDMI_OK;
    SEDMI_REQUIRE(feats && alphas && logits && ws, "null pointer");
    ShakeScratch sc = carve_shake(ws, n, T, SHAKE_RCNN);
    SEDMI_REQUIRE(ws_bytes >= sc.bytes, "workspace too small: need %zu bytes, got %zu", sc.bytes, ws_bytes);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int T4 = static_cast<int>(T / 4);
    if ((rc = run_shake_cnn(&m->cnn, feats, n, static_cast<int>(T), alphas, sc.seq, sc, s))) return rc;
    if ((rc = run_shake_gru(m, sc.seq, n, T4, sc, s))) return rc;
    const long rows = n * T4;
    sedmi::ProfileScope prof("shake_ln_head", s, 2.0 * rows * 2048.0 * m->n_class, 4.0 * rows * (2048.0 + m->n_class));
    hipLaunchKernelGGL((ln_head_kernel<2048, true>), dim3(static_cast<unsigned>(sedmi::ceil_div(rows, 4))), dim3(256), 0, s,
                       sc.rec, m->d_lw, m->d_lb, m->d_fw, m->d_fb, logits, rows, m->n_class, sc.sync_words);
    SEDMI_CHECK_LAUNCH();
    return SEDMI_OK;
}

int sedmi_shake_rcnn_set_recurrence(sedmi_shake_rcnn *m, int mode) {
    SEDMI_REQUIRE(m != nullptr, "model is null");
    SEDMI_REQUIRE(mode == SEDMI_RECURRENCE_AUTO || mode == SEDMI_RECURRENCE_PER_STEP, "unknown recurrence mode %d", mode);
    m->recurrence = mode;
    return SEDMI_OK;
}

int sedmi_shake_rcnn_check(const sedmi_shake_rcnn *m, const void *ws, size_t ws_bytes, int64_t n, int64_t T, void *stream) {
    int rc = shake_check(m, n, T);
    if (rc) return rc;
    if (n == 0 || T < 4) return SEDMI_OK;
    SEDMI_REQUIRE(ws != nullptr, "null pointer");
    ShakeScratch sc = carve_shake(const_cast<void *>(ws), n, T, SHAKE_RCNN);
    SEDMI_REQUIRE(ws_bytes >= sc.bytes, "workspace too small: need %zu bytes, got %zu", sc.bytes, ws_bytes);
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned status = 0;
    SEDMI_CHECK_HIP(hipMemcpyAsync(&status, sc.sync_words, sizeof(status), hipMemcpyDeviceToHost, s));
    SEDMI_CHECK_HIP(hipStreamSynchronize(s));
    if (status != 0)
        return sedmi::set_error(SEDMI_ERECURRENCE,
                                "the single-launch GRU recurrence gave up waiting for a neighbouring workgroup (its 128 or 256 "
                                "workgroups must all be resident at once, one per compute unit: the GPU was shared with "
                                "other work, or compute units are masked); the logits of that forward are NaN. Run the "
                                "forward again with sedmi_shake_rcnn_set_recurrence(model, SEDMI_RECURRENCE_PER_STEP) "
                                "or on an otherwise idle GPU");
    return SEDMI_OK;
}

int sedmi_shake_transformer_create(sedmi_shake_transformer **out, const sedmi_shake_transformer_weights *w) {
    SEDMI_REQUIRE(out && w, "null argument");
    *out = nullptr;
    SEDMI_REQUIRE(w->n_class > 0 && w->n_class <= 4096, "n_class %d out of range", w->n_class);
    SEDMI_REQUIRE(w->n_layers >= 0 && w->n_layers <= kMaxEncoderLayers, "n_layers %d out of range (max %d)", w->n_layers,
                  kMaxEncoderLayers);
    int rc = check_cnn_weights(w->cnn);
    if (rc) return rc;
    SEDMI_REQUIRE(w->fc1_w && w->fc1_b && w->fc2_w && w->fc2_b, "null fc1 / fc2 weight pointer");
    for (int i = 0; i < w->n_layers; ++i) {
        const sedmi_shake_encoder_layer_weights &l = w->layers[i];
        SEDMI_REQUIRE(l.in_proj_w && l.in_proj_b && l.out_proj_w && l.out_proj_b && l.lin1_w && l.lin1_b && l.lin2_w &&
                          l.lin2_b && l.norm1_w && l.norm1_b && l.norm2_w && l.norm2_b, "null weight pointer in encoder layer %d", i);
    }
    auto *m = new sedmi_shake_transformer();
    m->n_class = w->n_class;
    m->n_layers = w->n_layers;
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&m->num_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            m->num_cus = 0;
    }
    auto fail = [&](int code) { sedmi_shake_transformer_destroy(m); return code; };
    auto ident = [](int k) { return k; };
    if ((rc = create_shake_cnn(m->cnn, w->cnn))) return fail(rc);
    if ((rc = make_linear(m->fc1, w->fc1_w, w->fc1_b, 1024, 6144, seq_kmap))) return fail(rc);
    for (int i = 0; i < w->n_layers; ++i) {
        const sedmi_shake_encoder_layer_weights &l = w->layers[i];
        EncoderLayerDev &L = m->layers[i];
        if ((rc = make_linear(L.in_proj, l.in_proj_w, l.in_proj_b, 3072, 1024, ident)) ||
            (rc = make_linear(L.out_proj, l.out_proj_w, l.out_proj_b, 1024, 1024, ident)) ||
            (rc = make_linear(L.lin1, l.lin1_w, l.lin1_b, 2048, 1024, ident)) ||
            (rc = make_linear(L.lin2, l.lin2_w, l.lin2_b, 1024, 2048, ident)) ||
            (rc = upload_floats(&L.n1w, l.norm1_w, 1024)) || (rc = upload_floats(&L.n1b, l.norm1_b, 1024)) ||
            (rc = upload_floats(&L.n2w, l.norm2_w, 1024)) || (rc = upload_floats(&L.n2b, l.norm2_b, 1024)))
            return fail(rc);
    }
    if ((rc = upload_floats(&m->d_fw, w->fc2_w, static_cast<size_t>(w->n_class) * 1024)) ||
        (rc = upload_floats(&m->d_fb, w->fc2_b, w->n_class)))
        return fail(rc);
    *out = m;
    return SEDMI_OK;
}

int sedmi_shake_transformer_destroy(sedmi_shake_transformer *m) {
    if (!m) return SEDMI_OK;
    free_shake_cnn(m->cnn);
    free_linear(m->fc1);
    for (auto &L : m->layers) {
        free_linear(L.in_proj); free_linear(L.out_proj); free_linear(L.lin1); free_linear(L.lin2);
        (void)hipFree(L.n1w); (void)hipFree(L.n1b); (void)hipFree(L.n2w); (void)hipFree(L.n2b);
    }
    (void)hipFree(m->d_fw); (void)hipFree(m->d_fb);
    delete m;
    return SEDMI_OK;
}

size_t sedmi_shake_transformer_workspace_bytes(const sedmi_shake_transformer *m, int64_t n_clips, int64_t T) {
    if (!m || n_clips <= 0 || T < 4) return 0;
    return carve_shake(nullptr, n_clips, T, SHAKE_TRANSFORMER, m->num_cus).bytes;
}

int sedmi_shake_transformer_forward_f32(const sedmi_shake_transformer *m, const float *feats, int64_t n, int64_t T,
                                        const float *alphas, float *logits, void *ws, size_t ws_bytes, void *stream) {
    int rc = shake_check(m, n, T);
    if (rc) return rc;
    if (n == 0 || T < 4) return SEDMI_OK;
    SEDMI_REQUIRE(feats && alphas && logits && ws, "null pointer");
    SEDMI_REQUIRE(n <= 65535, "at most 65535 clips per call");
    ShakeScratch sc = carve_shake(ws, n, T, SHAKE_TRANSFORMER, m->num_cus);
    SEDMI_REQUIRE(ws_bytes >= sc.bytes, "workspace too small: need %zu bytes, got %zu", sc.bytes, ws_bytes);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int T4 = static_cast<int>(T / 4);
    if ((rc = run_shake_cnn(&m->cnn, feats, n, static_cast<int>(T), alphas, sc.seq, sc, s))) return rc;
    if ((rc = run_encoder(m, n, T4, sc, s))) return rc;
    const long rows = n * T4;
    sedmi::ProfileScope prof("shake_tf_head", s, 2.0 * rows * 1024.0 * m->n_class, 4.0 * rows * (1024.0 + m->n_class));
    hipLaunchKernelGGL((ln_head_kernel<1024, false>), dim3(static_cast<unsigned>(sedmi::ceil_div(rows, 4))), dim3(256), 0, s,
                       sc.x, nullptr, nullptr, m->d_fw, m->d_fb, logits, rows, m->n_class, nullptr);
    SEDMI_CHECK_LAUNCH();
    return SEDMI_OK;
}

// ---------------------------------------------------------------------------
// BasicRCNN
// ---------------------------------------------------------------------------
static constexpr int kBasicChunk = 32;      // clips per pass through the conv stack (16 MB of activations per clip)

struct BasicScratch {
    float *a, *b, *seq, *gi;
    size_t bytes;
};

static BasicScratch carve_basic(void *ws, const sedmi_basic_rcnn *m, long n, long T) {
    BasicScratch sc{};
    size_t off = 0;
    auto take = [&](size_t n_floats) {
        float *p = reinterpret_cast<float *>(static_cast<char *>(ws) + off);
        off += sedmi::align_up(n_floats * sizeof(float), 256);
        return p;
    };
    const long ch = std::min<long>(n, kBasicChunk), T2 = T / 2;
    sc.a = take(static_cast<size_t>(ch) * T * m->F * BR_C);
    sc.b = take(static_cast<size_t>(ch) * T * m->F * BR_C);
    sc.seq = take(static_cast<size_t>(n) * T2 * m->K);
    sc.gi = take(static_cast<size_t>(n) * T2 * 128);
    sc.bytes = off;
    return sc;
}

int sedmi_basic_rcnn_create(sedmi_basic_rcnn **out, const sedmi_basic_rcnn_weights *w) {
    SEDMI_REQUIRE(out && w, "null argument");
    *out = nullptr;
    SEDMI_REQUIRE(w->n_class > 0 && w->n_class <= 4096, "n_class %d out of range", w->n_class);
    // odd widths are floored by the last max-pool like the reference's ((F - 8) // 2, models.py:36)
    SEDMI_REQUIRE(w->frequency_dim >= 10 && w->frequency_dim <= 4096, "frequency_dim %d: need 10 .. 4096", w->frequency_dim);
    for (int i = 0; i < 3; ++i) SEDMI_REQUIRE(w->conv_w[i] && w->conv_b[i], "null conv weight pointer");
    SEDMI_REQUIRE(w->gru_w_ih && w->gru_w_hh && w->gru_b_ih && w->gru_b_hh && w->fc_w && w->fc_b, "null weight pointer");
    auto *m = new sedmi_basic_rcnn();
    m->n_class = w->n_class;
    m->F = w->frequency_dim;
    m->K = BR_C * ((m->F - 8) / 2);
    int rc;
    auto fail = [&](int code) { sedmi_basic_rcnn_destroy(m); return code; };
    if ((rc = upload_floats(&m->d_w1, w->conv_w[0], BR_C * 9)) || (rc = upload_floats(&m->d_b1, w->conv_b[0], BR_C)))
        return fail(rc);
    for (int l = 0; l < 2; ++l) {
        const float *src = w->conv_w[l + 1];
        const int S = split_scale_exp(src, static_cast<size_t>(BR_C) * BR_C * 9);
        auto packed = pack_split(9 * BR_C / 16, BR_C, S, [&](int kidx, int k16, int n) {
            const int ks = kidx % 2, tap = (kidx / 2) % 9, ch = (kidx / 18) * 32 + ks * 16 + k16;   // chunk / tap / k
            return src[(static_cast<size_t>(n) * BR_C + ch) * 9 + tap];
        });
        std::vector<float> es(BR_C, std::ldexp(1.0f, -S));
        if ((rc = sedmi::upload(&m->d_wc[l], packed)) || (rc = sedmi::upload(&m->d_es[l], es)) ||
            (rc = upload_floats(&m->d_et[l], w->conv_b[l + 1], BR_C)))
            return fail(rc);
    }
    {
        std::vector<float> wpad(static_cast<size_t>(128) * m->K, 0.f), bi(128, 0.f);
        std::memcpy(wpad.data(), w->gru_w_ih, sizeof(float) * 3 * BR_H * m->K);
        for (int n = 0; n < 3 * BR_H; ++n) bi[n] = w->gru_b_ih[n] + (n < 2 * BR_H ? w->gru_b_hh[n] : 0.f);
        if ((rc = make_linear(m->wih, wpad.data(), bi.data(), 128, m->K, [](int k) { return k; }))) return fail(rc);
    }
    if ((rc = upload_floats(&m->d_whh, w->gru_w_hh, 3 * BR_H * BR_H)) ||
        (rc = upload_floats(&m->d_bhn, w->gru_b_hh + 2 * BR_H, BR_H)) ||
        (rc = upload_floats(&m->d_fw, w->fc_w, static_cast<size_t>(w->n_class) * BR_H)) ||
        (rc = upload_floats(&m->d_fb, w->fc_b, w->n_class)))
        return fail(rc);
    *out = m;
    return SEDMI_OK;
}

int sedmi_basic_rcnn_destroy(sedmi_basic_rcnn *m) {
    if (!m) return SEDMI_OK;
    (void)hipFree(m->d_w1); (void)hipFree(m->d_b1);
    for (int l = 0; l < 2; ++l) { (void)hipFree(m->d_wc[l]); (void)hipFree(m->d_es[l]); (void)hipFree(m->d_et[l]); }
    free_linear(m->wih);
    (void)hipFree(m->d_whh); (void)hipFree(m->d_bhn); (void)hipFree(m->d_fw); (void)hipFree(m->d_fb);
    delete m;
    return SEDMI_OK;
}

size_t sedmi_basic_rcnn_workspace_bytes(const sedmi_basic_rcnn *m, int64_t n_clips, int64_t T) {
    if (!m || n_clips <= 0 || T < 2) return 0;
    return carve_basic(nullptr, m, n_clips, T).bytes;
}

int sedmi_basic_rcnn_forward_f32(const sedmi_basic_rcnn *m, const float *feats, int64_t n, int64_t T, float *logits,
                                 void *ws, size_t ws_bytes, void *stream) {
    SEDMI_REQUIRE(m != nullptr, "model is null");
    SEDMI_REQUIRE(n >= 0 && T >= 0 && T < (1 << 24) && n < (1LL << 31), "bad shape");
    if (n == 0 || T < 2) return SEDMI_OK;
    SEDMI_REQUIRE(feats && logits && ws, "null pointer");
    BasicScratch sc = carve_basic(ws, m, n, T);
    SEDMI_REQUIRE(ws_bytes >= sc.bytes, "workspace too small: need %zu bytes, got %zu", sc.bytes, ws_bytes);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int F = m->F, Ti = static_cast<int>(T), T2 = Ti / 2;
    int rc;
    auto blocks = [](long total) { return dim3(static_cast<unsigned>(sedmi::ceil_div(total, 256))); };
    for (long c0 = 0; c0 < n; c0 += kBasicChunk) {
        const long nc = std::min<long>(kBasicChunk, n - c0);
        const long rows = nc * Ti;
        {
            sedmi::ProfileScope prof("basic_conv1_pool", s, 2.0 * 9 * BR_C * rows * F, 4.0 * rows * (F + (F - 4.0) * BR_C));
            const long n_runs = sedmi::ceil_div(F - 4, BC1_RUN);
            hipLaunchKernelGGL(basic_conv1_pool_kernel, blocks(rows * n_runs * (BR_C / 4)), dim3(256), 0, s,
                               feats + c0 * static_cast<long>(Ti) * F, m->d_w1, m->d_b1, sc.b, rows, Ti, F);
            SEDMI_CHECK_LAUNCH();
        }
        // conv2 with its (1,5) max-pool + ReLU in the epilogue
        if ((rc = launch_conv2d<128, 128, 32, MODE_POOL_W5, 1, 1, C2_WN>("basic_conv<128,128>", m->d_wc[0], m->d_es[0], m->d_et[0], 0.f,
                                                                 nullptr, nullptr, sc.b, sc.a, nc, Ti, F - 4, s))) return rc;
        // conv3 with MaxPool2d(2, 2) + ReLU and the reference's memory order in its epilogue
        if ((rc = launch_conv2d<128, 128, 32, MODE_POOL_NCHW, 2, 2, C2_WN>("basic_conv<128,128>", m->d_wc[1], m->d_es[1], m->d_et[1],
                                                                       0.f, nullptr, nullptr, sc.a,
                                                                       sc.seq + c0 * static_cast<long>(T2) * m->K, nc, Ti,
                                                                       F - 8, s))) return rc;
    }
    const long grows = n * T2;
    if ((rc = launch_linear("basic_gru_input_proj", m->wih, sc.seq, sc.gi, grows, false, s))) return rc;
    sedmi::ProfileScope prof("basic_gru_head", s, 2.0 * grows * (3.0 * BR_H * BR_H + BR_H * m->n_class),
                             4.0 * grows * (128.0 + m->n_class));
    hipLaunchKernelGGL(basic_gru_head_kernel, dim3(static_cast<unsigned>(n)), dim3(64), 0, s, sc.gi, m->d_whh, m->d_bhn,
                       m->d_fw, m->d_fb, logits, T2, m->n_class);
    SEDMI_CHECK_LAUNCH();
    return SEDMI_OK;
}

}  // extern "C"
