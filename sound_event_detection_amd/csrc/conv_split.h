// Conv block (Conv3x3 + BatchNorm affine + GLU + AvgPool(1,2)) on the f16 matrix
// cores with float32-grade accuracy: every float32 operand x is split into
// x = hi + lo, hi = f16(x), lo = f16(x - hi) (22 significand bits together), and
// each product is formed as  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  by three
// v_mfma_f32_32x32x16_f16 with float32 accumulation; the dropped lo*lo term is
// 2^-22 relative.  That is 3 MFMAs at 16x the f32-MFMA rate = 5.3x the
// throughput of v_mfma_f32_32x32x2_f32 for the same contraction.
//
// Range: f16 tops out at 65504 and loses lo bits below 2^-14*2^11, so weights
// are pre-multiplied on the host by a power of two that puts the layer's largest
// |w| in [64, 128) (undone exactly inside the BatchNorm scale), and activations
// are used as they are (BatchNorm keeps them O(1..100); |x| >= 65504 saturates).
//
// Include after mfma_common.h, inside the same anonymous namespace (uses f32x16, acc_row,
// sigmoidf_fast from there).
#pragma once

#ifndef SPLIT_BDIST
#define SPLIT_BDIST 1
#endif
#ifndef SPLIT_GLU_BDIST
#define SPLIT_GLU_BDIST 1     // same, for the GEMM of the GLU phases
#endif
#ifndef COLUMN_TILES
#define COLUMN_TILES 1        // 0: the narrow blocks keep position-major m-tiles (the comparison build)
#endif
#ifdef SPLIT_TIMING
// diagnostic build only: per-layer sums of s_memtime ticks spent in each section of conv_split_kernel
// (wave 0 of every workgroup), read back through sedmi_debug_split_ticks
__device__ unsigned long long g_split_ticks[6][12];   // [9] tiles, [10] min start, [11] max end
#define SPLIT_TICK(sec)                                                 \
    do {                                                                \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
        tick_acc[sec] += now_ - tick_last;                              \
        tick_last = now_;                                               \
    } while (0)
#else
#define SPLIT_TICK(sec) do { } while (0)
#endif
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;

__device__ __forceinline__ void split_f16(float x, _Float16 &hi, _Float16 &lo) {
    x = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
    hi = static_cast<_Float16>(x);
    lo = static_cast<_Float16>(x - static_cast<float>(hi));
}

// Two values at once with the packed round-toward-zero convert: hi is then truncated rather than
// rounded, which costs one bit (hi + lo still carries >= 21 significand bits) and halves the
// conversion instructions.  Round-toward-zero never overflows to infinity (a finite input beyond
// the f16 range converts to +-65504), so no clamp is needed: finite inputs give finite halves.
// Returns the packed f16 pairs as 32-bit words.
__device__ __forceinline__ void split2_f16(float x0, float x1, unsigned &hi, unsigned &lo) {
    const auto h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    const auto l = __builtin_amdgcn_cvt_pkrtz(x0 - static_cast<float>(h[0]), x1 - static_cast<float>(h[1]));
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}

// CIN/COUT/W: layer geometry (input width W, output width W/2).
// WAVES_M x WAVES_N = 4 waves; each wave owns MT x NT tiles of 32x32.
// CCH: input channels staged per pass through LDS.
// XT: workgroup tiles across the width (1: a tile spans the image; 2: a tile is one half of it, column tiles only).
template <int CIN, int COUT, int W, int WAVES_M, int MT, int CCH, int XT = 1>
struct SplitGeom {
    static constexpr int WT = W / XT;                    // columns per workgroup tile
    static constexpr int WAVES_N = 4 / WAVES_M;
    static constexpr int NT = COUT / (32 * WAVES_N);
    static constexpr int M = 32 * MT * WAVES_M;          // positions per workgroup
    static constexpr int TH = M / WT;                    // rows per workgroup
    // Column tiles (workgroup tiles of up to 8 columns): an m-tile is 32 consecutive ROWS of ONE column (m-tile g of
    // the workgroup = column g % WT of row group g / WT) instead of 32 consecutive positions.  The tile of an image
    // edge column then skips the taps that would read the zero padding beside the image -- a third of the conv's
    // MFMAs at W = 2, a sixth at W = 4, a twelfth at W = 8 -- and the pooled pair of columns is the same register of
    // two of the lane's accumulators.  No tap reads a padding column any more, so the planes keep a single (never
    // read) one: rows are W + 1 positions apart, an odd number of 16-byte units, which keeps the row-strided
    // fragment reads conflict-free.
    // (MT % WT == 0: which tiles are edge tiles is then the same in every wave, and known at compile time)
    static constexpr bool COLT = WT <= 8 && MT % WT == 0 && TH % 32 == 0 && COLUMN_TILES;
    // XT == 2: a half's inner neighbour column is real data (the other half's edge column) and both halves use one
    // layout: WT + 2 positions, plus one of padding where the row would otherwise be an even number of 16-byte units
    static constexpr int TW = !COLT ? W + 2 : XT == 1 ? W + 1 : (((WT + 2) * (CCH * 2 + 16) / 16) % 2 ? WT + 2 : WT + 3);
    static constexpr int PSB = CCH * 2 + 16;             // bytes per position per plane (16 B pad)
    static constexpr int PLANE_A = (TH + 2) * TW * PSB;
    static constexpr int ZSB = COUT * 2 + 16;
    // m-tiles per wave that go through the GLU together: one where the accumulators already fill the
    // register file (two would spill ~90 B per lane, and spills are written through to HBM), else two
    static constexpr int MG = NT == 1 ? (MT >= 2 ? 2 : 1) : ((MT >= 4 || MT < 2) ? 1 : 2);
    static constexpr int ZROWS = WAVES_M * 32 * MG;      // rows of the z planes (one GLU phase)
    static constexpr int PLANE_Z = ZROWS * ZSB;
    static constexpr int NPHASE = MT / MG;               // GLU phases; their z planes are double-buffered
    static constexpr int Z_BYTES = (NPHASE > 1 ? 4 : 2) * PLANE_Z;
    static constexpr int LDS_BYTES = 2 * PLANE_A > Z_BYTES ? 2 * PLANE_A : Z_BYTES;
    static constexpr int NCHUNK = CIN / CCH;
    static constexpr int KS = CCH / 16;                  // k-steps per tap per chunk
    static constexpr int KSTEPS = 9 * CIN / 16;
    static_assert(WAVES_M * WAVES_N == 4 && NT >= 1 && M % WT == 0 && CIN % CCH == 0 && CCH % 16 == 0 &&
                      COUT % 32 == 0 && W % 2 == 0 && LDS_BYTES <= 160 * 1024 && (XT == 1 || (XT == 2 && COLT && WT % 2 == 0)),
                  "unsupported split-conv geometry");
};

// B (weight) fragments of one k-step for NT column tiles: hi and lo planes.
template <int NT>
struct SplitB {
    f16x8 h[NT], l[NT];
};

// b: packed weights of one k-step, [plane][N][16] f16 as uint4 (16 B) units, already offset to
// this lane's (column, half) of n-tile 0; n2 = 2 * N (uint4 units between the hi and lo planes).
// CTS: distance between a wave's column tiles, in tiles (1 = neighbouring tiles).
template <int NT, int CTS = 1>
__device__ __forceinline__ void split_load_b(SplitB<NT> &f, const uint4 *b, int n2) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const uint4 h = b[nt * 64 * CTS], l = b[n2 + nt * 64 * CTS];
        f.h[nt] = __builtin_bit_cast(f16x8, h);
        f.l[nt] = __builtin_bit_cast(f16x8, l);
    }
}

// One k-step.  The A fragments of m-tile 0 arrive in (ph, pl), read from LDS during the previous
// step; the other m-tiles' fragments are read here (this lane's rows in the hi / lo planes, pointers
// already offset by 16*(lane>>5)); the weights come in registers.  Before the last m-tile's MFMAs are
// issued, m-tile 0 of the NEXT step (LDS offset next_off, < 0 for none) is requested into (ph, pl),
// so that its LDS latency is covered by those MFMAs rather than paid at the head of the next step.
//
// hook(slot) runs after every MFMA (slot = running MFMA index) and is
// pinned there: vector-ALU work issued between the MFMAs of a wave overlaps with the MFMAs of the other
// wave on the SIMD, while a long run of pure vector work next to a long run of MFMAs does not.
// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits until every global store of
// the wave has been acknowledged, which puts an HBM write latency into each GLU phase.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

struct SplitNoHook {
    __device__ __forceinline__ void operator()(int) const {}
};
template <int MT, int NT, typename Hook = SplitNoHook>
__device__ __forceinline__ void split_step(f32x16 (&acc)[MT][NT], const char *(&ah)[MT], const char *(&al)[MT],
                                           int a_off, int next_off, const SplitB<NT> &b, f16x8 &ph, f16x8 &pl,
                                           int slot0 = 0, Hook hook = Hook()) {
    constexpr bool HOOKED = !__is_same(Hook, SplitNoHook);
    f16x8 vah[MT], val[MT];
    vah[0] = ph;
    val[0] = pl;
#pragma unroll
    for (int mt = 1; mt < MT; ++mt) {
        vah[mt] = *reinterpret_cast<const f16x8 *>(ah[mt] + a_off);
        val[mt] = *reinterpret_cast<const f16x8 *>(al[mt] + a_off);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (mt == MT - 1) {
            if (MT > 1) __builtin_amdgcn_sched_barrier(0);
            if (next_off >= 0) {
                ph = *reinterpret_cast<const f16x8 *>(ah[0] + next_off);
                pl = *reinterpret_cast<const f16x8 *>(al[0] + next_off);
            }
            if (MT > 1) __builtin_amdgcn_sched_barrier(0);
        }
        // the three products of a tile are NT MFMAs apart: consecutive MFMAs never wait for each other's result
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w == 0 ? val[mt] : vah[mt], w == 1 ? b.l[nt] : b.h[nt],
                                                                     acc[mt][nt], 0, 0, 0);
                if (HOOKED) { hook(slot0 + (3 * mt + w) * NT + nt); __builtin_amdgcn_sched_barrier(0); }
            }
    }
}

// acc += A . B over NSTEPS consecutive k-steps.  The weights of step i+1 are requested from L2
// before the MFMAs of step i issue (their latency is several hundred cycles).
// a_off(i) = LDS byte offset of step i relative to ah/al; N = columns of the packed weight matrix
// (B advances by 4*N uint4 per k-step: 2 planes x N columns x 32 B).
template <int MT, int NT, int NSTEPS, int D = SPLIT_BDIST, int CTS = 1, typename AOff, typename Hook = SplitNoHook>
__device__ __forceinline__ void split_gemm(f32x16 (&acc)[MT][NT], const char *(&ah)[MT], const char *(&al)[MT],
                                           const uint4 *b, int N, AOff a_off, Hook hook = Hook()) {
    const int BS = N * 4, n2 = N * 2;
    // weights are requested SPLIT_BDIST k-steps ahead of their use, through a ring of register buffers
    constexpr int NB = D + 1;
    SplitB<NT> bb[NB];
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < NSTEPS) split_load_b<NT, CTS>(bb[d], b + d * BS, n2);
    f16x8 ph = *reinterpret_cast<const f16x8 *>(ah[0] + a_off(0));
    f16x8 pl = *reinterpret_cast<const f16x8 *>(al[0] + a_off(0));
#pragma unroll
    for (int i = 0; i < NSTEPS; ++i) {
        // sched_barrier pins "request the later step's weights, then run this step": left alone, the
        // scheduler sinks every load to just above its first use and the wave stalls on each one
        if (i + D < NSTEPS) split_load_b<NT, CTS>(bb[(i + D) % NB], b + (i + D) * BS, n2);
        __builtin_amdgcn_sched_barrier(0);
        split_step<MT, NT, Hook>(acc, ah, al, a_off(i), i + 1 < NSTEPS ? a_off(i + 1) : -1, bb[i % NB], ph, pl,
                                 i * 3 * MT * NT, hook);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The same with m-tiles that sit some k-steps out: use(i, mt) says whether m-tile mt takes part in step i (every step
// has at least one that does).  The fragment requested one step ahead is that of the step's FIRST participating
// m-tile, and the request goes out in front of the MFMAs of the current step's LAST one.
template <int MT, int NT, int NSTEPS, int D = SPLIT_BDIST, typename AOff, typename Use>
__device__ __forceinline__ void split_gemm_some(f32x16 (&acc)[MT][NT], const char *(&ah)[MT], const char *(&al)[MT],
                                                const uint4 *b, int N, AOff a_off, Use use) {
    const int BS = N * 4, n2 = N * 2;
    constexpr int NB = D + 1;
    SplitB<NT> bb[NB];
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < NSTEPS) split_load_b<NT>(bb[d], b + d * BS, n2);
    auto first_of = [&](int i) {
        int f = 0;
#pragma unroll
        for (int mt = MT - 1; mt >= 0; --mt)
            if (use(i, mt)) f = mt;
        return f;
    };
    auto last_of = [&](int i) {
        int l = 0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            if (use(i, mt)) l = mt;
        return l;
    };
    f16x8 ph = *reinterpret_cast<const f16x8 *>(ah[first_of(0)] + a_off(0));
    f16x8 pl = *reinterpret_cast<const f16x8 *>(al[first_of(0)] + a_off(0));
#pragma unroll
    for (int i = 0; i < NSTEPS; ++i) {
        if (i + D < NSTEPS) split_load_b<NT>(bb[(i + D) % NB], b + (i + D) * BS, n2);
        __builtin_amdgcn_sched_barrier(0);
        const SplitB<NT> &w = bb[i % NB];
        const int first = first_of(i), last = last_of(i);
        f16x8 vah[MT], val[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (!use(i, mt)) continue;
            if (mt == first) {
                vah[mt] = ph;
                val[mt] = pl;
            } else {
                vah[mt] = *reinterpret_cast<const f16x8 *>(ah[mt] + a_off(i));
                val[mt] = *reinterpret_cast<const f16x8 *>(al[mt] + a_off(i));
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (!use(i, mt)) continue;
            if (mt == last) {
                if (last != first) __builtin_amdgcn_sched_barrier(0);
                if (i + 1 < NSTEPS) {
                    ph = *reinterpret_cast<const f16x8 *>(ah[first_of(i + 1)] + a_off(i + 1));
                    pl = *reinterpret_cast<const f16x8 *>(al[first_of(i + 1)] + a_off(i + 1));
                }
                if (last != first) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q == 0 ? val[mt] : vah[mt], q == 1 ? w.l[nt] : w.h[nt],
                                                                         acc[mt][nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// XI: which of the XT tiles across the width this launch works on (a launch per half: which column is the image's
// edge is then a compile-time fact).
template <int CIN, int COUT, int W, int WAVES_M, int MT, int CCH, int XT = 1, int XI = 0>
__global__ __launch_bounds__(256, 2) void conv_split_kernel(
    const float *__restrict__ x,       // [clip][H][W][CIN] f32
    const uint4 *__restrict__ wp,      // packed split weights [KSTEPS][2][COUT][16] f16, order chunk/tap/k
    const float *__restrict__ bn_s, const float *__restrict__ bn_t,   // z = acc * s + t (weight scale folded into s)
    const uint4 *__restrict__ gp,      // packed split GLU weights [COUT/16][2][COUT][16] f16
    const float *__restrict__ g_s, const float *__restrict__ gb,      // lin / 2 = acc * g_s + gb (pool factor folded in)
    float *__restrict__ y,             // [clip][H][W/2][COUT] f32
    int H, int tiles_per_clip, unsigned n_tiles) {
    using G = SplitGeom<CIN, COUT, W, WAVES_M, MT, CCH, XT>;
    constexpr int NT = G::NT, WT = G::WT, X0 = XI * WT;      // X0: the tile's first image column
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *plane_h = smem, *plane_l = smem + G::LDS_BYTES / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, col32 = lane & 31;
    const int wm = wave / G::WAVES_N, wn = wave % G::WAVES_N;
    // Workgroup b runs on XCD b % 8 (round-robin dispatch), each XCD has its own L2, and vertically
    // neighbouring tiles share two halo rows: give every XCD a contiguous range of tiles, so that
    // workgroups b and b + 8, which it runs back to back, are neighbours and the halo comes from L2.
    const unsigned per_xcd = (gridDim.x + 7) / 8;
    const unsigned tile = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (tile >= n_tiles) return;
#ifdef SPLIT_TIMING
    unsigned long long tick_acc[10] = {}, tick_last = __builtin_amdgcn_s_memtime();
    const unsigned long long tick_first = tick_last;
#endif
    const long clip = tile / tiles_per_clip;
    const int h0 = static_cast<int>(tile % tiles_per_clip) * G::TH;
    const float *xc = x + clip * static_cast<long>(H) * W * CIN;

    // zero the halo columns once (staging never writes them; with column tiles nothing reads them)
    for (int i = tid; !G::COLT && i < (G::TH + 2) * 2 * (G::PSB / 16); i += 256) {
        const int rr = i / (2 * (G::PSB / 16)), rem = i - rr * 2 * (G::PSB / 16);
        const int side = rem / (G::PSB / 16), o = (rem - side * (G::PSB / 16)) * 16;
        const int off = (rr * G::TW + side * (W + 1)) * G::PSB + o;
        *reinterpret_cast<uint4 *>(plane_h + off) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4 *>(plane_l + off) = make_uint4(0, 0, 0, 0);
    }

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
        const int pos = 32 * MT * wm + 32 * mt + col32;
        int off = ((pos / W) * G::TW + pos % W) * G::PSB + 16 * half;
        // column tiles: row 32 * (g / WT) + lane of the tile's column g % WT, g = MT * wm + mt
        if (G::COLT) off = ((32 * ((MT * wm + mt) / WT) + col32) * G::TW + (MT * wm + mt) % WT) * G::PSB + 16 * half;
        ah[mt] = plane_h + off;
        al[mt] = plane_l + off;
    }
    const int ncol = wn * 32 * NT + col32;
    const uint4 *bp = wp + ncol * 2 + half;

    // Staging map: 256 threads fetch RPJ whole rows of a chunk per pass, a thread the same 16-byte piece of its row
    // in every pass and chunk, so everything that depends on the thread is computed once per tile: a load costs one
    // add, an LDS store none (its offset is an immediate).  All of a thread's loads of a chunk are issued before the
    // first one is consumed, and the loads of chunk c + 1 before the MFMA loop of chunk c: only the first chunk's
    // latency is exposed.  The clip's input sits behind a buffer descriptor that spans exactly the clip: rows above
    // it (negative offsets wrap to huge unsigned ones) and below it fall outside and read as zero -- the conv's
    // padding -- with no branch and no 64-bit address arithmetic.
    constexpr int ROW_F4 = WT * CCH / 4;
    static_assert(ROW_F4 <= 256 && 256 % ROW_F4 == 0, "a pass covers whole rows");
    constexpr int RPJ = 256 / ROW_F4, ROWS = G::TH + 2;
    constexpr int NLD = (ROWS + RPJ - 1) / RPJ;
    using f32x4v = __attribute__((ext_vector_type(4))) float;
    f32x4v v[NLD];
    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xc), 0, H * W * CIN * 4, 0x00020000);
    const int rr0 = tid / ROW_F4, e0 = (tid % ROW_F4) * 4, w0 = e0 / CCH, ch0 = e0 % CCH;
    const unsigned g_base = static_cast<unsigned>(((h0 - 1 + rr0) * W + X0 + w0) * CIN + ch0) * 4u;
    const int l_base = (rr0 * G::TW + 1 + w0) * G::PSB + ch0 * 2;
    // the last pass may reach past the staged rows: those threads read nothing (an offset out of range) and write nothing
    constexpr bool RAGGED = NLD * RPJ > ROWS;
    const bool in_last = !RAGGED || rr0 < ROWS - (NLD - 1) * RPJ;
    // XT == 2: the column of the other half that borders this one, in passes of 256 / (CCH / 4) rows
    constexpr int EQ = CCH / 4, ERP = 256 / EQ, NE = XT == 2 ? (ROWS + ERP - 1) / ERP : 0;
    f32x4v ve[NE > 0 ? NE : 1];
    const int er0 = tid / EQ, ech = (tid % EQ) * 4;
    const unsigned ge_base =
        static_cast<unsigned>(((h0 - 1 + er0) * W + (XI == 0 ? X0 + WT : X0 - 1)) * CIN + ech) * 4u;
    const int le_base = (er0 * G::TW + (XI == 0 ? WT + 1 : 0)) * G::PSB + ech * 2;
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            unsigned off = g_base + static_cast<unsigned>(c * CCH * 4 + j * RPJ * W * CIN * 4);
            if (RAGGED && j == NLD - 1 && !in_last) off = 0xfffffff0u;
            v[j] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(xrs, static_cast<int>(off), 0, 0));
        }
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            unsigned off = ge_base + static_cast<unsigned>(c * CCH * 4 + j * ERP * W * CIN * 4);
            if (er0 + j * ERP >= ROWS) off = 0xfffffff0u;
            ve[j] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(xrs, static_cast<int>(off), 0, 0));
        }
    };
    load_chunk(0);
    SPLIT_TICK(0);
    for (int c = 0; c < G::NCHUNK; ++c) {
        if (c) __syncthreads();                        // every wave is done reading the previous chunk
        SPLIT_TICK(1);
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            uint2 hi, lo;
            split2_f16(v[j][0], v[j][1], hi.x, lo.x);
            split2_f16(v[j][2], v[j][3], hi.y, lo.y);
            if (!RAGGED || j < NLD - 1 || in_last) {
                *reinterpret_cast<uint2 *>(plane_h + l_base + j * RPJ * G::TW * G::PSB) = hi;
                *reinterpret_cast<uint2 *>(plane_l + l_base + j * RPJ * G::TW * G::PSB) = lo;
            }
        }
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            uint2 hi, lo;
            split2_f16(ve[j][0], ve[j][1], hi.x, lo.x);
            split2_f16(ve[j][2], ve[j][3], hi.y, lo.y);
            if (er0 + j * ERP < ROWS) {
                *reinterpret_cast<uint2 *>(plane_h + le_base + j * ERP * G::TW * G::PSB) = hi;
                *reinterpret_cast<uint2 *>(plane_l + le_base + j * ERP * G::TW * G::PSB) = lo;
            }
        }
        __syncthreads();
        SPLIT_TICK(2);
        if (c + 1 < G::NCHUNK) load_chunk(c + 1);
        auto tap_off = [](int i) {
            const int tap = i / G::KS, ks = i % G::KS;
            return ((tap / 3) * G::TW + (tap % 3)) * G::PSB + ks * 32;
        };
        if constexpr (G::COLT) {
            // the leftmost column has nothing to its left, the rightmost nothing to its right
            split_gemm_some<MT, NT, 9 * G::KS>(acc, ah, al, bp + static_cast<long>(c) * (9 * G::KS) * (COUT * 4), COUT, tap_off,
                                               [](int i, int mt) {
                                                   const int kx = (i / G::KS) % 3, col = X0 + mt % WT;     // image column
                                                   return !(kx == 0 && col == 0) && !(kx == 2 && col == W - 1);
                                               });
        } else {
            split_gemm<MT, NT, 9 * G::KS>(acc, ah, al, bp + static_cast<long>(c) * (9 * G::KS) * (COUT * 4), COUT, tap_off);
        }
        SPLIT_TICK(3);
    }

    // ---- BatchNorm affine (bn_s, bn_t arrive multiplied by -log2 e: the accumulators hold z' = -log2(e) z, the
    // exponent of the sigmoid; g_s undoes the factor for the GLU product); z' stays in the accumulators and goes
    // to LDS as hi/lo planes
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float sc = bn_s[ncol + 32 * nt], sh = bn_t[ncol + 32 * nt];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = fmaf(acc[mt][nt][r], sc, sh);
    }

    SPLIT_TICK(4);
    // ---- GLU: lin = z . Wg^T on the matrix cores, out = lin * sigmoid(z), pooled over column pairs.
    // MG m-tiles per wave go through it at a time (a "phase").  z reaches the matrix cores as hi/lo f16
    // planes in LDS (they alias the input planes; two buffers when there is more than one phase).
    // Everything on the vector ALU that does not depend on the phase's own GEMM -- the sigmoid of the
    // phase's z (in place, in the accumulators) and the split of the NEXT phase's z -- is issued between
    // the MFMAs of that GEMM, a few instructions after each.
    // stores go through a buffer descriptor that spans exactly this clip's output: rows past the end of
    // the clip (last tile) fall outside it and the hardware drops them, so there is no branch per store
    float *yc = y + clip * static_cast<long>(H) * (W / 2) * COUT;
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(yc, 0, H * (W / 2) * COUT * 4, 0x00020000);
    // pooled position of (row h0 + pos / W, column pos % W) is h0 * W/2 + pos / 2 for even pos
    const int ybase = ((h0 * (W / 2) + 16 * MT * wm + 2 * half) * COUT + ncol) * 4;
    static_assert(!G::COLT || (G::MG <= 2 && W % 2 == 0), "column tiles go through the GLU one or one column pair at a time");
    const uint4 *gbp = gp + ncol * 2 + half;
    constexpr int MG = G::MG, NPH = G::NPHASE;
    constexpr int NV = MG * NT * 16;          // z values per lane and phase
    constexpr int NS = MG * NT * 8;           // packed splits (two values each) per lane and phase
    constexpr int NSLOT = (COUT / 16) * MG * NT * 3;   // MFMAs of one phase's GEMM
    // value v of phase p lives in acc[p * MG + v / (16 * NT)][(v / 16) % NT][v % 16]
    auto split_z = [&](int p, int sidx) {
        const int mt = sidx / (8 * NT), nt = (sidx / 8) % NT, r = (sidx % 8) * 2;   // registers r, r + 1: neighbouring rows
        unsigned hi, lo;
        split2_f16(acc[p * MG + mt][nt][r], acc[p * MG + mt][nt][r + 1], hi, lo);
        char *zh = smem + (p & 1) * 2 * G::PLANE_Z, *zl = zh + G::PLANE_Z;
        const int off = (32 * MG * wm + 32 * mt + acc_row(r, half)) * G::ZSB + (ncol + 32 * nt) * 2;
        *reinterpret_cast<unsigned short *>(zh + off) = static_cast<unsigned short>(hi);
        *reinterpret_cast<unsigned short *>(zl + off) = static_cast<unsigned short>(lo);
        *reinterpret_cast<unsigned short *>(zh + off + G::ZSB) = static_cast<unsigned short>(hi >> 16);
        *reinterpret_cast<unsigned short *>(zl + off + G::ZSB) = static_cast<unsigned short>(lo >> 16);
    };
    // sigmoid in two halves (each ends in a transcendental, whose latency the other half's covers)
    auto sig_a = [&](int p, int v) {
        f32x16 &a = acc[p * MG + v / (16 * NT)][(v / 16) % NT];
        a[v % 16] = __builtin_amdgcn_exp2f(a[v % 16]);
    };
    auto sig_b = [&](int p, int v) {
        f32x16 &a = acc[p * MG + v / (16 * NT)][(v / 16) % NT];
        a[v % 16] = __builtin_amdgcn_rcpf(1.0f + a[v % 16]);
    };
    __syncthreads();                     // every wave is done reading the input planes
    SPLIT_TICK(5);
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) split_z(0, sidx);
    __syncthreads();
    SPLIT_TICK(6);
#pragma unroll
    for (int p = 0; p < NPH; ++p) {
        f32x16 lin[MG][NT];
#pragma unroll
        for (int mt = 0; mt < MG; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) lin[mt][nt][r] = 0.f;
        const char *zh[MG], *zl[MG];
#pragma unroll
        for (int mt = 0; mt < MG; ++mt) {
            const int off = (p & 1) * 2 * G::PLANE_Z + (32 * MG * wm + 32 * mt + col32) * G::ZSB + 16 * half;
            zh[mt] = smem + off;
            zl[mt] = smem + off + G::PLANE_Z;
        }
        // GLU scale and bias of this lane's columns: requested half way through the GEMM (late enough not to
        // add to its register peak, early enough for the rest of it to cover the latency)
        float gsc[NT], gbias[NT];
        split_gemm<MG, NT, COUT / 16, SPLIT_GLU_BDIST>(lin, zh, zl, gbp, COUT, [](int i) { return i * 32; }, [&](int j) {
            if (j == NSLOT / 2) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    gsc[nt] = g_s[ncol + 32 * nt];
                    gbias[nt] = gb[ncol + 32 * nt];
                }
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (v * NSLOT / NV == j) sig_a(p, v);
                if (v * NSLOT / NV == j - 1) sig_b(p, v);
            }
            if (p + 1 < NPH) {
#pragma unroll
                for (int sidx = 0; sidx < NS; ++sidx)
                    if (sidx * NSLOT / NS == j) split_z(p + 1, sidx);
            }
        });
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if (v * NSLOT / NV == NSLOT - 1) sig_b(p, v);
        SPLIT_TICK(7);
        // the next phase's planes are complete, and this phase's are free to be overwritten after it
        if (p + 1 < NPH) lds_barrier();
        if constexpr (G::COLT) {
            // m-tile mt is the tile's column mt % WT and the lane's register r row acc_row(r, half) of the tile's 32 rows.  With one
            // m-tile per phase an even column parks its gated values in its own (now free) accumulators and the odd
            // column to its right adds its own and stores the pooled pair; with two, the phase is the pair.
            const int last = p * MG + MG - 1;                                    // the m-tile that stores: an odd column
            const int row0 = 32 * ((MT * wm + last) / WT) + 4 * half;            // of register 0
            const int yb = (((h0 + row0) * (W / 2) + (X0 + last % WT) / 2) * COUT + ncol) * 4;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float g = fmaf(lin[MG - 1][nt][r], gsc[nt], gbias[nt]) * acc[last][nt][r];
                    if (MG == 1 && p % 2 == 0) {
                        acc[p][nt][r] = g;
                    } else {
                        const float left = MG == 1 ? acc[last - 1][nt][r]
                                                   : fmaf(lin[0][nt][r], gsc[nt], gbias[nt]) * acc[last - 1][nt][r];
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, left + g), yrsrc,
                                                              yb + (acc_row(r, 0) * (W / 2) * COUT + 32 * nt) * 4, 0, 0);
                    }
                }
        } else
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MG; ++mt)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float o0 = fmaf(lin[mt][nt][r], gsc[nt], gbias[nt]) * acc[p * MG + mt][nt][r];
                    const float o1 = fmaf(lin[mt][nt][r + 1], gsc[nt], gbias[nt]) * acc[p * MG + mt][nt][r + 1];
                    // acc_row(r, half) / 2 = (r & 3) / 2 + 4 * (r >> 2) + 2 * half for even r
                    const int yoff = ((16 * (p * MG + mt) + (r & 3) / 2 + 4 * (r >> 2)) * COUT + 32 * nt) * 4;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o0 + o1), yrsrc,
                                                          ybase + yoff, 0, 0);
                }
        SPLIT_TICK(8);
    }
#ifdef SPLIT_TIMING
    if (tid == 0) {
        constexpr int layer = W == 64 ? 0 : W == 32 ? 1 : W == 16 ? 2 : W == 8 ? 3 : W == 4 ? 4 : 5;
        for (int i = 0; i < 9; ++i) atomicAdd(&g_split_ticks[layer][i], tick_acc[i]);
        atomicAdd(&g_split_ticks[layer][9], 1ull);
        atomicMin(&g_split_ticks[layer][10], tick_first);
        atomicMax(&g_split_ticks[layer][11], tick_last);
    }
#endif
}


// C[M][N] = A[M][K] . B + bias with the same split-f16 arithmetic (GRU input projections).
// A row-major f32; B packed [K/16][2][N][16] f16 (hi, lo) pre-scaled by 2^S; out = acc * oscale + bias.
// A workgroup owns 128 rows at a time: it stages them ONCE as hi/lo planes for the whole K, then walks
// the N columns 64*WN at a time (2 x WN waves; WN = 4 where the planes leave room for one workgroup per
// CU only, so that it still brings two waves per SIMD) with no barrier in that loop, so the stores of one
// column block drain while the next block's MFMAs run.  The output is 3-6x the input here, so the
// kernel is bounded by the rate at which C can be written.  Workgroups are persistent over row tiles
// (tile, tile + gridDim.x, ...) and request the next tile's rows into registers before they start on
// the current one, so that the only exposed input latency is the first tile's.
template <int N, int K, int WN>
__global__ __launch_bounds__(128 * WN, (K <= 128 ? 2 : 1)) void linear_split_kernel(const float *__restrict__ A,
                                                                             const uint4 *__restrict__ bpk,
                                                                             const float *__restrict__ bias, float oscale,
                                                                             float *__restrict__ C, long M) {
    constexpr int PSB = K * 2 + 16, PLANE = 128 * PSB;
    constexpr int NTHR = 128 * WN, NB = 64 * WN;
    static_assert(N % NB == 0 && K % 64 == 0 && (128 * (K / 4)) % NTHR == 0, "unsupported linear geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *plane_h = smem, *plane_l = smem + PLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, col32 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    const long n_tiles = (M + 127) / 128;
    constexpr int NLD = 128 * (K / 4) / NTHR;              // float4 per thread and tile
    // rows of a tile are addressed through buffer descriptors that end at the matrix' last row: loads past
    // it return zero and stores past it are dropped.  The range check covers the per-lane offset only, not
    // the scalar offset operand, so everything that selects a ROW has to be in the per-lane offset.
    // (the builtin's result is converted as a whole: indexing its integer vector type element by element
    // and bit-casting each element to float compiles to a one-dword load of element 0)
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    f32x4 v[NLD];
    const int ld_off = ((tid / (K / 4)) * K + (tid % (K / 4)) * 4) * 4;
    static_assert(NTHR % (K / 4) == 0, "a thread keeps its column over the tile's loads");
    auto tile_rows_bytes = [&](long m0, int row_bytes) {
        const long rows = M - m0 < 128 ? M - m0 : 128;
        return static_cast<int>(rows) * row_bytes;
    };
    auto load_tile = [&](long tile) {
        const long m0 = tile * 128;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A) + m0 * K, 0,
                                                                            tile_rows_bytes(m0, K * 4), 0x00020000);
#pragma unroll
        for (int j = 0; j < NLD; ++j)
            v[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rs, ld_off + j * (NTHR / (K / 4)) * K * 4, 0, 0));
    };
    const char *ah[2], *al[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int off = (64 * wm + 32 * mt + col32) * PSB + 16 * half;
        ah[mt] = plane_h + off;
        al[mt] = plane_l + off;
    }
    if (blockIdx.x < n_tiles) load_tile(blockIdx.x);
#pragma unroll 1
    for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long m0 = tile * 128;
        if (tile != blockIdx.x) lds_barrier();             // every wave is done with the previous tile's planes
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + NTHR * j, r = i / (K / 4), c = (i - r * (K / 4)) * 4;
            uint2 hi, lo;
            split2_f16(v[j][0], v[j][1], hi.x, lo.x);
            split2_f16(v[j][2], v[j][3], hi.y, lo.y);
            *reinterpret_cast<uint2 *>(plane_h + r * PSB + c * 2) = hi;
            *reinterpret_cast<uint2 *>(plane_l + r * PSB + c * 2) = lo;
        }
        lds_barrier();
        if (tile + gridDim.x < n_tiles) load_tile(tile + gridDim.x);
        const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(C + m0 * N, 0, tile_rows_bytes(m0, N * 4), 0x00020000);
        const int st_off = ((64 * wm + 4 * half) * N + wn * 64 + col32) * 4;
#pragma unroll 1
        for (int nb = 0; nb < N / NB; ++nb) {
            const int ncol = nb * NB + wn * 64 + col32;
            // the column block is folded into the lane offset: with a loop-invariant base the compiler keeps
            // all 64 store offsets of a lane in registers across the loop, which spills
            int so = st_off + nb * NB * 4;
            asm volatile("" : "+v"(so));
            f32x16 acc[2][2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
            split_gemm<2, 2, K / 16>(acc, ah, al, bpk + ncol * 2 + half, N, [](int i) { return i * 32; });
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const float bv = bias[ncol + 32 * nt];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)       // row 64 wm + 32 mt + acc_row(r, half), column ncol + 32 nt
                        __builtin_amdgcn_raw_buffer_store_b32(
                            __builtin_bit_cast(unsigned, fmaf(acc[mt][nt][r], oscale, bv)), crs,
                            so + ((32 * mt + acc_row(r, 0)) * N + 32 * nt) * 4, 0, 0);
            }
        }
    }
}
