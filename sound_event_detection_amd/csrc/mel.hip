// Mel frontend for gfx950: framing + Hann window + real FFT + power + sparse
// Slaney mel projection (+ optional log / normalisation), fused in one kernel.
//
// Stands in for librosa.feature.melspectrogram as the reference calls it at
// feature_extraction.py:45-46; the arithmetic follows librosa's published
// algorithm (stft with center=True / pad_mode="constant" / periodic Hann padded
// to n_fft, power 2.0, filters.mel with Slaney scale and normalisation).
//
// Kernel shape
//   * one wave (64 lanes) transforms one frame: the n_fft real samples are
//     packed into NC = n_fft/2 complex points and run through a
//     R1 x 4 x 16 Cooley-Tukey split (R1 = NC/64) whose butterflies live in
//     registers; the three stages exchange data through a per-wave LDS
//     scratch (three transposes), then each lane untangles bin pairs
//     (k, NC-k) of the real spectrum and leaves |X|^2 in the same scratch;
//   * the wave then applies the filterbank itself as a sparse contraction
//     (every filter is a short contiguous run of bins: 2020 non-zeros of
//     131200 for 128 mels at n_fft 2048), lane = a PAIR of filters (a narrow
//     low one with a wide high one, so that all lanes carry about the same
//     number of bins), eight bins in flight per lane, weights in LDS with run
//     starts skewed across bank groups;
//   * a wave never waits for another wave: no workgroup barrier in the frame
//     loop, LDS = 8.7 KB of scratch per wave + the weights, which leaves room
//     for three waves per SIMD to cover the LDS round trips;
//   * output is mel-major [clip][mel][frame]; neighbouring waves work on
//     neighbouring frames, so the 4-byte writes of a line merge in L2.  The
//     reference's "reshape" of that array is a free reinterpretation;
//   * waves are persistent over frames so that the per-lane twiddles and
//     window taps are loaded once and stay in VGPRs.
#include "common.h"

#include <cmath>

namespace {

struct MelArgs {
    const void *y;           // float32 samples, or int16 PCM (taken as sample / 32768) in the I16 instances
    long n_samples;
    int T;
    int hop;
    int n_fft;
    const float *win;        // [n_fft] window, zero outside the centred Hann
    const float2 *tw_nc;     // [NC]      exp(-2 pi i j / NC)
    const float2 *tw_post;   // [NC/2+1]  exp(-2 pi i k / (2 NC))
    int zlo, zhi;            // complex indices with a non-zero window tap: [zlo, zhi)
    int n_mels;
    // Filters are applied in PAIRS, one pair per lane: filter widths grow with frequency (6 bins at the bottom,
    // ~70 at the top for 128 mels at n_fft 2048), so lane = filter leaves most lanes idle while the widest
    // filter of the pass finishes; pairing filter l of a chunk of 128 with filter 127 - l gives every lane about
    // the same number of bins.  Per pair (5 ints): first bin of A, bins of A (rounded up to a multiple of 8),
    // first bin of B minus the rounded bins of A, rounded bins of A and B together, offset of the pair's
    // weights (A's run, then B's) in mel_w.
    const int *mel_meta;
    int n_pairs;             // 64 per chunk of 128 filters
    const float *mel_w;      // every pair's weights, each run zero-padded to its rounded length
    int nnz;                 // floats in mel_w
    float *out;
    int flags;
    const float *mean;
    const float *istd;
    int tiles_per_clip;
    long total_tiles;
};

// cos / sin of 2*pi*i/32 from exact literals (so that the unrolled butterflies
// get compile-time twiddles).
__host__ __device__ constexpr float cos32(int i) {
    constexpr float C[9] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f,
                            0.83146961230254523708f, 0.70710678118654752440f,
                            0.55557023301960222474f, 0.38268343236508977173f,
                            0.19509032201612826785f, 0.0f};
    i &= 31;
    if (i > 16) i = 32 - i;
    return i > 8 ? -C[16 - i] : C[i];
}
__host__ __device__ constexpr float sin32(int i) { return cos32(i - 8); }

template <int R>
__host__ __device__ constexpr int bitrev(int x) {
    int r = 0;
    for (int b = 1; b < R; b <<= 1) {
        r = (r << 1) | (x & 1);
        x >>= 1;
    }
    return r;
}

// In-register radix-2 decimation-in-frequency DFT of R points, forward sign.
// On return element p holds X[bitrev<R>(p)].
template <int R>
__device__ __forceinline__ void dif(float (&re)[R], float (&im)[R]) {
    constexpr float kS = 0.70710678118654752440f;
#pragma unroll
    for (int half = R / 2; half >= 1; half >>= 1) {
#pragma unroll
        for (int base = 0; base < R; base += 2 * half) {
#pragma unroll
            for (int j = 0; j < half; ++j) {
                const int a = base + j, b = a + half;
                const float ur = re[a], ui = im[a], vr = re[b], vi = im[b];
                re[a] = ur + vr;
                im[a] = ui + vi;
                const float dr = ur - vr, di = ui - vi;
                const int ang = j * (16 / half);  // twiddle angle in units of 2*pi/32
                if (ang == 0) {
                    re[b] = dr;
                    im[b] = di;
                } else if (ang == 8) {            // * (-i)
                    re[b] = di;
                    im[b] = -dr;
                } else if (ang == 4) {
                    re[b] = (dr + di) * kS;
                    im[b] = (di - dr) * kS;
                } else if (ang == 12) {
                    re[b] = (di - dr) * kS;
                    im[b] = -(dr + di) * kS;
                } else {
                    const float c = cos32(ang), s = sin32(ang);
                    re[b] = dr * c + di * s;
                    im[b] = di * c - dr * s;
                }
            }
        }
    }
}

// Orders this wave's LDS traffic: DS operations of one wave execute in issue
// order, so all that is needed is that the compiler keeps the order too.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One complex value from LDS as a single ds_read_b64.  Two neighbouring 8-byte reads would otherwise be fused into
// ds_read2_b64, which occupies the LDS array for 8 cycles per wave where two ds_read_b64 take 2 each; a volatile
// access is never fused.
__device__ __forceinline__ float2 lds_read_c(const float2 *p) {
    // (the LDS address space is spelled out: address-space inference leaves volatile accesses as flat ones)
    typedef const volatile unsigned long long __attribute__((address_space(3))) *lds_u64;
    const unsigned long long raw = *(lds_u64)(p);
    return __builtin_bit_cast(float2, raw);
}

// four floats stored as one 16-byte access that is only 4-byte aligned (dword alignment is all a
// multi-dword global access needs)
typedef float Float4U __attribute__((ext_vector_type(4), aligned(4)));

template <int R1>
struct MelGeom {
    static constexpr int NC = 64 * R1;
    // row stride (complex) of transpose A chosen so that the 4 lane groups of
    // stage 2a read disjoint bank halves
    static constexpr int STRIDE_A = R1 == 8 ? 72 : (R1 == 16 ? 68 : 66);
    static constexpr int STRIDE_B = 4 * R1 + 1;       // [q][4*R1 (+1 pad)]
    static constexpr int SCR_A = R1 * STRIDE_A;
    static constexpr int SCR_B = 16 * STRIDE_B;
    static constexpr int SCR = (SCR_A > SCR_B ? (SCR_A > NC ? SCR_A : NC) : (SCR_B > NC ? SCR_B : NC));
    static constexpr int NPAIR = R1 / 2 + 1;           // bins lane + 64*j <= NC/2
    // column of element (lane group g, radix-4 output r, i = k1 % IB) in a row of transpose B (4 R1 columns):
    // bits from the bottom: i, the low bits of r that fill up to three bits, g >> 1, g & 1, the rest of r
    static constexpr int IB = R1 / 4, RL = 8 / IB;
    __device__ static constexpr int col_b(int g, int r, int i) {
        return i + IB * (r % RL) + 8 * (g >> 1) + 16 * (g & 1) + 32 * (r / RL);
    }
};

// ROW0 / NROW: the first-stage rows n1 in [ROW0, ROW0 + NROW) are the only ones whose 64 complex
// points can carry a non-zero window tap (win 640 centred in n_fft 2048: rows 5..10 of 16); the
// others are compile-time zeros, are never loaded, and their butterflies fold away.
//
// A wave is self-contained: it transforms a frame, leaves the power spectrum in its own LDS scratch
// and applies the filterbank to it (lane = a pair of filters), so there is no workgroup barrier in the frame
// loop and the LDS footprint is the scratch alone (three 4-wave workgroups per CU at n_fft 2048:
// three waves per SIMD to cover the LDS round trips of the transposes).
template <int R1, int WAVES, int ROW0, int NROW, bool I16>
__global__ __launch_bounds__(WAVES * 64, (NROW < R1 ? 3 : 1)) void mel_kernel(const MelArgs a) {
    using G = MelGeom<R1>;
    constexpr int NC = G::NC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *scr_all = reinterpret_cast<float2 *>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform by construction; tells the compiler so
    float2 *scr = scr_all + wave * G::SCR;
    float *melw = reinterpret_cast<float *>(smem + sizeof(float2) * G::SCR * WAVES);
    int *melmeta = reinterpret_cast<int *>(melw + ((a.nnz + 3) & ~3));       // [n_pairs][5]
    float *pw = reinterpret_cast<float *>(scr);        // power spectrum [NC + 1], aliases the scratch

    // ---- one-time per-workgroup / per-lane constants ------------------------
    for (int i = tid; i < a.nnz; i += WAVES * 64) melw[i] = a.mel_w[i];
    for (int i = tid; i < 5 * a.n_pairs; i += WAVES * 64) melmeta[i] = a.mel_meta[i];

    float w0[NROW], w1[NROW];      // window taps of this lane's samples
    float t1r[R1], t1i[R1];        // stage-1 twiddles  W_NC^(lane * k1), k1 = bitrev(p)
#pragma unroll
    for (int i = 0; i < NROW; ++i) {
        const int n = 64 * (ROW0 + i) + lane;
        w0[i] = a.win[2 * n];
        w1[i] = a.win[2 * n + 1];
    }
#pragma unroll
    for (int n1 = 0; n1 < R1; ++n1) {
        const int k1 = bitrev<R1>(n1);
        const float2 t = a.tw_nc[(lane * k1) & (NC - 1)];
        t1r[n1] = t.x;
        t1i[n1] = t.y;
    }
    const int q = lane & 15, g = lane >> 4;
    float t2r[4], t2i[4];          // stage-2a twiddles W_64^(q*r), r = bitrev<4>(p)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float2 t = a.tw_nc[(q * bitrev<4>(p) * (NC / 64)) & (NC - 1)];
        t2r[p] = t.x;
        t2i[p] = t.y;
    }
    // untangle twiddles W_2NC^k, k = lane + 64 j.  For R1 <= 16 only j = 0 is kept in registers: the
    // factor W_2NC^(64 j) = exp(-2 pi i j / (2 R1)) is a compile-time constant, and the 16 registers
    // this frees are what lets three waves share a SIMD.
    constexpr bool kT3Const = R1 <= 16;
    constexpr int NT3 = kT3Const ? 1 : G::NPAIR;
    float t3r[NT3], t3i[NT3];
#pragma unroll
    for (int j = 0; j < NT3; ++j) {
        const int k = lane + 64 * j;
        const float2 t = a.tw_post[k <= NC / 2 ? k : 0];
        t3r[j] = t.x;
        t3i[j] = t.y;
    }
    __syncthreads();

    // A workgroup takes blocks of FB = 32 consecutive frames of one clip, each wave FB / WAVES consecutive
    // frames of it: the overlapping input samples are re-read from the same CU's caches, and a lane
    // collects its filter's value over four frames before it writes them as one 16-byte piece of the
    // mel-major row (single 4-byte writes cost 4-5x the output bytes in HBM write traffic).
    constexpr int FB = 32, ROUNDS = FB / WAVES;
    static_assert(ROUNDS % 4 == 0, "frames per wave and block must be a multiple of 4");
    // A lane collects its two filters' values over the wave's KEEP consecutive frames and writes them as KEEP / 4
    // adjacent 16-byte pieces issued back to back (32 contiguous bytes of a mel-major row): PMC WRITE_SIZE per output
    // byte is 4.7x for single 4-byte writes, 2.6-3.0x for lone 16-byte pieces.
    constexpr int KEEP = 8;
    static_assert(ROUNDS % KEEP == 0, "a wave's frames of a block are whole groups");
    float keep[2][KEEP];           // [filter of the pair][frame of the group]; later chunks of filters write directly
    const unsigned blocks_per_clip = a.tiles_per_clip, total_blocks = static_cast<unsigned>(a.total_tiles);
    // block -> (clip, first frame of this wave in it); one 32-bit division per block, not per frame
    auto block_pos = [&](unsigned blk, unsigned &clip, int &tb) {
        clip = blk / blocks_per_clip;
        // a wave owns ROUNDS consecutive frames of the block, so that four of its results make one 16-byte store
        tb = static_cast<int>(blk - clip * blocks_per_clip) * FB + wave * ROUNDS;
    };
    // raw samples of one frame for this lane (two per active row); loaded one frame ahead of use, through a
    // buffer descriptor that spans exactly the clip: samples before its start (negative offsets wrap to huge
    // unsigned ones) and past its end read as zero -- the centred frames' padding -- with no branch, and a
    // frame that does not exist gets an empty descriptor
    auto load_raw = [&](bool live, unsigned clip, int t, float (&r0)[NROW], float (&r1)[NROW]) {
        constexpr unsigned SB = I16 ? 2u : 4u;               // bytes per sample
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char *>(static_cast<const char *>(a.y)) + static_cast<long>(live ? clip : 0) * a.n_samples * SB, 0,
            live ? static_cast<int>(static_cast<unsigned>(a.n_samples) * SB) : 0, 0x00020000);
        // byte offset of this lane's first sample, modulo 2^32
        const unsigned s0 = (static_cast<unsigned>(t) * a.hop - a.n_fft / 2 + 2 * (64 * ROW0 + lane)) * SB;
#pragma unroll
        for (int i = 0; i < NROW; ++i) {
            // each offset is clamped on its own: with an odd hop a pair can be (sample -1, sample 0), whose offsets
            // are 2^32 - 4 and 0 -- two loads the compiler must not fuse into one wider access
            const unsigned o0 = s0 + i * 128 * SB, o1 = o0 + SB;
            const int c0 = static_cast<int>(o0 < 0xfffffff0u ? o0 : 0xfffffff0u), c1 = static_cast<int>(o1 < 0xfffffff0u ? o1 : 0xfffffff0u);
            if constexpr (I16) {
                // PCM as the WAV file holds it: sample / 32768 is exact in float32, so the result is the one the
                // float32 entry point gives for the converted samples
                r0[i] = static_cast<float>(static_cast<short>(__builtin_amdgcn_raw_buffer_load_b16(rs, c0, 0, 0))) * (1.0f / 32768.0f);
                r1[i] = static_cast<float>(static_cast<short>(__builtin_amdgcn_raw_buffer_load_b16(rs, c1, 0, 0))) * (1.0f / 32768.0f);
            } else {
                r0[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, c0, 0, 0));
                r1[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, c1, 0, 0));
            }
        }
    };
    float raw0[NROW], raw1[NROW];
    // loop state: block blk of clip bclip, this wave's frames tb .. tb + ROUNDS - 1 of it, frame index fr
    unsigned blk = blockIdx.x, bclip = 0;
    int tb = 0, fr = 0;
    if (blk < total_blocks) block_pos(blk, bclip, tb);
    load_raw(blk < total_blocks && tb < a.T, bclip, tb, raw0, raw1);

    while (blk < total_blocks) {
        const long clip = bclip;
        const int t = tb + fr;
        // ---- stage 1: R1-point DFTs over n1, lane = n2 ---------------------
        float re[R1], im[R1];
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) {
            re[n1] = 0.f;
            im[n1] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < NROW; ++i) {
            re[ROW0 + i] = raw0[i] * w0[i];
            im[ROW0 + i] = raw1[i] * w1[i];
        }
        // advance to this wave's next frame (the next one of the block, or the first one of its next block)
        // and request its samples
        if (++fr == ROUNDS) {
            fr = 0;
            blk += gridDim.x;
            if (blk < total_blocks) block_pos(blk, bclip, tb);
        }
        load_raw(blk < total_blocks && tb + fr < a.T, bclip, tb + fr, raw0, raw1);
        if (t >= a.T) continue;                        // wave-uniform: past the end of the clip (last block)
        dif<R1>(re, im);
#pragma unroll
        for (int p = 0; p < R1; ++p) {                 // twiddle, then transpose A: [k1][n2]
            const float xr = re[p] * t1r[p] - im[p] * t1i[p];
            const float xi = re[p] * t1i[p] + im[p] * t1r[p];
            scr[bitrev<R1>(p) * G::STRIDE_A + lane] = make_float2(xr, xi);
        }
        wave_lds_fence();
        // ---- stage 2a: radix-4 over p (n2 = 16 p + q), lane = (q, g) --------
        float br[R1 / 4][4], bi[R1 / 4][4];
#pragma unroll
        for (int i = 0; i < R1 / 4; ++i) {
            const int k1 = g * (R1 / 4) + i;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float2 v = lds_read_c(&scr[k1 * G::STRIDE_A + 16 * p + q]);
                br[i][p] = v.x;
                bi[i][p] = v.y;
            }
        }
        wave_lds_fence();                              // all reads before scratch is reused
#pragma unroll
        for (int i = 0; i < R1 / 4; ++i) {
            dif<4>(br[i], bi[i]);
            // twiddle W_64^(q r), transpose B: row q, column col_b(g, r, i) (k1 = g * (R1 / 4) + i).  The column
            // bits are arranged so that neither side of the transpose meets a bank conflict: bit 4 is the low
            // bit of the lane group g, so the 32 lanes that store together here (16 q x 2 g, 65 q + 16 g complex
            // apart) fall into 32 different bank pairs, and bits 0-4 are exactly the bits of the output position
            // k1 + R1 r (mod 32) that the lanes of stage 2b store to.
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int r = bitrev<4>(p);
                const float xr = br[i][p] * t2r[p] - bi[i][p] * t2i[p];
                const float xi = br[i][p] * t2i[p] + bi[i][p] * t2r[p];
                scr[q * G::STRIDE_B + G::col_b(g, r, i)] = make_float2(xr, xi);
            }
        }
        wave_lds_fence();
        // ---- stage 2b: radix-16 over q, one butterfly per (r, k1) ----------
        constexpr int NB = (4 * R1 + 63) / 64;
        float cr[NB][16], ci[NB][16];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int idx = lane + 64 * u;
            if (idx < 4 * R1) {
#pragma unroll
                for (int qq = 0; qq < 16; ++qq) {
                    const float2 v = lds_read_c(&scr[qq * G::STRIDE_B + idx]);
                    cr[u][qq] = v.x;
                    ci[u][qq] = v.y;
                }
            }
        }
        wave_lds_fence();
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int idx = lane + 64 * u;
            if (idx < 4 * R1) {
                dif<16>(cr[u], ci[u]);
                // inverse of col_b: which (r, k1) column idx of transpose B holds
                const int r = (idx >> 5) * G::RL + (idx / G::IB) % G::RL;
                const int k1 = (2 * ((idx >> 3) & 1) + ((idx >> 4) & 1)) * G::IB + idx % G::IB;
#pragma unroll
                for (int p = 0; p < 16; ++p) {         // natural order Z[k], k = k1 + R1 (r + 4 s)
                    const int s = bitrev<16>(p);
                    scr[k1 + R1 * (r + 4 * s)] = make_float2(cr[u][p], ci[u][p]);
                }
            }
        }
        wave_lds_fence();
        // ---- untangle real spectrum, power --------------------------------
        // X[k] = E + W^k O, X[NC-k] = conj(E - W^k O), E = (Z[k] + conj Z[NC-k]) / 2,
        // O = -i (Z[k] - conj Z[NC-k]) / 2
        float pa[G::NPAIR], pb[G::NPAIR];
        if constexpr (kT3Const) {
            // opaque to the optimiser: otherwise the rotated twiddles are loop invariants, get hoisted out
            // of the frame loop and occupy the very registers this form is meant to free
            asm volatile("" : "+v"(t3r[0]), "+v"(t3i[0]));
        }
#pragma unroll
        for (int j = 0; j < G::NPAIR; ++j) {
            const int k = lane + 64 * j;
            pa[j] = pb[j] = 0.f;
            if (k <= NC / 2) {
                const float2 za = lds_read_c(&scr[k]);
                const float2 zb = lds_read_c(&scr[(NC - k) & (NC - 1)]);
                const float er = 0.5f * (za.x + zb.x), ei = 0.5f * (za.y - zb.y);
                const float oR = 0.5f * (za.y + zb.y), oI = -0.5f * (za.x - zb.x);
                float wr, wi;
                if constexpr (kT3Const) {
                    const int ang = j * (16 / R1);                 // units of 2 pi / 32
                    const float c = cos32(ang), sn = -sin32(ang);  // exp(-i * 2 pi ang / 32)
                    wr = t3r[0] * c - t3i[0] * sn;
                    wi = t3r[0] * sn + t3i[0] * c;
                } else {
                    wr = t3r[j];
                    wi = t3i[j];
                }
                const float tr = oR * wr - oI * wi;
                const float ti = oR * wi + oI * wr;
                const float ar = er + tr, ai = ei + ti;
                const float cr2 = er - tr, ci2 = ei - ti;
                pa[j] = ar * ar + ai * ai;
                pb[j] = cr2 * cr2 + ci2 * ci2;
            }
        }
        wave_lds_fence();                              // every Z read before the scratch becomes the power array
#pragma unroll
        for (int j = 0; j < G::NPAIR; ++j) {
            const int k = lane + 64 * j;
            if (k <= NC / 2) {
                pw[k] = pa[j];
                pw[NC - k] = pb[j];
            }
        }
        if (lane < 8) pw[NC + 1 + lane] = 0.f;
        wave_lds_fence();
        // ---- sparse mel projection: lane = a pair of filters -----------------------
        // weight runs are zero-padded to a multiple of 8 on the host, so eight bins go in flight together
        // (the reads past a filter's end meet zero weights; pw carries 8 floats of zero padding)
        for (int mp = lane; mp < a.n_pairs; mp += 64) {
            const int chunk = mp >> 6;
            const int start_a = melmeta[5 * mp + 0], len_a = melmeta[5 * mp + 1], base_b = melmeta[5 * mp + 2];
            const int total = melmeta[5 * mp + 3], ptr = melmeta[5 * mp + 4];
            float acc_a = 0.f, acc_b = 0.f;
            for (int i = 0; i < total; i += 8) {
                const bool in_a = i < len_a;
                const float4 wa = *reinterpret_cast<const float4 *>(melw + ptr + i);
                const float4 wb = *reinterpret_cast<const float4 *>(melw + ptr + i + 4);
                // run starts are multiples of four bins (the host pads the front of a run with zero weights), so
                // the eight bins come as two 16-byte reads
                const float *pp = pw + (in_a ? start_a : base_b) + i;
                const float4 pa4 = *reinterpret_cast<const float4 *>(pp), pb4 = *reinterpret_cast<const float4 *>(pp + 4);
                const float p0 = pa4.x, p1 = pa4.y, p2 = pa4.z, p3 = pa4.w, p4 = pb4.x, p5 = pb4.y, p6 = pb4.z, p7 = pb4.w;
                float s0 = wa.x * p0, s1 = wa.y * p1;
                s0 = fmaf(wa.z, p2, s0); s1 = fmaf(wa.w, p3, s1);
                s0 = fmaf(wb.x, p4, s0); s1 = fmaf(wb.y, p5, s1);
                s0 = fmaf(wb.z, p6, s0); s1 = fmaf(wb.w, p7, s1);
                const float sum8 = s0 + s1;
                acc_a += in_a ? sum8 : 0.f;
                acc_b += in_a ? 0.f : sum8;
            }
            // chunk of C filters starting at 128 chunk: A = first + lane, B = first + C - 1 - lane
            const int first = 128 * chunk, C = a.n_mels - first < 128 ? a.n_mels - first : 128;
            const int l = mp & 63;
            const int m_a = first + l, m_b = first + C - 1 - l;
            const bool has_a = l < (C + 1) / 2, has_b = l < C / 2;      // an odd chunk's middle filter has no partner
            if (a.flags & SEDMI_MEL_LOG) {
                acc_a = 10.0f * log10f(fmaxf(acc_a, 1e-10f));
                acc_b = 10.0f * log10f(fmaxf(acc_b, 1e-10f));
            }
            if (a.flags & SEDMI_MEL_NORMALISE) {
                if (has_a) acc_a = (acc_a - a.mean[m_a]) * a.istd[m_a];
                if (has_b) acc_b = (acc_b - a.mean[m_b]) * a.istd[m_b];
            }
            if (chunk == 0) {
                const int j4 = t & (KEEP - 1);
#pragma unroll
                for (int jj = 0; jj < KEEP; ++jj)
                    if (jj == j4) {
                        keep[0][jj] = acc_a;
                        keep[1][jj] = acc_b;
                    }
            } else {
                if (has_a) a.out[(clip * a.n_mels + m_a) * static_cast<long>(a.T) + t] = acc_a;
                if (has_b) a.out[(clip * a.n_mels + m_b) * static_cast<long>(a.T) + t] = acc_b;
            }
        }
        if ((t & (KEEP - 1)) == KEEP - 1 || t + 1 >= a.T) {   // last frame of its group (or of the clip)
            const int t0 = t & ~(KEEP - 1), nl = t - t0 + 1;
            const int C = a.n_mels < 128 ? a.n_mels : 128;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int m = pass == 0 ? lane : C - 1 - lane;
                const bool has = pass == 0 ? lane < (C + 1) / 2 : lane < C / 2;
                if (has) {
                    float *dst = a.out + (clip * a.n_mels + m) * static_cast<long>(a.T) + t0;
#pragma unroll
                    for (int q4 = 0; q4 < KEEP / 4; ++q4) {
                        if (nl >= 4 * q4 + 4) {        // rows are T floats apart: 4-byte aligned only
                            *reinterpret_cast<Float4U *>(dst + 4 * q4) =
                                Float4U{keep[pass][4 * q4], keep[pass][4 * q4 + 1], keep[pass][4 * q4 + 2], keep[pass][4 * q4 + 3]};
                        } else {
#pragma unroll
                            for (int jj = 0; jj < 3; ++jj)
                                if (4 * q4 + jj < nl) dst[4 * q4 + jj] = keep[pass][4 * q4 + jj];
                        }
                    }
                }
            }
        }
        wave_lds_fence();                              // scratch is reused by the next frame
    }
}

template <int R1, int WAVES>
size_t mel_lds_bytes(int n_mels, int nnz) {
    using G = MelGeom<R1>;
    const int n_pairs = 64 * ((n_mels + 127) / 128);
    return sizeof(float2) * G::SCR * WAVES + sizeof(float) * ((nnz + 3) & ~3) + sizeof(int) * 5 * n_pairs;
}

// float64 -> per-bin statistics for the optional normalisation stage.  One workgroup per mel bin walks every
// clip's row of that bin in a fixed order (thread (x, y): frames x, x + 64, ... of clips y, y + 16, ...) and
// reduces through a fixed tree, so the result does not depend on scheduling: the same array gives the same
// sums bit for bit on every run and every rank.
__global__ __launch_bounds__(1024) void mel_stats_kernel(const float *mel, long n_clips, int n_mels, long T,
                                                         double *sum, double *sumsq) {
    const int m = blockIdx.x;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    // four independent partial sums per thread (frames tx + 64 (4 i + u)), so that four loads are in flight
    // per thread; the partials are added in a fixed order
    double ps[4] = {0.0, 0.0, 0.0, 0.0}, ps2[4] = {0.0, 0.0, 0.0, 0.0};
    for (long c = ty; c < n_clips; c += 16) {
        const float *row = mel + (c * n_mels + m) * T;
        for (long t0 = tx; t0 < T; t0 += 256) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = t0 + 64 * u < T ? row[t0 + 64 * u] : 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double d = v[u];
                ps[u] += d;
                ps2[u] += d * d;
            }
        }
    }
    const double s = (ps[0] + ps[1]) + (ps[2] + ps[3]), s2 = (ps2[0] + ps2[1]) + (ps2[2] + ps2[3]);
    __shared__ double sh[2][1024];
    sh[0][threadIdx.x] = s;
    sh[1][threadIdx.x] = s2;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {           // the only writer of bin m in this launch
        sum[m] += sh[0][0];
        sumsq[m] += sh[1][0];
    }
}

// The optional stages on an array that already holds the power mel spectrogram, in place, with the very
// expressions mel_kernel uses when the flags ride along with the transform.
__global__ __launch_bounds__(256) void mel_apply_kernel(float *mel, long n_rows, int n_mels, long T, int flags,
                                                        const float *mean, const float *istd) {
    for (long r = blockIdx.x; r < n_rows; r += gridDim.x) {
        const int m = static_cast<int>(r % n_mels);
        const float mu = (flags & SEDMI_MEL_NORMALISE) ? mean[m] : 0.f;
        const float is = (flags & SEDMI_MEL_NORMALISE) ? istd[m] : 1.f;
        float *row = mel + r * T;
        for (long t = threadIdx.x; t < T; t += blockDim.x) {
            float v = row[t];
            if (flags & SEDMI_MEL_LOG) v = 10.0f * log10f(fmaxf(v, 1e-10f));
            if (flags & SEDMI_MEL_NORMALISE) v = (v - mu) * is;
            row[t] = v;
        }
    }
}

// ---- host side: librosa's filterbank / window in double precision -----------

double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = std::log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
double mel_to_hz(double m) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = std::log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

}  // namespace

struct sedmi_mel_plan {
    int sr, n_fft, win_length, hop, n_mels, n_bins;
    int zlo, zhi, nnz, nnz_real;      // nnz: stored weights incl. the zero padding of each run
    int device, num_cus;
    std::vector<float> fb_dense;   // host copy [n_mels, n_bins]
    float *d_win = nullptr;
    float2 *d_tw_nc = nullptr, *d_tw_post = nullptr;
    int *d_meta = nullptr;         // [n_pairs][5], see MelArgs
    int n_pairs = 0;
    float *d_w = nullptr;
};

extern "C" {

int sedmi_mel_plan_create(sedmi_mel_plan **out, int sr, int n_fft, int win_length, int hop_length,
                          int n_mels) {
    SEDMI_REQUIRE(out != nullptr, "plan pointer is null");
    *out = nullptr;
    SEDMI_REQUIRE(n_fft == 1024 || n_fft == 2048 || n_fft == 4096,
                  "n_fft must be 1024, 2048 or 4096 (got %d)", n_fft);
    SEDMI_REQUIRE(win_length > 0 && win_length <= n_fft, "win_length %d not in (0, n_fft]", win_length);
    SEDMI_REQUIRE(hop_length > 0, "hop_length must be positive");
    SEDMI_REQUIRE(n_mels > 0 && n_mels <= 1024, "n_mels %d not in (0, 1024]", n_mels);
    SEDMI_REQUIRE(sr > 0, "sr must be positive");

    // the device is queried before the plan exists, so that a failing query leaks nothing
    int device = 0;
    SEDMI_CHECK_HIP(hipGetDevice(&device));
    hipDeviceProp_t prop;
    SEDMI_CHECK_HIP(hipGetDeviceProperties(&prop, device));
    auto *p = new sedmi_mel_plan();
    p->sr = sr; p->n_fft = n_fft; p->win_length = win_length; p->hop = hop_length;
    p->n_mels = n_mels; p->n_bins = 1 + n_fft / 2; p->nnz_real = 0;
    p->device = device;
    p->num_cus = prop.multiProcessorCount;

    // periodic Hann, centred (librosa.util.pad_center)
    std::vector<float> win(n_fft, 0.f);
    const int lpad = (n_fft - win_length) / 2;
    for (int i = 0; i < win_length; ++i)
        win[lpad + i] = static_cast<float>(0.5 - 0.5 * std::cos(2.0 * M_PI * i / win_length));
    int lo = n_fft, hi = 0;
    for (int i = 0; i < n_fft; ++i)
        if (win[i] != 0.f) { lo = std::min(lo, i); hi = std::max(hi, i + 1); }
    if (hi <= lo) { lo = 0; hi = 0; }
    p->zlo = lo / 2;
    p->zhi = (hi + 1) / 2;

    const int NC = n_fft / 2;
    std::vector<float2> tw_nc(NC), tw_post(NC / 2 + 1);
    for (int j = 0; j < NC; ++j) {
        const double ang = -2.0 * M_PI * j / NC;
        tw_nc[j] = make_float2(static_cast<float>(std::cos(ang)), static_cast<float>(std::sin(ang)));
    }
    for (int k = 0; k <= NC / 2; ++k) {
        const double ang = -2.0 * M_PI * k / (2.0 * NC);
        tw_post[k] = make_float2(static_cast<float>(std::cos(ang)), static_cast<float>(std::sin(ang)));
    }

    // librosa.filters.mel(sr, n_fft, n_mels): Slaney scale, Slaney norm, float32 result
    const int nb = p->n_bins;
    std::vector<double> mel_f(n_mels + 2);
    const double mmin = hz_to_mel(0.0), mmax = hz_to_mel(sr / 2.0);
    const double mstep = (mmax - mmin) / (n_mels + 1);       // numpy.linspace: i * step + start
    for (int i = 0; i < n_mels + 2; ++i)
        mel_f[i] = mel_to_hz(i == n_mels + 1 ? mmax : i * mstep + mmin);
    p->fb_dense.assign(static_cast<size_t>(n_mels) * nb, 0.f);
    std::vector<int> start(n_mels), len(n_mels);
    for (int i = 0; i < n_mels; ++i) {
        const double fd0 = mel_f[i + 1] - mel_f[i], fd1 = mel_f[i + 2] - mel_f[i + 1];
        const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        int first = -1, last = -1;
        for (int k = 0; k < nb; ++k) {
            const double f = (sr / 2.0) * k / (nb - 1);
            const double lower = -(mel_f[i] - f) / fd0, upper = (mel_f[i + 2] - f) / fd1;
            const double w = std::max(0.0, std::min(lower, upper));
            // float32 triangle, then scaled by the float64 norm and rounded again (as numpy's
            // in-place multiply of a float32 array by a float64 vector does)
            const float w32 = static_cast<float>(static_cast<double>(static_cast<float>(w)) * enorm);
            p->fb_dense[static_cast<size_t>(i) * nb + k] = w32;
            if (w32 != 0.f) { if (first < 0) first = k; last = k; }
        }
        start[i] = first < 0 ? 0 : first;
        len[i] = first < 0 ? 0 : last - first + 1;
        p->nnz_real += len[i];
    }
    // one run of weights per PAIR of filters (lane l of a chunk of C <= 128 filters: filters l and C - 1 - l)
    p->n_pairs = 64 * ((n_mels + 127) / 128);
    std::vector<int> meta(static_cast<size_t>(5) * p->n_pairs, 0);
    std::vector<float> wsp;
    unsigned used_groups[4] = {0, 0, 0, 0};
    for (int mp = 0; mp < p->n_pairs; ++mp) {
        const int first = 128 * (mp / 64), C = std::min(128, n_mels - first), l = mp % 64;
        const int fa = l < (C + 1) / 2 ? first + l : -1;
        const int fb = l < C / 2 ? first + C - 1 - l : -1;
        // a run starts at the multiple of four bins at or below the filter's first bin (16-byte reads of the
        // power spectrum) and is zero-padded at both ends to a multiple of eight bins
        const int sa = fa >= 0 ? start[fa] & ~3 : 0, sb = fb >= 0 ? start[fb] & ~3 : 0;
        const int la = fa >= 0 ? (len[fa] + start[fa] - sa + 7) & ~7 : 0, lb = fb >= 0 ? (len[fb] + start[fb] - sb + 7) & ~7 : 0;
        // A 16-byte LDS read is served in four groups of 16 lanes (lanes {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31}
        // of each half wave), and inside a group the 16 reads should fall into 16 different quads of banks: start
        // each run at the first multiple of four floats whose bank quad no other lane of its group uses yet (all
        // lanes advance by the same eight floats per trip, so the quads stay distinct through the loop)
        const int lane = mp % 64, l32 = lane % 32;
        const int group = 2 * (lane / 32) + ((l32 < 4 || (l32 >= 12 && l32 < 16) || (l32 >= 20 && l32 < 28)) ? 0 : 1);
        if (lane == 0) used_groups[0] = used_groups[1] = used_groups[2] = used_groups[3] = 0;
        while (wsp.size() % 4) wsp.push_back(0.f);
        while (used_groups[group] >> ((wsp.size() / 4) % 16) & 1u) wsp.insert(wsp.end(), 4, 0.f);
        used_groups[group] |= 1u << ((wsp.size() / 4) % 16);
        const int ptr = static_cast<int>(wsp.size());
        // (bins outside [start, start + len) of the filter's dense row are zero)
        for (int k = 0; k < la; ++k)
            wsp.push_back(sa + k < nb ? p->fb_dense[static_cast<size_t>(fa) * nb + sa + k] : 0.f);
        for (int k = 0; k < lb; ++k)
            wsp.push_back(sb + k < nb ? p->fb_dense[static_cast<size_t>(fb) * nb + sb + k] : 0.f);
        meta[5 * mp + 0] = sa;
        meta[5 * mp + 1] = la;
        meta[5 * mp + 2] = sb - la;
        meta[5 * mp + 3] = la + lb;
        meta[5 * mp + 4] = ptr;
    }
    p->nnz = static_cast<int>(wsp.size());

    int rc;
    if ((rc = sedmi::upload(&p->d_win, win)) || (rc = sedmi::upload(&p->d_tw_nc, tw_nc)) ||
        (rc = sedmi::upload(&p->d_tw_post, tw_post)) || (rc = sedmi::upload(&p->d_meta, meta)) ||
        (rc = sedmi::upload(&p->d_w, wsp))) {
        sedmi_mel_plan_destroy(p);
        return rc;
    }
    *out = p;
    return SEDMI_OK;
}

int sedmi_mel_plan_destroy(sedmi_mel_plan *p) {
    if (!p) return SEDMI_OK;
    (void)hipFree(p->d_win); (void)hipFree(p->d_tw_nc); (void)hipFree(p->d_tw_post);
    (void)hipFree(p->d_meta); (void)hipFree(p->d_w);
    delete p;
    return SEDMI_OK;
}

int64_t sedmi_mel_num_frames(const sedmi_mel_plan *p, int64_t n_samples) {
    if (!p || n_samples < 0) return sedmi::set_error(SEDMI_EINVAL, "bad plan or n_samples");
    return 1 + n_samples / p->hop;
}

int sedmi_mel_filterbank(const sedmi_mel_plan *p, float *host_out) {
    SEDMI_REQUIRE(p && host_out, "null argument");
    std::memcpy(host_out, p->fb_dense.data(), p->fb_dense.size() * sizeof(float));
    return SEDMI_OK;
}

}  // extern "C"

template <int R1, int WAVES, bool I16, int ROW0 = 0, int NROW = R1>
static int launch_mel(const sedmi_mel_plan *p, MelArgs &a, long n_clips, hipStream_t stream) {
    a.tiles_per_clip = static_cast<int>(sedmi::ceil_div(a.T, 32));     // work items: blocks of 32 frames
    a.total_tiles = n_clips * a.tiles_per_clip;
    SEDMI_REQUIRE(a.total_tiles < (1LL << 31) - (1LL << 20), "too many frames in one call: %ld blocks of 32 frames",
                  a.total_tiles);
    const size_t lds = mel_lds_bytes<R1, WAVES>(p->n_mels, p->nnz);
    SEDMI_REQUIRE(lds <= 160 * 1024, "mel kernel needs %zu B of LDS (> 160 KiB): n_mels too large", lds);
    auto kern = mel_kernel<R1, WAVES, ROW0, NROW, I16>;
    if (int rc_lds = sedmi::set_max_dynamic_lds(reinterpret_cast<const void *>(kern), lds)) return rc_lds;
    const long per_cu = std::max<long>(1, std::min<long>(8, (160 * 1024) / static_cast<long>(lds)));
    const long grid = std::min<long>(a.total_tiles, per_cu * p->num_cus);
    const double frames = static_cast<double>(n_clips) * a.T;
    sedmi::ProfileScope prof("mel_kernel", stream,
                             frames * (2.5 * p->n_fft * std::log2(static_cast<double>(p->n_fft)) + 3.0 * p->n_bins +
                                       2.0 * p->nnz_real),
                             (I16 ? 2.0 : 4.0) * static_cast<double>(n_clips) * a.n_samples + 4.0 * frames * p->n_mels);
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(grid)), dim3(WAVES * 64), lds, stream, a);
    SEDMI_CHECK_LAUNCH();
    return SEDMI_OK;
}

template <bool I16>
static int melspec(const sedmi_mel_plan *p, const void *y, int64_t n_clips, int64_t n_samples, int flags,
                   const float *mean, const float *istd, float *out, void *stream) {
    SEDMI_REQUIRE(p != nullptr, "plan is null");
    SEDMI_REQUIRE(n_clips >= 0 && n_samples >= 0, "negative size");
    SEDMI_REQUIRE((flags & ~(SEDMI_MEL_LOG | SEDMI_MEL_NORMALISE)) == 0, "unknown flags 0x%x", flags);
    SEDMI_REQUIRE(!(flags & SEDMI_MEL_NORMALISE) || (mean && istd), "normalisation needs mean and istd");
    if (n_clips == 0) return SEDMI_OK;
    SEDMI_REQUIRE(y && out, "null data pointer");
    const int64_t T = 1 + n_samples / p->hop;
    // samples are addressed with 32-bit byte offsets inside a clip
    SEDMI_REQUIRE(n_samples <= (1LL << 30) - (1LL << 14), "clip too long: at most 2^30 - 2^14 samples");
    MelArgs a{};
    a.y = y; a.n_samples = n_samples; a.T = static_cast<int>(T); a.hop = p->hop; a.n_fft = p->n_fft;
    a.win = p->d_win; a.tw_nc = p->d_tw_nc; a.tw_post = p->d_tw_post; a.zlo = p->zlo; a.zhi = p->zhi;
    a.n_mels = p->n_mels; a.mel_meta = p->d_meta; a.n_pairs = p->n_pairs;
    a.mel_w = p->d_w; a.nnz = p->nnz; a.out = out; a.flags = flags; a.mean = mean; a.istd = istd;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (p->n_fft) {
        case 1024: return launch_mel<8, 4, I16>(p, a, n_clips, s);
        case 2048:
            // window confined to first-stage rows 5..10 (e.g. win 640): specialised instance
            if (p->zlo >= 5 * 64 && p->zhi <= 11 * 64) return launch_mel<16, 4, I16, 5, 6>(p, a, n_clips, s);
            return launch_mel<16, 4, I16>(p, a, n_clips, s);
        default:   return launch_mel<32, 4, I16>(p, a, n_clips, s);
    }
}

extern "C" {

int sedmi_melspec_f32(const sedmi_mel_plan *p, const float *y, int64_t n_clips, int64_t n_samples,
                      int flags, const float *mean, const float *istd, float *out, void *stream) {
    return melspec<false>(p, y, n_clips, n_samples, flags, mean, istd, out, stream);
}

int sedmi_melspec_i16(const sedmi_mel_plan *p, const int16_t *y, int64_t n_clips, int64_t n_samples,
                      int flags, const float *mean, const float *istd, float *out, void *stream) {
    return melspec<true>(p, y, n_clips, n_samples, flags, mean, istd, out, stream);
}

int sedmi_mel_stats_f64(const float *mel, int64_t n_clips, int n_mels, int64_t T, double *sum,
                        double *sumsq, void *stream) {
    SEDMI_REQUIRE(n_clips >= 0 && n_mels > 0 && T >= 0, "bad shape");
    if (n_clips == 0 || T == 0) return SEDMI_OK;
    SEDMI_REQUIRE(mel && sum && sumsq, "null data pointer");
    sedmi::ProfileScope prof("mel_stats_kernel", static_cast<hipStream_t>(stream), 3.0 * n_clips * n_mels * T,
                             4.0 * n_clips * n_mels * T);
    hipLaunchKernelGGL(mel_stats_kernel, dim3(n_mels), dim3(1024), 0,
                       static_cast<hipStream_t>(stream), mel, n_clips, n_mels, T, sum, sumsq);
    SEDMI_CHECK_LAUNCH();
    return SEDMI_OK;
}

int sedmi_mel_apply_f32(float *mel, int64_t n_clips, int n_mels, int64_t T, int flags, const float *mean,
                        const float *istd, void *stream) {
    SEDMI_REQUIRE(n_clips >= 0 && n_mels > 0 && T >= 0, "bad shape");
    SEDMI_REQUIRE((flags & ~(SEDMI_MEL_LOG | SEDMI_MEL_NORMALISE)) == 0, "unknown flags 0x%x", flags);
    SEDMI_REQUIRE(!(flags & SEDMI_MEL_NORMALISE) || (mean && istd), "normalisation needs mean and istd");
    if (n_clips == 0 || T == 0 || flags == 0) return SEDMI_OK;
    SEDMI_REQUIRE(mel != nullptr, "null data pointer");
    const long n_rows = static_cast<long>(n_clips) * n_mels;
    const unsigned grid = static_cast<unsigned>(std::min<long>(n_rows, 256L * 32));
    sedmi::ProfileScope prof("mel_apply_kernel", static_cast<hipStream_t>(stream), 2.0 * n_rows * T, 8.0 * n_rows * T);
    hipLaunchKernelGGL(mel_apply_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), mel, n_rows,
                       n_mels, T, flags, mean, istd);
    SEDMI_CHECK_LAUNCH();
    return SEDMI_OK;
}

}  // extern "C"
