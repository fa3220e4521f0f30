// Mel projection [128 filters] x [1025 bins] applied to power spectra: the sparse vector-ALU form mel_kernel uses
// against a block-banded form on the matrix cores, both timed alone (the power spectra sit in LDS as they do
// right after the transform; nothing else runs).
//
//   sparse   lane = a pair of filters (a narrow low one with a wide high one), each filter a contiguous run of
//            bins (2020 non-zeros of 131 200), weights in LDS, eight bins per trip; one wave per frame
//            (the loop of csrc/mel.hip).
//   mfma32   v_mfma_f32_32x32x2_f32 (exact float32 products, as the 1e-4 relative tolerance on a power
//            spectrum that spans ten decades wants): M = 32 frames, N = one block of 32 filters, K = the bins
//            any filter of the block touches (the band); 4 blocks: 86 + 148 + 300 + 604 bins -> 569 MFMAs
//            per 32 frames, spread evenly over the four waves of a workgroup (the probe does not reduce the
//            partial sums a real kernel would have to: best case).  A from an LDS tile of 32 power spectra
//            (32 x 1028 x 4 B = 131.6 KB: one workgroup per CU), B = dense weights from L2.
//   mfma16   the same bands on v_mfma_f32_32x32x16_f16 with hi/lo split operands (3 MFMAs per product), A
//            planes prepared beforehand (the split of 1025 values per frame is NOT timed: best case).
//
//   hipcc -O3 --offload-arch=gfx950 tools/melproj_probe.hip -o tools/melproj_probe && tools/melproj_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NB = 1025, NM = 128, PWS = 1040;      // bins, filters, floats per power spectrum in LDS (zero padded)

// ---- sparse: the projection loop of mel_kernel, one wave per frame
__global__ __launch_bounds__(256, 3) void sparse_kernel(const float *mel_w, const int *mel_meta, int nnz, float *out,
                                                        long frames) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *pw_all = reinterpret_cast<float *>(smem);
    float *melw = pw_all + 4 * PWS;
    int *meta = reinterpret_cast<int *>(melw + ((nnz + 3) & ~3));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *pw = pw_all + wave * PWS;
    for (int i = tid; i < nnz; i += 256) melw[i] = mel_w[i];
    for (int i = tid; i < 5 * 64; i += 256) meta[i] = mel_meta[i];
    unsigned h = (tid + 1) * 2654435761u + blockIdx.x;
    for (int i = lane; i < PWS; i += 64) {
        h = h * 1664525u + 1013904223u;
        pw[i] = i < NB ? (h >> 8) * (1.0f / 16777216.0f) : 0.f;
    }
    __syncthreads();
    const long per = (frames + gridDim.x * 4 - 1) / (gridDim.x * 4);
    float keep_a = 0.f, keep_b = 0.f;
    for (long f = 0; f < per; ++f) {
        const int mp = lane;
        const int start_a = meta[5 * mp + 0], len_a = meta[5 * mp + 1], base_b = meta[5 * mp + 2];
        const int total = meta[5 * mp + 3], ptr = meta[5 * mp + 4];
        float acc_a = 0.f, acc_b = 0.f;
        for (int i = 0; i < total; i += 8) {
            const bool in_a = i < len_a;
            const float4 wa = *reinterpret_cast<const float4 *>(melw + ptr + i);
            const float4 wb = *reinterpret_cast<const float4 *>(melw + ptr + i + 4);
            const float *pp = pw + (in_a ? start_a : base_b) + i;
            const float4 pa4 = *reinterpret_cast<const float4 *>(pp), pb4 = *reinterpret_cast<const float4 *>(pp + 4);
            float s0 = wa.x * pa4.x, s1 = wa.y * pa4.y;
            s0 = fmaf(wa.z, pa4.z, s0); s1 = fmaf(wa.w, pa4.w, s1);
            s0 = fmaf(wb.x, pb4.x, s0); s1 = fmaf(wb.y, pb4.y, s1);
            s0 = fmaf(wb.z, pb4.z, s0); s1 = fmaf(wb.w, pb4.w, s1);
            const float sum8 = s0 + s1;
            acc_a += in_a ? sum8 : 0.f;
            acc_b += in_a ? 0.f : sum8;
        }
        keep_a += acc_a;
        keep_b += acc_b;
        // a new frame's power spectrum would be here now: touch the array so that the loop is not hoisted
        pw[(lane + f) & 1023] = keep_a * 1e-30f + 0.5f;
        __builtin_amdgcn_wave_barrier();
    }
    out[blockIdx.x * 256 + tid] = keep_a + keep_b;
}

// ---- matrix cores: K bins of a band per wave, 32 frames x 32 filters per MFMA tile
struct Band { int k0, k1; };                        // bins [k0, k1) of a block of 32 filters

template <bool F16>
__global__ __launch_bounds__(256, 1) void mfma_kernel(const float *wd, const _Float16 *wsp, const int *wave_k0,
                                                      const int *wave_k1, const int *wave_blk, float *out, long tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = lane & 31, half = lane >> 5;
    // tile of 32 power spectra: f32 [32][1028], or two f16 planes [32][1040] (hi, lo)
    unsigned h = (tid + 1) * 2654435761u + blockIdx.x;
    if (F16) {
        _Float16 *p = reinterpret_cast<_Float16 *>(smem);
        for (int i = tid; i < 2 * 32 * 1040; i += 256) {
            h = h * 1664525u + 1013904223u;
            p[i] = static_cast<_Float16>((h >> 20) * (1.0f / 4096.0f));
        }
    } else {
        float *p = reinterpret_cast<float *>(smem);
        for (int i = tid; i < 32 * 1028; i += 256) {
            h = h * 1664525u + 1013904223u;
            p[i] = (h >> 8) * (1.0f / 16777216.0f);
        }
    }
    __syncthreads();
    const int k0 = wave_k0[wave], k1 = wave_k1[wave], blk = wave_blk[wave];
    const long per = (tiles + gridDim.x - 1) / gridDim.x;
    float sink = 0.f;
    for (long t = 0; t < per; ++t) {
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (F16) {
            const char *ph = smem + (row * 1040 + 8 * half) * 2, *pl = ph + 32 * 1040 * 2;
            // weights [bin / 16][plane][filter 128][16]: this lane's 8 f16 of filter 32 blk + row
            const uint4 *wl = reinterpret_cast<const uint4 *>(wsp) + (32 * blk + row) * 2 + half;
            uint4 bh = wl[(k0 / 16 * 2 + 0) * 256], bl = wl[(k0 / 16 * 2 + 1) * 256];
#pragma unroll 2
            for (int k = k0; k < k1; k += 16) {
                const f16x8 ah = *reinterpret_cast<const f16x8 *>(ph + 2 * k), al = *reinterpret_cast<const f16x8 *>(pl + 2 * k);
                const f16x8 vbh = __builtin_bit_cast(f16x8, bh), vbl = __builtin_bit_cast(f16x8, bl);
                const int kn = k + 16 < k1 ? k + 16 : k0;
                bh = wl[(kn / 16 * 2 + 0) * 256];
                bl = wl[(kn / 16 * 2 + 1) * 256];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, vbh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, vbl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, vbh, acc, 0, 0, 0);
            }
        } else {
            const float *pa = reinterpret_cast<const float *>(smem) + row * 1028 + half;
            const float *wb = wd + static_cast<long>(half) * NM + 32 * blk + row;      // dense [bin][filter]
            float b = wb[static_cast<long>(k0) * NM];
#pragma unroll 4
            for (int k = k0; k < k1; k += 2) {
                const float a = pa[k], bv = b;
                const int kn = k + 2 < k1 ? k + 2 : k0;
                b = wb[static_cast<long>(kn) * NM];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
            }
        }
        for (int r = 0; r < 16; ++r) sink += acc[r];
        // the next tile's spectra would be staged here
        reinterpret_cast<float *>(smem)[(tid + t) & 4095] = sink * 1e-30f + 0.25f;
        __syncthreads();
    }
    out[blockIdx.x * 256 + tid] = sink;
}

double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
double mel_to_hz(double m) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

template <typename T>
T *upload(const std::vector<T> &v) {
    T *d;
    CHECK(hipMalloc(&d, v.size() * sizeof(T)));
    CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

template <typename F>
double time_ms(F launch) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0) best = std::min(best, ms);
    }
    CHECK(hipGetLastError());
    return best;
}

int main() {
    const int sr = 16000;
    // Slaney filterbank (sr 16 kHz, n_fft 2048, 128 filters), dense and as runs
    std::vector<double> mel_f(NM + 2);
    const double mmin = hz_to_mel(0.0), mmax = hz_to_mel(sr / 2.0);
    for (int i = 0; i < NM + 2; ++i) mel_f[i] = mel_to_hz(mmin + (mmax - mmin) * i / (NM + 1));
    std::vector<float> dense(static_cast<size_t>(NM) * NB, 0.f);
    std::vector<int> start(NM), len(NM);
    long nnz_real = 0;
    for (int i = 0; i < NM; ++i) {
        const double fd0 = mel_f[i + 1] - mel_f[i], fd1 = mel_f[i + 2] - mel_f[i + 1], enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        int first = -1, last = -1;
        for (int k = 0; k < NB; ++k) {
            const double f = (sr / 2.0) * k / (NB - 1);
            const float w = static_cast<float>(std::max(0.0, std::min((f - mel_f[i]) / fd0, (mel_f[i + 2] - f) / fd1)) * enorm);
            dense[static_cast<size_t>(i) * NB + k] = w;
            if (w != 0.f) { if (first < 0) first = k; last = k; }
        }
        start[i] = first;
        len[i] = last - first + 1;
        nnz_real += len[i];
    }
    // pairs (l, 127 - l), runs starting at multiples of 4 bins and padded to multiples of 8, as the product packs them
    std::vector<int> meta(5 * 64);
    std::vector<float> wsp;
    for (int l = 0; l < 64; ++l) {
        const int fa = l, fb = NM - 1 - l;
        const int sa = start[fa] & ~3, sb = start[fb] & ~3;
        const int la = (len[fa] + start[fa] - sa + 7) & ~7, lb = (len[fb] + start[fb] - sb + 7) & ~7;
        const int ptr = static_cast<int>(wsp.size());
        for (int k = 0; k < la; ++k) wsp.push_back(sa + k < NB ? dense[static_cast<size_t>(fa) * NB + sa + k] : 0.f);
        for (int k = 0; k < lb; ++k) wsp.push_back(sb + k < NB ? dense[static_cast<size_t>(fb) * NB + sb + k] : 0.f);
        meta[5 * l + 0] = sa; meta[5 * l + 1] = la; meta[5 * l + 2] = sb - la; meta[5 * l + 3] = la + lb; meta[5 * l + 4] = ptr;
    }
    // bands of the four blocks of 32 filters, rounded to multiples of 16 bins
    Band band[4];
    long band_bins = 0;
    for (int b = 0; b < 4; ++b) {
        int lo = NB, hi = 0;
        for (int i = 32 * b; i < 32 * b + 32; ++i) { lo = std::min(lo, start[i]); hi = std::max(hi, start[i] + len[i]); }
        band[b].k0 = lo & ~15;
        band[b].k1 = std::min((hi + 15) & ~15, 1024);
        band_bins += band[b].k1 - band[b].k0;
    }
    // spread the bands' bins evenly over four waves (a wave = a contiguous piece of one band)
    std::vector<int> wk0(4), wk1(4), wblk(4);
    {
        // the widest band is cut in two, the three narrow ones share a wave with... keep it simple: wave w gets
        // band_bins / 4 bins of whichever band its share falls into (timing only)
        const int share = static_cast<int>(band_bins / 4) & ~15;
        int b = 0, pos = band[0].k0;
        for (int w = 0; w < 4; ++w) {
            while (band[b].k1 - pos < 16 && b < 3) { ++b; pos = band[b].k0; }
            wblk[w] = b;
            wk0[w] = pos;
            wk1[w] = std::min(pos + share, band[b].k1);
            if (wk1[w] - wk0[w] < share) wk1[w] = std::min(wk0[w] + share, 1024);      // same MFMA count for every wave
            pos = wk1[w];
        }
    }
    std::vector<float> wd(static_cast<size_t>(1040) * NM, 0.f);
    for (int k = 0; k < NB; ++k)
        for (int i = 0; i < NM; ++i) wd[static_cast<size_t>(k) * NM + i] = dense[static_cast<size_t>(i) * NB + k];
    std::vector<_Float16> wsplit(static_cast<size_t>(65) * 2 * NM * 16);
    for (int ks = 0; ks < 65; ++ks)
        for (int i = 0; i < NM; ++i)
            for (int k = 0; k < 16; ++k) {
                const float v = ks * 16 + k < NB ? dense[static_cast<size_t>(i) * NB + ks * 16 + k] * 1024.f : 0.f;
                const _Float16 hi = static_cast<_Float16>(v);
                wsplit[((static_cast<size_t>(ks) * 2 + 0) * NM + i) * 16 + k] = hi;
                wsplit[((static_cast<size_t>(ks) * 2 + 1) * NM + i) * 16 + k] = static_cast<_Float16>(v - static_cast<float>(hi));
            }

    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *d_w = upload(wsp), *d_wd = upload(wd), *d_out;
    int *d_meta = upload(meta), *d_k0 = upload(wk0), *d_k1 = upload(wk1), *d_blk = upload(wblk);
    _Float16 *d_ws = upload(wsplit);
    CHECK(hipMalloc(&d_out, sizeof(float) * 256 * 3 * cus));
    const long frames = 4096L * 501;
    const int nnz = static_cast<int>(wsp.size());
    const size_t lds_s = sizeof(float) * (4 * PWS + ((nnz + 3) & ~3)) + sizeof(int) * 5 * 64;
    const size_t lds_m32 = sizeof(float) * 32 * 1028, lds_m16 = 2 * 2 * 32 * 1040;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(sparse_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_s)));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(mfma_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_m32)));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(mfma_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_m16)));
    const double ts = time_ms([&] { hipLaunchKernelGGL(sparse_kernel, dim3(3 * cus), dim3(256), lds_s, 0, d_w, d_meta, nnz, d_out, frames); });
    const double t32 = time_ms([&] { hipLaunchKernelGGL(mfma_kernel<false>, dim3(cus), dim3(256), lds_m32, 0, d_wd, d_ws, d_k0, d_k1, d_blk, d_out, frames / 32); });
    const double t16 = time_ms([&] { hipLaunchKernelGGL(mfma_kernel<true>, dim3(cus), dim3(256), lds_m16, 0, d_wd, d_ws, d_k0, d_k1, d_blk, d_out, frames / 32); });
    printf("%d CUs; 4096 clips x 501 frames = %ld power spectra of 1025 bins -> 128 mel values each\n", cus, frames);
    printf("filterbank: %ld non-zeros of %d; bands of the four 32-filter blocks:", nnz_real, NM * NB);
    for (int b = 0; b < 4; ++b) printf(" [%d, %d)", band[b].k0, band[b].k1);
    printf(" = %ld bins; per wave [%d,%d) [%d,%d) [%d,%d) [%d,%d)\n", band_bins, wk0[0], wk1[0], wk0[1], wk1[1], wk0[2], wk1[2], wk0[3], wk1[3]);
    printf("sparse vector-ALU projection (the product's loop, 3 workgroups per CU): %.3f ms\n", ts);
    printf("block-banded v_mfma_f32_32x32x2_f32 (exact float32, 1 workgroup per CU: the 131 KB tile):  %.3f ms\n", t32);
    printf("block-banded split-f16 v_mfma_f32_32x32x16_f16 x3 (hi/lo split of the spectra not timed):   %.3f ms\n", t16);
    return 0;
}
