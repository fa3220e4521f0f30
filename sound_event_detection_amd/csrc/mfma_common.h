// Device and host helpers shared by the MFMA kernels of libsedmi.  Meant to be included INSIDE an
// anonymous namespace of the including translation unit.
#pragma once

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ float sigmoidf_fast(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
__device__ __forceinline__ float tanhf_fast(float x) {
    const float e = __expf(-2.0f * fabsf(x));
    const float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    return copysignf(t, x);
}

// row of accumulator register `reg` inside a 32x32 MFMA tile (column = lane & 31)
__host__ __device__ __forceinline__ constexpr int acc_row(int reg, int half) {
    return (reg & 3) + 8 * (reg >> 2) + 4 * half;
}

// ---- host: weight repacking -------------------------------------------------

// pack value(k, n) into [K/8][N][8]
template <typename F>
std::vector<float> pack_kn8(int K, int N, F value) {
    std::vector<float> out(static_cast<size_t>(K) * N);
    for (int k = 0; k < K; ++k)
        for (int n = 0; n < N; ++n) out[(static_cast<size_t>(k / 8) * N + n) * 8 + (k % 8)] = value(k, n);
    return out;
}


// float32 -> [ksteps][2 planes][N][16] f16 (hi, lo) of value(kstep, k16, n) * 2^scale_exp
template <typename F>
std::vector<uint16_t> pack_split(int ksteps, int N, int scale_exp, F value) {
    std::vector<uint16_t> out(static_cast<size_t>(ksteps) * 2 * N * 16);
    for (int ks = 0; ks < ksteps; ++ks)
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < 16; ++k) {
                const float v = std::ldexp(value(ks, k, n), scale_exp);
                const _Float16 hi = static_cast<_Float16>(v);
                const _Float16 lo = static_cast<_Float16>(v - static_cast<float>(hi));
                uint16_t hb, lb;
                std::memcpy(&hb, &hi, 2);
                std::memcpy(&lb, &lo, 2);
                out[((static_cast<size_t>(ks) * 2 + 0) * N + n) * 16 + k] = hb;
                out[((static_cast<size_t>(ks) * 2 + 1) * N + n) * 16 + k] = lb;
            }
    return out;
}

// exponent S such that max|w| * 2^S lies in [64, 128)
int split_scale_exp(const float *w, size_t n) {
    float mx = 0.f;
    for (size_t i = 0; i < n; ++i) mx = std::max(mx, std::fabs(w[i]));
    if (!(mx > 0.f) || !std::isfinite(mx)) return 0;
    int e;
    std::frexp(mx, &e);
    return 7 - e;
}

