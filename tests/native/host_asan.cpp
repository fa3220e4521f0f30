// Host-side memory-safety check of libsedmi's C ABI.
//
// Built together with the library's sources with AddressSanitizer on the HOST code only
// (tools/build_host_asan.sh: hipcc ... -Xarch_host -fsanitize=address; device code is not instrumented) and run on
// a GPU: every create / pack / launch / destroy path of include/sedmi.h is driven once with exact-size heap
// buffers, so that a host-side read or write past a weight tensor, a table or a workspace aborts the run.
// Results are not checked here (the parity tests do that); error paths must return negative codes, not crash.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "../../include/sedmi.h"

static unsigned g_seed = 12345;
static float rnd() {
    g_seed = g_seed * 1664525u + 1013904223u;
    return (static_cast<int>(g_seed >> 8 & 0xffff) - 32768) * (1.0f / 65536.0f);
}
// exact-size heap array of n random floats in [-scale/2, scale/2) (+ offset)
static std::unique_ptr<float[]> arr(size_t n, float scale = 0.2f, float offset = 0.f) {
    std::unique_ptr<float[]> p(new float[n]);
    for (size_t i = 0; i < n; ++i) p[i] = rnd() * scale + offset;
    return p;
}
#define REQUIRE(cond)                                                                          \
    do {                                                                                       \
        if (!(cond)) {                                                                         \
            std::fprintf(stderr, "FAILED %s:%d: %s (last error: %s)\n", __FILE__, __LINE__, #cond, sedmi_last_error()); \
            std::exit(1);                                                                      \
        }                                                                                      \
    } while (0)
#define HIP_OK(x) REQUIRE((x) == hipSuccess)

template <typename T>
static T *dev(size_t n, const T *host = nullptr) {
    T *d = nullptr;
    HIP_OK(hipMalloc(reinterpret_cast<void **>(&d), (n ? n : 1) * sizeof(T)));
    if (host) HIP_OK(hipMemcpy(d, host, n * sizeof(T), hipMemcpyHostToDevice));
    else HIP_OK(hipMemset(d, 0, (n ? n : 1) * sizeof(T)));
    return d;
}

static void check_mel() {
    const int cfgs[][5] = {{16000, 2048, 640, 320, 128}, {44100, 2048, 1764, 882, 128}, {16000, 1024, 1024, 256, 64},
                           {16000, 4096, 2400, 1200, 128}, {16000, 2048, 640, 160, 768}, {16000, 2048, 2048, 512, 1},
                           {8000, 1024, 400, 100, 1024}, {16000, 2048, 1, 7, 13}};
    for (const auto &c : cfgs) {
        sedmi_mel_plan *plan = nullptr;
        REQUIRE(sedmi_mel_plan_create(&plan, c[0], c[1], c[2], c[3], c[4]) == SEDMI_OK);
        std::unique_ptr<float[]> fb(new float[static_cast<size_t>(c[4]) * (1 + c[1] / 2)]);
        REQUIRE(sedmi_mel_filterbank(plan, fb.get()) == SEDMI_OK);
        const int64_t n_clips = 3, n_samples = 5000 + c[3];
        const int64_t T = sedmi_mel_num_frames(plan, n_samples);
        REQUIRE(T == 1 + n_samples / c[3]);
        auto y = arr(n_clips * n_samples);
        float *dy = dev<float>(n_clips * n_samples, y.get());
        float *dout = dev<float>(n_clips * c[4] * T);
        auto mean = arr(c[4]), istd = arr(c[4], 0.1f, 1.0f);
        float *dmean = dev<float>(c[4], mean.get()), *distd = dev<float>(c[4], istd.get());
        double *dsum = dev<double>(c[4]), *dsq = dev<double>(c[4]);
        REQUIRE(sedmi_melspec_f32(plan, dy, n_clips, n_samples, 0, nullptr, nullptr, dout, nullptr) == SEDMI_OK);
        REQUIRE(sedmi_mel_stats_f64(dout, n_clips, c[4], T, dsum, dsq, nullptr) == SEDMI_OK);
        REQUIRE(sedmi_mel_apply_f32(dout, n_clips, c[4], T, SEDMI_MEL_LOG | SEDMI_MEL_NORMALISE, dmean, distd, nullptr) == SEDMI_OK);
        REQUIRE(sedmi_melspec_f32(plan, dy, n_clips, n_samples, SEDMI_MEL_LOG | SEDMI_MEL_NORMALISE, dmean, distd, dout, nullptr) == SEDMI_OK);
        std::vector<int16_t> pcm(n_clips * n_samples, 1000);
        int16_t *dpcm = dev<int16_t>(pcm.size(), pcm.data());
        REQUIRE(sedmi_melspec_i16(plan, dpcm, n_clips, n_samples, 0, nullptr, nullptr, dout, nullptr) == SEDMI_OK);
        HIP_OK(hipDeviceSynchronize());
        // error paths
        REQUIRE(sedmi_melspec_f32(plan, dy, n_clips, n_samples, SEDMI_MEL_NORMALISE, nullptr, nullptr, dout, nullptr) < 0);
        REQUIRE(sedmi_melspec_f32(plan, dy, n_clips, n_samples, 64, nullptr, nullptr, dout, nullptr) < 0);
        REQUIRE(sedmi_melspec_f32(nullptr, dy, n_clips, n_samples, 0, nullptr, nullptr, dout, nullptr) < 0);
        REQUIRE(sedmi_melspec_f32(plan, dy, 0, n_samples, 0, nullptr, nullptr, dout, nullptr) == SEDMI_OK);
        for (void *p : {static_cast<void *>(dy), static_cast<void *>(dout), static_cast<void *>(dmean), static_cast<void *>(distd),
                        static_cast<void *>(dsum), static_cast<void *>(dsq), static_cast<void *>(dpcm)})
            HIP_OK(hipFree(p));
        REQUIRE(sedmi_mel_plan_destroy(plan) == SEDMI_OK);
    }
    sedmi_mel_plan *bad = nullptr;
    REQUIRE(sedmi_mel_plan_create(&bad, 16000, 3000, 640, 320, 128) < 0 && bad == nullptr);
    REQUIRE(sedmi_mel_plan_create(&bad, 16000, 2048, 4096, 320, 128) < 0);
    REQUIRE(sedmi_mel_plan_create(&bad, 16000, 2048, 640, 320, 5000) < 0);
    REQUIRE(sedmi_mel_plan_destroy(nullptr) == SEDMI_OK);
    std::printf("mel ok\n");
}

static void check_crnn(int n_class) {
    const int filters[7] = {16, 32, 64, 128, 128, 128, 128};
    std::vector<std::unique_ptr<float[]>> keep;
    auto mk = [&](size_t n, float scale = 0.2f, float off = 0.f) {
        keep.push_back(arr(n, scale, off));
        return keep.back().get();
    };
    sedmi_crnn_weights w;
    std::memset(&w, 0, sizeof(w));
    w.n_class = n_class; w.n_freq = 128; w.hidden = 128; w.bn_eps = 1e-3f;
    int cin = 1;
    for (int i = 0; i < 7; ++i) {
        const int co = filters[i];
        w.filters[i] = co; w.pool_t[i] = i == 0 ? 2 : 1; w.pool_f[i] = 2;
        w.conv_w[i] = mk(static_cast<size_t>(co) * cin * 9); w.conv_b[i] = mk(co);
        w.bn_w[i] = mk(co, 0.2f, 1.f); w.bn_b[i] = mk(co); w.bn_mean[i] = mk(co); w.bn_var[i] = mk(co, 0.2f, 1.f);
        w.glu_w[i] = mk(static_cast<size_t>(co) * co); w.glu_b[i] = mk(co);
        cin = co;
    }
    for (int l = 0; l < 2; ++l)
        for (int d = 0; d < 2; ++d) {
            w.gru_w_ih[l][d] = mk(static_cast<size_t>(384) * (l ? 256 : 128)); w.gru_w_hh[l][d] = mk(384 * 128);
            w.gru_b_ih[l][d] = mk(384); w.gru_b_hh[l][d] = mk(384);
        }
    w.dense_w = mk(static_cast<size_t>(n_class) * 256); w.dense_b = mk(n_class);
    sedmi_crnn *m = nullptr;
    REQUIRE(sedmi_crnn_create(&m, &w) == SEDMI_OK);
    for (int precision : {SEDMI_PRECISION_SPLIT_F16, SEDMI_PRECISION_F32}) {
        REQUIRE(sedmi_crnn_set_precision(m, precision) == SEDMI_OK);
        for (const auto &bt : {std::pair<int64_t, int64_t>{3, 64}, {1, 2}, {5, 501}}) {
            const int64_t B = bt.first, T = bt.second, Tp = sedmi_crnn_out_frames(m, T);
            REQUIRE(Tp == T / 2);
            auto x = arr(B * T * 128, 2.f, 1.f);
            float *dx = dev<float>(B * T * 128, x.get()), *dp = dev<float>(B * Tp * n_class);
            const size_t wsb = sedmi_crnn_workspace_bytes(m, B, T);
            char *ws = dev<char>(wsb);
            REQUIRE(sedmi_crnn_forward_f32(m, dx, B, T, dp, ws, wsb, nullptr) == SEDMI_OK);
            float *dseq = dev<float>(B * Tp * 128), *drec = dev<float>(B * Tp * 256);
            REQUIRE(sedmi_crnn_cnn_f32(m, dx, B, T, dseq, ws, wsb, nullptr) == SEDMI_OK);
            REQUIRE(sedmi_crnn_gru_f32(m, dseq, B, Tp, drec, ws, wsb, nullptr) == SEDMI_OK);
            REQUIRE(sedmi_crnn_head_f32(m, drec, B * Tp, dp, nullptr) == SEDMI_OK);
            HIP_OK(hipDeviceSynchronize());
            REQUIRE(sedmi_crnn_forward_f32(m, dx, B, T, dp, ws, wsb / 2, nullptr) < 0);      // workspace too small
            REQUIRE(sedmi_crnn_forward_f32(m, nullptr, B, T, dp, ws, wsb, nullptr) < 0);
            for (void *p : {static_cast<void *>(dx), static_cast<void *>(dp), static_cast<void *>(ws), static_cast<void *>(dseq),
                            static_cast<void *>(drec)})
                HIP_OK(hipFree(p));
        }
    }
    REQUIRE(sedmi_crnn_set_precision(m, 7) < 0);
    REQUIRE(sedmi_crnn_destroy(m) == SEDMI_OK);
    sedmi_crnn_weights bad = w;
    bad.conv_w[3] = nullptr;
    sedmi_crnn *mb = nullptr;
    REQUIRE(sedmi_crnn_create(&mb, &bad) < 0 && mb == nullptr);
    std::printf("crnn (%d classes) ok\n", n_class);
}

static void fill_shake_cnn(sedmi_shake_cnn_weights &c, std::vector<std::unique_ptr<float[]>> &keep) {
    auto mk = [&](size_t n, float scale = 0.2f, float off = 0.f) {
        keep.push_back(arr(n, scale, off));
        return keep.back().get();
    };
    const int cin[3] = {1, 16, 32}, cout[3] = {16, 32, 64}, k[3] = {5, 3, 3};
    for (int b = 0; b < 3; ++b) {
        c.conv_w[b] = mk(static_cast<size_t>(2) * cout[b] * cin[b] * k[b] * k[b]); c.conv_b[b] = mk(2 * cout[b]);
        c.bn1_w[b] = mk(2 * cout[b], 0.2f, 1.f); c.bn1_b[b] = mk(2 * cout[b]); c.bn1_mean[b] = mk(2 * cout[b]);
        c.bn1_var[b] = mk(2 * cout[b], 0.2f, 1.f);
        c.br1_w[b] = mk(static_cast<size_t>(cout[b]) * cout[b] * 9); c.br1_b[b] = mk(cout[b]);
        c.br2_w[b] = mk(static_cast<size_t>(cout[b]) * cout[b] * 9); c.br2_b[b] = mk(cout[b]);
        c.bn2_w[b] = mk(cout[b], 0.2f, 1.f); c.bn2_b[b] = mk(cout[b]); c.bn2_mean[b] = mk(cout[b]); c.bn2_var[b] = mk(cout[b], 0.2f, 1.f);
    }
}

static void check_shake() {
    const float alphas[3] = {0.3f, 0.5f, 0.7f};
    {
        std::vector<std::unique_ptr<float[]>> keep;
        auto mk = [&](size_t n, float scale = 0.02f, float off = 0.f) {
            keep.push_back(arr(n, scale, off));
            return keep.back().get();
        };
        sedmi_shake_rcnn_weights w;
        std::memset(&w, 0, sizeof(w));
        w.n_class = 10;
        fill_shake_cnn(w.cnn, keep);
        for (int d = 0; d < 2; ++d) {
            w.gru_w_ih[d] = mk(static_cast<size_t>(3072) * 6144); w.gru_w_hh[d] = mk(static_cast<size_t>(3072) * 1024);
            w.gru_b_ih[d] = mk(3072); w.gru_b_hh[d] = mk(3072);
        }
        w.ln_w = mk(2048, 0.2f, 1.f); w.ln_b = mk(2048); w.fc_w = mk(10 * 2048); w.fc_b = mk(10);
        sedmi_shake_rcnn *m = nullptr;
        REQUIRE(sedmi_shake_rcnn_create(&m, &w) == SEDMI_OK);
        for (int mode : {SEDMI_RECURRENCE_AUTO, SEDMI_RECURRENCE_PER_STEP}) {
            REQUIRE(sedmi_shake_rcnn_set_recurrence(m, mode) == SEDMI_OK);
            const int64_t B = 2, T = 37;
            auto x = arr(B * T * 768, 2.f);
            float *dx = dev<float>(B * T * 768, x.get()), *dl = dev<float>(B * (T / 4) * 10), *dseq = dev<float>(B * (T / 4) * 6144);
            const size_t wsb = sedmi_shake_rcnn_workspace_bytes(m, B, T);
            char *ws = dev<char>(wsb);
            REQUIRE(sedmi_shake_rcnn_forward_f32(m, dx, B, T, alphas, dl, ws, wsb, nullptr) == SEDMI_OK);
            REQUIRE(sedmi_shake_rcnn_check(m, ws, wsb, B, T, nullptr) == SEDMI_OK);
            REQUIRE(sedmi_shake_cnn_f32(m, dx, B, T, alphas, dseq, ws, wsb, nullptr) == SEDMI_OK);
            HIP_OK(hipDeviceSynchronize());
            REQUIRE(sedmi_shake_rcnn_forward_f32(m, dx, B, T, alphas, dl, ws, 16, nullptr) < 0);
            REQUIRE(sedmi_shake_rcnn_check(m, ws, 16, B, T, nullptr) < 0);
            for (void *p : {static_cast<void *>(dx), static_cast<void *>(dl), static_cast<void *>(dseq), static_cast<void *>(ws)}) HIP_OK(hipFree(p));
        }
        REQUIRE(sedmi_shake_rcnn_set_recurrence(m, 9) < 0);
        REQUIRE(sedmi_shake_rcnn_destroy(m) == SEDMI_OK);
        std::printf("shake rcnn ok\n");
    }
    {
        std::vector<std::unique_ptr<float[]>> keep;
        auto mk = [&](size_t n, float scale = 0.02f, float off = 0.f) {
            keep.push_back(arr(n, scale, off));
            return keep.back().get();
        };
        sedmi_shake_transformer_weights w;
        std::memset(&w, 0, sizeof(w));
        w.n_class = 6; w.n_layers = 2;
        fill_shake_cnn(w.cnn, keep);
        w.fc1_w = mk(static_cast<size_t>(1024) * 6144); w.fc1_b = mk(1024); w.fc2_w = mk(6 * 1024); w.fc2_b = mk(6);
        for (int i = 0; i < 2; ++i) {
            auto &l = w.layers[i];
            l.in_proj_w = mk(static_cast<size_t>(3072) * 1024); l.in_proj_b = mk(3072); l.out_proj_w = mk(1024 * 1024); l.out_proj_b = mk(1024);
            l.lin1_w = mk(2048 * 1024); l.lin1_b = mk(2048); l.lin2_w = mk(1024 * 2048); l.lin2_b = mk(1024);
            l.norm1_w = mk(1024, 0.2f, 1.f); l.norm1_b = mk(1024); l.norm2_w = mk(1024, 0.2f, 1.f); l.norm2_b = mk(1024);
        }
        sedmi_shake_transformer *m = nullptr;
        REQUIRE(sedmi_shake_transformer_create(&m, &w) == SEDMI_OK);
        const int64_t B = 2, T = 41;
        auto x = arr(B * T * 768, 2.f);
        float *dx = dev<float>(B * T * 768, x.get()), *dl = dev<float>(B * (T / 4) * 6);
        const size_t wsb = sedmi_shake_transformer_workspace_bytes(m, B, T);
        char *ws = dev<char>(wsb);
        REQUIRE(sedmi_shake_transformer_forward_f32(m, dx, B, T, alphas, dl, ws, wsb, nullptr) == SEDMI_OK);
        HIP_OK(hipDeviceSynchronize());
        for (void *p : {static_cast<void *>(dx), static_cast<void *>(dl), static_cast<void *>(ws)}) HIP_OK(hipFree(p));
        REQUIRE(sedmi_shake_transformer_destroy(m) == SEDMI_OK);
        w.n_layers = 99;
        REQUIRE(sedmi_shake_transformer_create(&m, &w) < 0);
        std::printf("shake transformer ok\n");
    }
    for (int F : {64, 10, 37}) {
        std::vector<std::unique_ptr<float[]>> keep;
        auto mk = [&](size_t n, float scale = 0.1f) {
            keep.push_back(arr(n, scale));
            return keep.back().get();
        };
        sedmi_basic_rcnn_weights w;
        std::memset(&w, 0, sizeof(w));
        w.n_class = 10; w.frequency_dim = F;
        w.conv_w[0] = mk(128 * 9); w.conv_w[1] = mk(128 * 128 * 9); w.conv_w[2] = mk(128 * 128 * 9);
        for (int i = 0; i < 3; ++i) w.conv_b[i] = mk(128);
        const size_t K = static_cast<size_t>(128) * ((F - 8) / 2);
        w.gru_w_ih = mk(96 * K); w.gru_w_hh = mk(96 * 32); w.gru_b_ih = mk(96); w.gru_b_hh = mk(96);
        w.fc_w = mk(10 * 32); w.fc_b = mk(10);
        sedmi_basic_rcnn *m = nullptr;
        REQUIRE(sedmi_basic_rcnn_create(&m, &w) == SEDMI_OK);
        const int64_t B = 2, T = 21;
        auto x = arr(B * T * F, 2.f);
        float *dx = dev<float>(B * T * F, x.get()), *dl = dev<float>(B * (T / 2) * 10);
        const size_t wsb = sedmi_basic_rcnn_workspace_bytes(m, B, T);
        char *ws = dev<char>(wsb);
        REQUIRE(sedmi_basic_rcnn_forward_f32(m, dx, B, T, dl, ws, wsb, nullptr) == SEDMI_OK);
        HIP_OK(hipDeviceSynchronize());
        for (void *p : {static_cast<void *>(dx), static_cast<void *>(dl), static_cast<void *>(ws)}) HIP_OK(hipFree(p));
        REQUIRE(sedmi_basic_rcnn_destroy(m) == SEDMI_OK);
    }
    std::printf("basic rcnn ok\n");
}

static void check_decode() {
    const int n_cls = 7;
    const int64_t lens[4] = {250, 0, 65, 1000};
    int64_t start[5] = {0}, cut[4] = {251, 3, 10, 2000};
    for (int i = 0; i < 4; ++i) start[i + 1] = start[i] + lens[i];
    const int64_t total = start[4];
    auto p = arr(total * n_cls, 1.0f, 0.5f);
    float *dp = dev<float>(total * n_cls, p.get());
    int64_t *dstart = dev<int64_t>(5, start), *dcut = dev<int64_t>(4, cut), *dn = dev<int64_t>(1);
    const int64_t cap = total * n_cls / 2 + 4 * n_cls + 1;
    int32_t *ev[4];
    for (auto &e : ev) e = dev<int32_t>(cap);
    const size_t wsb = sedmi_decode_workspace_bytes(4, n_cls);
    char *ws = dev<char>(wsb);
    for (int med : {1, 5}) {
        REQUIRE(sedmi_decode_events(dp, total, n_cls, dstart, dcut, 4, 0.5f, 0, med, ev[0], ev[1], ev[2], ev[3], cap, dn, ws, wsb, nullptr) == SEDMI_OK);
        REQUIRE(sedmi_decode_events_static(dp, total, n_cls, dstart, dcut, 4, 0.5f, 1, med, ev[0], ev[1], ev[2], ev[3], cap, dn, nullptr) == SEDMI_OK);
    }
    REQUIRE(sedmi_decode_events(dp, total, n_cls, dstart, dcut, 4, 0.5f, 0, 1, ev[0], ev[1], ev[2], ev[3], cap, dn, ws, 1, nullptr) < 0);
    REQUIRE(sedmi_decode_events_static(dp, total, 1 << 20, dstart, dcut, 4, 0.5f, 0, 1, ev[0], ev[1], ev[2], ev[3], cap, dn, nullptr) < 0);
    int64_t *dth = dev<int64_t>(total * n_cls), *dcounts = dev<int64_t>(n_cls * 4);
    REQUIRE(sedmi_threshold_i64(dp, total * n_cls, 0.5f, 1, dth, nullptr) == SEDMI_OK);
    REQUIRE(sedmi_confusion_counts(dp, dp, total, n_cls, 0.5f, 0, dcounts, nullptr) == SEDMI_OK);
    HIP_OK(hipDeviceSynchronize());
    sedmi_kernel_stat stats[8];
    REQUIRE(sedmi_profile_enable(1) == SEDMI_OK);
    REQUIRE(sedmi_threshold_i64(dp, total * n_cls, 0.5f, 0, dth, nullptr) == SEDMI_OK);
    REQUIRE(sedmi_decode_events(dp, total, n_cls, dstart, dcut, 4, 0.5f, 0, 1, ev[0], ev[1], ev[2], ev[3], cap, dn, ws, wsb, nullptr) == SEDMI_OK);
    REQUIRE(sedmi_profile_collect(stats, 2) >= 0);             // fewer slots than kernels: the rest are dropped
    REQUIRE(sedmi_profile_enable(0) == SEDMI_OK);
    for (void *q : {static_cast<void *>(dp), static_cast<void *>(dstart), static_cast<void *>(dcut), static_cast<void *>(dn), static_cast<void *>(ws),
                    static_cast<void *>(dth), static_cast<void *>(dcounts), static_cast<void *>(ev[0]), static_cast<void *>(ev[1]),
                    static_cast<void *>(ev[2]), static_cast<void *>(ev[3])})
        HIP_OK(hipFree(q));
    std::printf("decode ok\n");
}

int main() {
    REQUIRE(sedmi_version() >= 100);
    REQUIRE(sedmi_device_count() >= 1);
    check_mel();
    for (int n_class : {10, 1, 92}) check_crnn(n_class);
    check_shake();
    check_decode();
    std::printf("host asan check: all ok\n");
    return 0;
}
