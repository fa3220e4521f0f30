/*
 * sedmi.h — C ABI of libsedmi.so: the MI355X (gfx950) implementation of the
 * sound-event-detection inference hot path
 *
 *     mel frontend  ->  baseline CRNN (conv blocks + bidirectional GRU + head)
 *                   ->  threshold / run-length event decode
 *
 * The reference (Willi-bot/Sound-Event-Detection) is pure Python and has no
 * FFI of its own; each entry point below states which reference call it
 * stands in for (file:line under the reference's root).
 *
 * Conventions
 *   - every function returns 0 on success or a negative SEDMI_E* code; the
 *     text of the last error on the calling thread is sedmi_last_error();
 *   - no C++ exceptions cross this boundary, no torch types appear in it;
 *   - "device pointer" = memory of the current HIP device, owned by the
 *     caller (typically a PyTorch-ROCm tensor's data_ptr()); "host pointer" =
 *     ordinary CPU memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *     work is enqueued on it and the call returns without synchronising,
 *     except where a function says it synchronises;
 *   - handles own the device memory they allocate and free it in *_destroy;
 *     a handle belongs to the device that was current when it was created.
 */
#ifndef SEDMI_H
#define SEDMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEDMI_OK            0
#define SEDMI_EINVAL       -1   /* bad argument / unsupported shape          */
#define SEDMI_EHIP         -2   /* a HIP runtime call failed                 */
#define SEDMI_ENOMEM       -3   /* workspace or capacity too small           */
#define SEDMI_ENODEV       -4   /* no usable gfx950 device                   */
#define SEDMI_ERECURRENCE  -5   /* sedmi_shake_rcnn_check: the single-launch recurrence gave up, outputs invalid */

/* Library version (major*10000 + minor*100 + patch). */
int sedmi_version(void);
/* Message of the last failing call on this thread ("" if none). */
const char *sedmi_last_error(void);
/* Number of visible HIP devices, or a negative SEDMI_E* code. */
int sedmi_device_count(void);

/* ------------------------------------------------------------------------
 * Mel frontend.
 * Replaces librosa.feature.melspectrogram(y=, sr=, n_fft=, n_mels=,
 * hop_length=, win_length=) as called at feature_extraction.py:45-46
 * (periodic Hann window centred in n_fft, centre padding with zeros,
 * power 2.0, Slaney mel scale and area normalisation, fmin 0, fmax sr/2).
 * ------------------------------------------------------------------------ */
typedef struct sedmi_mel_plan sedmi_mel_plan;

/* flags for sedmi_melspec_f32 — both OFF reproduces the reference, which has
 * neither stage */
#define SEDMI_MEL_LOG        1  /* 10*log10(max(S, 1e-10)) after the projection */
#define SEDMI_MEL_NORMALISE  2  /* (x - mean[m]) * istd[m] per mel bin, after log */

/* Builds window, twiddle and sparse filterbank tables on the current device.
 * n_fft must be 1024, 2048 or 4096; 0 < win_length <= n_fft; hop_length > 0;
 * 0 < n_mels <= 1024. */
int sedmi_mel_plan_create(sedmi_mel_plan **plan, int sr, int n_fft, int win_length,
                          int hop_length, int n_mels);
int sedmi_mel_plan_destroy(sedmi_mel_plan *plan);
/* Frames per clip of n_samples samples: 1 + n_samples / hop_length. */
int64_t sedmi_mel_num_frames(const sedmi_mel_plan *plan, int64_t n_samples);
/* Copies the dense float32 filterbank [n_mels, 1 + n_fft/2] to a host buffer
 * (for inspection and tests). */
int sedmi_mel_filterbank(const sedmi_mel_plan *plan, float *host_out);

/* y: device float32 [n_clips, n_samples] (contiguous).
 * out: device float32 [n_clips, n_mels, T], T = sedmi_mel_num_frames(): the
 * mel-major array librosa returns.  The reference then re-reads the same
 * bytes as [T, n_mels] (np.reshape at feature_extraction.py:49-50), which
 * needs no data movement: pass `out` on as the [n_clips, T, n_mels] feature.
 * mean / istd: device float32 [n_mels], read only with SEDMI_MEL_NORMALISE.
 * Limits (SEDMI_EINVAL beyond them): n_samples <= 2^30 - 2^14 per clip (samples are
 * addressed with 32-bit byte offsets inside a clip), n_clips * ceil(T / 32) < 2^31 - 2^20. */
int sedmi_melspec_f32(const sedmi_mel_plan *plan, const float *y, int64_t n_clips,
                      int64_t n_samples, int flags, const float *mean, const float *istd,
                      float *out, void *stream);

/* The same from 16-bit PCM as a WAV file holds it (device int16 [n_clips, n_samples]): each sample is
 * taken as sample / 32768, which is exact in float32, so the output is bit for bit the one
 * sedmi_melspec_f32 gives for the converted samples (librosa.load at feature_extraction.py:37 does
 * that conversion on the CPU and hands float32 on).  Half the bytes across PCIe and from HBM. */
int sedmi_melspec_i16(const sedmi_mel_plan *plan, const int16_t *y, int64_t n_clips,
                      int64_t n_samples, int flags, const float *mean, const float *istd,
                      float *out, void *stream);

/* Per-mel-bin sum and sum of squares over all clips and frames of a
 * [n_clips, n_mels, T] array, accumulated in float64 into device buffers
 * sum[n_mels], sumsq[n_mels] (ADDED to their current contents, so that ranks
 * can all-reduce them).  Fixed summation order, no atomics: the same array
 * gives the same sums bit for bit on every call.  Supports the optional normalisation stage; the
 * reference has no counterpart (its --normalize flag is never read:
 * training.py:91-94). */
int sedmi_mel_stats_f64(const float *mel, int64_t n_clips, int n_mels, int64_t T,
                        double *sum, double *sumsq, void *stream);

/* The optional stages applied IN PLACE to a [n_clips, n_mels, T] array that holds
 * the power mel spectrogram (or, with SEDMI_MEL_NORMALISE alone, its log): the
 * second pass of a dataset-level normalisation, whose mean and standard
 * deviation are only known once every clip's statistics have been summed
 * (sedmi_mel_stats_f64, all-reduced across ranks).  Same expressions, hence the
 * same bits, as sedmi_melspec_f32 with these flags.  No counterpart in the
 * reference (training.py:91-94: --normalize is parsed and never read). */
int sedmi_mel_apply_f32(float *mel, int64_t n_clips, int n_mels, int64_t T, int flags,
                        const float *mean, const float *istd, void *stream);

/* ------------------------------------------------------------------------
 * Baseline CRNN, eval-mode forward.
 * Replaces CRNN.forward (nnet/CRNN.py:191-268) for the one configuration the
 * reference constructs (training.py:207-212): 7 x [Conv3x3 -> BatchNorm(eps
 * 1e-3) -> GLU -> AvgPool] (nnet/CNN.py:66-98), 2-layer bidirectional GRU
 * (nnet/RNN.py:19-30), strong head sigmoid(Linear) (nnet/CRNN.py:142-143).
 * ------------------------------------------------------------------------ */
typedef struct sedmi_crnn sedmi_crnn;

#define SEDMI_CRNN_LAYERS 7

/* Host pointers to float32 tensors in the reference's state_dict layout. */
typedef struct sedmi_crnn_weights {
    int n_class;                                   /* rows of dense.weight            */
    int n_freq;                                    /* input feature width (128)        */
    int filters[SEDMI_CRNN_LAYERS];                /* 16,32,64,128,128,128,128         */
    int pool_t[SEDMI_CRNN_LAYERS];                 /* time pooling per block (2,1,...) */
    int pool_f[SEDMI_CRNN_LAYERS];                 /* freq pooling per block (2,2,...) */
    int hidden;                                    /* GRU hidden size (128)            */
    float bn_eps;                                  /* 1e-3                             */
    const float *conv_w[SEDMI_CRNN_LAYERS];        /* cnn.cnn.conv{i}.weight [Co,Ci,3,3] */
    const float *conv_b[SEDMI_CRNN_LAYERS];        /* cnn.cnn.conv{i}.bias   [Co]        */
    const float *bn_w[SEDMI_CRNN_LAYERS];          /* cnn.cnn.batchnorm{i}.weight        */
    const float *bn_b[SEDMI_CRNN_LAYERS];          /* ....bias                           */
    const float *bn_mean[SEDMI_CRNN_LAYERS];       /* ....running_mean                   */
    const float *bn_var[SEDMI_CRNN_LAYERS];        /* ....running_var                    */
    const float *glu_w[SEDMI_CRNN_LAYERS];         /* cnn.cnn.glu{i}.linear.weight [Co,Co] */
    const float *glu_b[SEDMI_CRNN_LAYERS];         /* ....bias [Co]                      */
    /* rnn.rnn.{weight_ih,weight_hh,bias_ih,bias_hh}_l{L}[_reverse]; index [L][dir] */
    const float *gru_w_ih[2][2];
    const float *gru_w_hh[2][2];
    const float *gru_b_ih[2][2];
    const float *gru_b_hh[2][2];
    const float *dense_w;                          /* dense.weight [n_class, 2*hidden]   */
    const float *dense_b;                          /* dense.bias   [n_class]             */
} sedmi_crnn_weights;

/* Folds BatchNorm into the convolutions, repacks every matrix into the
 * layout its kernel reads, and uploads the result to the current device. */
int sedmi_crnn_create(sedmi_crnn **model, const sedmi_crnn_weights *w);
int sedmi_crnn_destroy(sedmi_crnn *model);
/* Arithmetic of the whole model's contractions (all seven conv blocks, the GRU input projections
 * and the recurrence; the strong head is always float32):
 *   SEDMI_PRECISION_F32        v_mfma_f32_32x32x2_f32 / float32 FMAs, exact float32 products;
 *   SEDMI_PRECISION_SPLIT_F16  every float32 operand split into two f16 halves (21-22 significand
 *                              bits), three f16 MFMAs per product, float32 accumulation: about
 *                              4x the relative rounding of float32 at ~5x the matrix-core rate.
 * Range in SPLIT_F16 mode.  The input features (block 0 reads the raw power mel spectrogram, whose
 * magnitude follows the audio level squared) may have ANY float32 magnitude: each tile of input rows
 * is scaled by an exact power of two to max |x| in [2^13, 2^14) before it is split, and the factor is undone
 * exactly behind the matrix product, so int16-scale PCM, loud recordings at n_fft 4096 or very quiet
 * ones lose nothing.  Activations between the layers are split as they are: a value of magnitude
 * >= 65504 there saturates.  With BatchNorm statistics that belong to the features being fed (any
 * trained checkpoint on its own data) they are O(1..100); a checkpoint evaluated on features many
 * orders of magnitude louder than it was trained on can exceed the range -- use SEDMI_PRECISION_F32
 * for that (the reference's float32 path has no such limit).
 * The default is SEDMI_PRECISION_SPLIT_F16. */
#define SEDMI_PRECISION_F32        0
#define SEDMI_PRECISION_SPLIT_F16  1
int sedmi_crnn_set_precision(sedmi_crnn *model, int precision);
/* Output frames for T input frames (product of the time poolings, floored). */
int64_t sedmi_crnn_out_frames(const sedmi_crnn *model, int64_t T);
/* Scratch bytes sedmi_crnn_forward_f32 needs for a batch of n_clips x T. */
size_t sedmi_crnn_workspace_bytes(const sedmi_crnn *model, int64_t n_clips, int64_t T);

/* feats: device float32 [n_clips, T, n_freq]; probs: device float32
 * [n_clips, T', n_class] (probabilities, as CRNN.forward returns them);
 * ws: device scratch of at least sedmi_crnn_workspace_bytes(). */
int sedmi_crnn_forward_f32(const sedmi_crnn *model, const float *feats, int64_t n_clips,
                           int64_t T, float *probs, void *ws, size_t ws_bytes, void *stream);

/* Stage-level entry points (same arithmetic as inside the full forward, except
 * that the full forward in SPLIT_F16 mode with at most 16 classes forms the
 * head's products inside the last GRU layer, in split-f16 arithmetic, where
 * sedmi_crnn_head_f32 uses the float32 matrix instruction: ~1e-7 apart; used
 * by the layer-level parity tests).
 * sedmi_crnn_cnn_f32:  feats [n, T, n_freq] -> seq [n, T', C_last]
 *   (= CNN.forward + squeeze(-1).permute(0,2,1), nnet/CRNN.py:200,212-213).
 * sedmi_crnn_gru_f32:  seq [n, T', C_last] -> rec [n, T', 2*hidden]
 *   (= BidirectionalGRU.forward, nnet/RNN.py:28-30).
 * sedmi_crnn_head_f32: rec [n*T', 2*hidden] -> probs [n*T', n_class]. */
int sedmi_crnn_cnn_f32(const sedmi_crnn *model, const float *feats, int64_t n_clips, int64_t T,
                       float *seq, void *ws, size_t ws_bytes, void *stream);
int sedmi_crnn_gru_f32(const sedmi_crnn *model, const float *seq, int64_t n_clips, int64_t Tp,
                       float *rec, void *ws, size_t ws_bytes, void *stream);
int sedmi_crnn_head_f32(const sedmi_crnn *model, const float *rec, int64_t n_rows,
                        float *probs, void *stream);

/* ------------------------------------------------------------------------
 * ShakeRCNN eval-mode forward — replaces ShakeRCNN.forward (models.py:124-142)
 * with its three ShakeCNNBlocks (models.py:155-180) and ShakeShakeBlocks
 * (models.py:70-100).
 *
 * Weight pointers are HOST float32 arrays in the layouts of the reference
 * state_dict; index b = block (cnn1, cnn2, cnn3):
 *   conv_w[b]  cnn{b+1}.conv.weight            [2*out, in, k, k]  (k = 5, 3, 3; in = 1, 16, 32; out = 16, 32, 64)
 *   conv_b[b]  cnn{b+1}.conv.bias              [2*out]
 *   bn1_*[b]   cnn{b+1}.bn1.{weight,bias,running_mean,running_var}   [2*out]   (eps 1e-5)
 *   br1_w/br2_w[b]  cnn{b+1}.shake.branch{1,2}.weight  [out, out, 3, 3];  br1_b/br2_b: bias [out]
 *   bn2_*[b]   cnn{b+1}.bn2.*                  [out]
 *   gru_*[d]   rnn.{weight_ih,weight_hh,bias_ih,bias_hh}_l0{,_reverse}  [3072,6144] [3072,1024] [3072] [3072]
 *   ln_w/ln_b  layernorm.{weight,bias}         [2048]
 *   fc_w/fc_b  fc1.{weight,bias}               [n_class, 2048] [n_class]
 * ------------------------------------------------------------------------ */
typedef struct sedmi_shake_rcnn sedmi_shake_rcnn;

typedef struct sedmi_shake_cnn_weights {
    const float *conv_w[3], *conv_b[3];
    const float *bn1_w[3], *bn1_b[3], *bn1_mean[3], *bn1_var[3];
    const float *br1_w[3], *br1_b[3], *br2_w[3], *br2_b[3];
    const float *bn2_w[3], *bn2_b[3], *bn2_mean[3], *bn2_var[3];
} sedmi_shake_cnn_weights;

typedef struct sedmi_shake_rcnn_weights {
    int n_class;
    sedmi_shake_cnn_weights cnn;
    const float *gru_w_ih[2], *gru_w_hh[2], *gru_b_ih[2], *gru_b_hh[2];
    const float *ln_w, *ln_b, *fc_w, *fc_b;
} sedmi_shake_rcnn_weights;

/* ShakeRCNN(num_classes) + load_state_dict (models.py:105-121): repacks the
 * weights for the kernels and uploads them to the current device. */
int sedmi_shake_rcnn_create(sedmi_shake_rcnn **out, const sedmi_shake_rcnn_weights *w);
int sedmi_shake_rcnn_destroy(sedmi_shake_rcnn *model);

/* Device scratch needed for n_clips clips of T frames x 768 bins (0 on bad input). */
size_t sedmi_shake_rcnn_workspace_bytes(const sedmi_shake_rcnn *model, int64_t n_clips, int64_t T);

/* feats [n_clips, T, 768] (device) -> logits [n_clips, T/4, n_class] (device),
 * T/4 floored as the reference's three AvgPool2d stages floor odd sizes
 * (models.py:109-111: 1001 frames -> 500 -> 250 -> 250).  alphas: three HOST floats, the ShakeShake mixing
 * weights of cnn1..cnn3; the reference draws each with torch.rand(1) on every
 * eval-mode forward (models.py:88-91), so the caller supplies them (draw them
 * the same way to reproduce the reference's stream, or pass 0.5 for the
 * expectation).  Dropout is the identity in eval mode. */
int sedmi_shake_rcnn_forward_f32(const sedmi_shake_rcnn *model, const float *feats, int64_t n_clips,
                                 int64_t T, const float *alphas, float *logits,
                                 void *ws, size_t ws_bytes, void *stream);

/* How the H = 1024 recurrence of sedmi_shake_rcnn_forward_f32 runs.
 *   SEDMI_RECURRENCE_AUTO      up to 192 clips per call: ONE launch for all steps, W_hh resident in the registers
 *                              of 128 (or, from 33 clips, 256) workgroups that exchange h through device memory
 *                              every step; beyond that, one launch per step.  The single launch needs all its
 *                              workgroups resident at the same time, one per compute unit: the GPU must not be
 *                              shared with other kernels (another stream, another process) while it runs, and no
 *                              compute unit may be masked off.  A workgroup that waits for a neighbour that never
 *                              arrives gives up after a bounded number of polls (seconds), the kernel drains, the
 *                              logits of that forward come out as NaN and sedmi_shake_rcnn_check reports it.
 *   SEDMI_RECURRENCE_PER_STEP  always one launch per step (no residency requirement; 1.7-2x slower for few,
 *                              long recordings).
 * The default is SEDMI_RECURRENCE_AUTO.  No counterpart in the reference (nn.GRU at models.py:114,136). */
#define SEDMI_RECURRENCE_AUTO      0
#define SEDMI_RECURRENCE_PER_STEP  1
int sedmi_shake_rcnn_set_recurrence(sedmi_shake_rcnn *model, int mode);

/* Waits for `stream` (it synchronises) and reports whether the recurrence of the
 * sedmi_shake_rcnn_forward_f32 call that last used workspace `ws` (same n_clips, T)
 * ran to completion: SEDMI_OK, or SEDMI_ERECURRENCE when a workgroup of the
 * single-launch recurrence gave up (the logits of that call are NaN; nothing else
 * is affected).  Callers that must never act on NaN logits call it before they
 * read the logits. */
int sedmi_shake_rcnn_check(const sedmi_shake_rcnn *model, const void *ws, size_t ws_bytes,
                           int64_t n_clips, int64_t T, void *stream);

/* Stage-level entry point for parity tests: the three ShakeCNNBlocks followed
 * by transpose(1,2) + flatten (models.py:127-134): feats -> seq [n_clips, T/4, 6144]
 * with the reference's (channel, frequency) order along the last axis. */
int sedmi_shake_cnn_f32(const sedmi_shake_rcnn *model, const float *feats, int64_t n_clips, int64_t T,
                        const float *alphas, float *seq, void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------
 * ShakeTransformer eval-mode forward — replaces ShakeTransformer.forward
 * (models.py:204-221): the same three ShakeCNNBlocks, Linear(6144, 1024),
 * n_layers x torch.nn.TransformerEncoderLayer(d_model 1024, nhead 8,
 * dim_feedforward 2048, ReLU, post-LayerNorm (eps 1e-5), batch_first; no mask,
 * no positional encoding) (models.py:195-196), Linear(1024, n_class).
 *
 * Host float32 pointers in the state_dict layouts; layer i is
 * encoder.layers.{i}.:  in_proj_w self_attn.in_proj_weight [3072,1024],
 * in_proj_b [3072], out_proj_w self_attn.out_proj.weight [1024,1024],
 * out_proj_b [1024], lin1_w linear1.weight [2048,1024], lin1_b [2048],
 * lin2_w linear2.weight [1024,2048], lin2_b [1024], norm{1,2}_{w,b} [1024];
 * fc1_w fc1.weight [1024,6144], fc2_w fc2.weight [n_class,1024].
 * ------------------------------------------------------------------------ */
#define SEDMI_SHAKE_MAX_LAYERS 12

typedef struct sedmi_shake_transformer sedmi_shake_transformer;

typedef struct sedmi_shake_encoder_layer_weights {
    const float *in_proj_w, *in_proj_b, *out_proj_w, *out_proj_b;
    const float *lin1_w, *lin1_b, *lin2_w, *lin2_b;
    const float *norm1_w, *norm1_b, *norm2_w, *norm2_b;
} sedmi_shake_encoder_layer_weights;

typedef struct sedmi_shake_transformer_weights {
    int n_class;
    int n_layers;                       /* 6 in the reference (models.py:196) */
    sedmi_shake_cnn_weights cnn;
    const float *fc1_w, *fc1_b;
    sedmi_shake_encoder_layer_weights layers[SEDMI_SHAKE_MAX_LAYERS];
    const float *fc2_w, *fc2_b;
} sedmi_shake_transformer_weights;

/* ShakeTransformer(num_classes, dropout) + load_state_dict (models.py:185-202). */
int sedmi_shake_transformer_create(sedmi_shake_transformer **out, const sedmi_shake_transformer_weights *w);
int sedmi_shake_transformer_destroy(sedmi_shake_transformer *model);
size_t sedmi_shake_transformer_workspace_bytes(const sedmi_shake_transformer *model, int64_t n_clips, int64_t T);

/* feats [n_clips, T, 768] (device) -> logits [n_clips, T/4, n_class] (device);
 * T/4 floored; alphas as for sedmi_shake_rcnn_forward_f32. */
int sedmi_shake_transformer_forward_f32(const sedmi_shake_transformer *model, const float *feats,
                                        int64_t n_clips, int64_t T, const float *alphas, float *logits,
                                        void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------
 * BasicRCNN eval-mode forward — replaces BasicRCNN.forward (models.py:45-66)
 * for the constructor defaults the reference uses (training.py:201:
 * BasicRCNN(num_classes, 64)): three 3x3 convs with 128 channels,
 * MaxPool((1,5), stride 1) twice and MaxPool(2,2), ReLU after each pool,
 * the reshape-not-permute of models.py:59-61, a one-directional GRU with 32
 * hidden units, ReLU, Linear(32, n_class).  Returns logits.
 *
 * Host float32 pointers in the state_dict layouts: conv_w[i] conv{i+1}.weight
 * ([128,1,3,3], [128,128,3,3], [128,128,3,3]), conv_b[i] conv{i+1}.bias [128],
 * gru_* gru.{weight_ih,weight_hh,bias_ih,bias_hh}_l0 ([96, 64*(F-8)], [96,32],
 * [96], [96]), fc_w fc1.weight [n_class, 32], fc_b fc1.bias [n_class].
 * ------------------------------------------------------------------------ */
typedef struct sedmi_basic_rcnn sedmi_basic_rcnn;

typedef struct sedmi_basic_rcnn_weights {
    int n_class;
    int frequency_dim;                  /* F: feature width, >= 10 (64 in the reference); GRU input 128 * ((F - 8) / 2) */
    const float *conv_w[3], *conv_b[3];
    const float *gru_w_ih, *gru_w_hh, *gru_b_ih, *gru_b_hh;
    const float *fc_w, *fc_b;
} sedmi_basic_rcnn_weights;

int sedmi_basic_rcnn_create(sedmi_basic_rcnn **out, const sedmi_basic_rcnn_weights *w);
int sedmi_basic_rcnn_destroy(sedmi_basic_rcnn *model);
size_t sedmi_basic_rcnn_workspace_bytes(const sedmi_basic_rcnn *model, int64_t n_clips, int64_t T);
/* feats [n_clips, T, F] (device) -> logits [n_clips, T/2, n_class] (device). */
int sedmi_basic_rcnn_forward_f32(const sedmi_basic_rcnn *model, const float *feats, int64_t n_clips, int64_t T,
                                 float *logits, void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------
 * Thresholding and event decoding.
 * ------------------------------------------------------------------------ */

/* out[i] = (sigmoid?(p[i]) > thr) ? 1 : 0 as int64 — the thresholding half of
 * evaluate() (evaluation.py:19-21): strict '>', int64 result. */
int sedmi_threshold_i64(const float *p, int64_t n, float thr, int apply_sigmoid,
                        int64_t *out, void *stream);

/* Run-length event decode of get_prediction_from_raw_output
 * (evaluation.py:49-80) for many recordings at once.
 *   p           device float32 [total_frames, n_class], recordings back to back
 *   seg_start   device int64 [n_seg + 1]: recording s owns frames
 *               [seg_start[s], seg_start[s+1])
 *   seg_cutoff  device int64 [n_seg]: frames kept per recording
 *               (int(duration / block) + 1, evaluation.py:58); values larger
 *               than the recording are clamped to it
 *   median_win  odd window of the optional median filter over time (1 = off,
 *               which is the reference's behaviour; reflect boundary)
 * Output, ordered by recording, then class, then onset (evaluation.py:61):
 *   ev_seg/ev_cls/ev_on/ev_off  device int32 [capacity]; offsets are
 *               exclusive frame indices relative to the recording
 *   n_events    device int64 [1]: number of events found (may exceed
 *               capacity, in which case only the first `capacity` are stored)
 *   ws          device scratch >= sedmi_decode_workspace_bytes(n_seg, n_class)
 * Event times in seconds are frame_index * (block_ms / 1000.0) in float64,
 * computed by the caller exactly as evaluation.py:55,75-76 does. */
size_t sedmi_decode_workspace_bytes(int64_t n_seg, int n_class);
int sedmi_decode_events(const float *p, int64_t total_frames, int n_class,
                        const int64_t *seg_start, const int64_t *seg_cutoff, int64_t n_seg,
                        float thr, int apply_sigmoid, int median_win,
                        int32_t *ev_seg, int32_t *ev_cls, int32_t *ev_on, int32_t *ev_off,
                        int64_t capacity, int64_t *n_events, void *ws, size_t ws_bytes,
                        void *stream);

/* Allocation-free variant of sedmi_decode_events.  It has no counterpart in the
 * reference; it exists for callers that cannot allocate device memory or hand
 * over a workspace: code running under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  Same arguments, results and ordering as
 * sedmi_decode_events minus `ws` / `ws_bytes`: the per-(recording, class)
 * counters and offsets live in a static device array inside the library
 * (about 770 KB per GPU, held until the process ends).  The call allocates
 * nothing, on the host side of the runtime or on the device, and records no
 * profiling events.
 *   - n_seg * n_class must not exceed SEDMI_DECODE_STATIC_MAX_PAIRS
 *     (SEDMI_EINVAL beyond it);
 *   - the array's device address is looked up on the first call on each
 *     device (hipGetSymbolAddress): make that first call outside graph capture;
 *   - all calls on one GPU share the array: calls that overlap on one GPU
 *     (different streams or threads) race on it and must be serialised by the
 *     caller.  Nothing in it needs clearing between calls. */
#define SEDMI_DECODE_STATIC_MAX_PAIRS 65536
int sedmi_decode_events_static(const float *p, int64_t total_frames, int n_class,
                               const int64_t *seg_start, const int64_t *seg_cutoff, int64_t n_seg,
                               float thr, int apply_sigmoid, int median_win,
                               int32_t *ev_seg, int32_t *ev_cls, int32_t *ev_on, int32_t *ev_off,
                               int64_t capacity, int64_t *n_events, void *stream);

/* Per-class confusion counts of thresholded predictions against 0/1 labels —
 * the sufficient statistics of the sklearn scores in evaluate()
 * (evaluation.py:26-45).  labels: device float32 [n_rows, n_class] (the
 * dtype data_loading.py:39 produces); counts: device int64 [n_class, 4] =
 * TP, FP, FN, TN, ADDED to current contents. */
int sedmi_confusion_counts(const float *p, const float *labels, int64_t n_rows, int n_class,
                           float thr, int apply_sigmoid, int64_t *counts, void *stream);

/* ------------------------------------------------------------------------
 * Per-kernel timing (no counterpart in the reference; it exists so that a
 * benchmark can quote each kernel's duration and its algorithmic FLOP / byte
 * rate from HIP events recorded on the very stream the kernels run on).
 * While enabled, every kernel launch of this library is bracketed by two
 * events.  sedmi_profile_collect waits for the recorded events (it
 * synchronises), sums them per kernel name and clears the record.
 * Process-wide; enable/collect from one thread at a time.
 * ------------------------------------------------------------------------ */
typedef struct sedmi_kernel_stat {
    char name[48];
    int64_t launches;
    double ms;        /* summed device time of those launches           */
    double flops;     /* summed algorithmic floating-point operations   */
    double bytes;     /* summed algorithmic bytes (inputs read once + outputs written once) */
} sedmi_kernel_stat;

int sedmi_profile_enable(int on);
/* Returns the number of entries written (<= capacity) or a negative code. */
int sedmi_profile_collect(sedmi_kernel_stat *out, int capacity);

#ifdef __cplusplus
}
#endif
#endif /* SEDMI_H */
