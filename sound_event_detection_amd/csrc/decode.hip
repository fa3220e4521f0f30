// Thresholding, optional median smoothing and run-length event decoding on
// gfx950.  Stands in for evaluation.py:19-21 (threshold half of evaluate) and
// evaluation.py:49-80 (get_prediction_from_raw_output): strict '>' against the
// float32 threshold, events per class as [onset, offset) frame indices cut at
// `cutoff` frames, emitted class-major then by time.
//
// HBM-bound byte work (10 KB of probabilities per 10 s clip): one wave owns one
// (recording, class) column, walks it 64 frames at a time, finds run starts and
// ends with wave ballots, and ranks them with popcounts; the k-th start pairs
// with the k-th end, so no search is needed.  A count pass, an exclusive scan
// over (recording, class) pairs and a write pass give the reference's output
// order without atomics, hence deterministically.
//
// The median filter (which the reference does not have; window 1 = off)
// commutes with a monotone threshold, so it is a sliding majority vote over the
// thresholded bits: each wave stages 64 + 2*half bits in LDS (reflect boundary)
// and every lane sums its window from there.
#include "common.h"

#include <mutex>

namespace {

constexpr int kMaxHalf = 64;                 // median window <= 129
constexpr int kWavesPerBlock = 4;

__device__ __forceinline__ int over_threshold(float v, float thr, int apply_sigmoid) {
    if (apply_sigmoid) v = 1.0f / (1.0f + expf(-v));
    return v > thr ? 1 : 0;
}

__device__ __forceinline__ long reflect_index(long j, long n) {
    // scipy.ndimage 'reflect': (d c b a | a b c d | d c b a)
    while (j < 0 || j >= n) j = j < 0 ? -j - 1 : 2 * n - j - 1;
    return j;
}

__global__ __launch_bounds__(256) void threshold_i64_kernel(const float *p, long n, float thr,
                                                            int apply_sigmoid, long long *out) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += 256L * gridDim.x)
        out[i] = over_threshold(p[i], thr, apply_sigmoid);
}

template <bool WRITE>
__global__ __launch_bounds__(64 * kWavesPerBlock) void decode_kernel(
    const float *p, int n_cls, const long long *seg_start, const long long *seg_cutoff, long n_seg,
    float thr, int apply_sigmoid, int half, int *counts, const long long *offsets, int *ev_seg,
    int *ev_cls, int *ev_on, int *ev_off, long capacity) {
    __shared__ unsigned char bits[kWavesPerBlock][64 + 2 * kMaxHalf];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long pair = static_cast<long>(blockIdx.x) * kWavesPerBlock + wave;
    if (pair >= n_seg * n_cls) return;                     // whole wave leaves together
    const long seg = pair / n_cls;
    const int c = static_cast<int>(pair - seg * n_cls);
    const long s0 = seg_start[seg];
    const long len = seg_start[seg + 1] - s0;
    long limit = seg_cutoff[seg];
    limit = limit < 0 ? 0 : (limit > len ? len : limit);
    const float *col = p + s0 * n_cls + c;

    int carry = 0;                  // filtered bit of frame base-1
    long n_start = 0, n_end = 0;    // events opened / closed so far
    const long out0 = WRITE ? offsets[pair] : 0;
    // frame index `limit` acts as a trailing zero so that a run reaching the cut is closed there
    for (long base = 0; base <= limit && limit > 0; base += 64) {
        const long i = base + lane;
        int cur = 0;
        if (half == 0) {
            if (i < limit) cur = over_threshold(col[i * n_cls], thr, apply_sigmoid);
        } else {
            for (int k = lane; k < 64 + 2 * half; k += 64) {
                const long j = reflect_index(base - half + k, len);
                bits[wave][k] = static_cast<unsigned char>(over_threshold(col[j * n_cls], thr, apply_sigmoid));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            int votes = 0;
            for (int k = 0; k <= 2 * half; ++k) votes += bits[wave][lane + k];
            if (i < limit) cur = votes > half ? 1 : 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        int prev = __shfl_up(cur, 1);
        if (lane == 0) prev = carry;
        const bool is_start = cur && !prev;
        const bool is_end = !cur && prev && i <= limit;
        const unsigned long long m_start = __ballot(is_start), m_end = __ballot(is_end);
        if (WRITE) {
            const unsigned long long below = (1ull << lane) - 1ull;
            if (is_start) {
                const long slot = out0 + n_start + __popcll(m_start & below);
                if (slot < capacity) {
                    ev_seg[slot] = static_cast<int>(seg);
                    ev_cls[slot] = c;
                    ev_on[slot] = static_cast<int>(i);
                }
            }
            if (is_end) {
                const long slot = out0 + n_end + __popcll(m_end & below);
                if (slot < capacity) ev_off[slot] = static_cast<int>(i);
            }
        }
        n_start += __popcll(m_start);
        n_end += __popcll(m_end);
        carry = __shfl(cur, 63);
    }
    if (!WRITE && lane == 0) counts[pair] = static_cast<int>(n_start);
}

// exclusive scan of counts[n] into offsets[n], total into *total (one workgroup)
__global__ __launch_bounds__(1024) void scan_kernel(const int *counts, long n, long long *offsets,
                                                    long long *total) {
    __shared__ long long part[1024];
    const int tid = threadIdx.x;
    const long per = (n + 1023) / 1024;
    const long lo = tid * per, hi = lo + per < n ? lo + per : n;
    long long s = 0;
    for (long i = lo; i < hi; ++i) s += counts[i];
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const long long v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    long long run = tid ? part[tid - 1] : 0;
    for (long i = lo; i < hi; ++i) {
        offsets[i] = run;
        run += counts[i];
    }
    if (tid == 1023) *total = part[1023];
}

__global__ __launch_bounds__(256) void confusion_kernel(const float *p, const float *labels, long n_rows,
                                                        int n_cls, float thr, int apply_sigmoid,
                                                        unsigned long long *counts) {
    // grid.x = class, grid.y = row slab; [TP, FP, FN, TN] per class
    const int c = blockIdx.x;
    unsigned int tp = 0, fp = 0, fn = 0, tn = 0;
    for (long r = blockIdx.y * 256L + threadIdx.x; r < n_rows; r += 256L * gridDim.y) {
        const int yh = over_threshold(p[r * n_cls + c], thr, apply_sigmoid);
        const int y = static_cast<long long>(labels[r * n_cls + c]) != 0;   // .astype(int64), evaluation.py:27
        tp += y & yh; fp += (!y) & yh; fn += y & (!yh); tn += (!y) & (!yh);
    }
    __shared__ unsigned int sh[4][256];
    sh[0][threadIdx.x] = tp; sh[1][threadIdx.x] = fp; sh[2][threadIdx.x] = fn; sh[3][threadIdx.x] = tn;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o)
            for (int k = 0; k < 4; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 4) atomicAdd(&counts[c * 4 + threadIdx.x], static_cast<unsigned long long>(sh[threadIdx.x][0]));
}

}  // namespace

// Library-owned scratch of sedmi_decode_events_static: counters (int) and offsets (int64) for up to
// SEDMI_DECODE_STATIC_MAX_PAIRS (recording, class) pairs, laid out as decode_launch expects.  Every GPU
// the library runs on holds its own copy of this array until the process ends.
constexpr size_t kStaticScratchBytes =
    (SEDMI_DECODE_STATIC_MAX_PAIRS * sizeof(int) + 255) / 256 * 256 +
    (SEDMI_DECODE_STATIC_MAX_PAIRS * sizeof(long long) + 255) / 256 * 256 + 256;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

// Device address of `scratch` on the current device.  In host code the array's name is only a handle for
// the runtime's symbol table, never a device address: the address is looked up once per device (with that
// device current) and kept; every use of the array's memory from the host goes through it.
static int static_scratch_address(void **out) {
    constexpr int kMaxDevices = 64;
    static std::mutex mutex;
    static void *dev_ptr[kMaxDevices] = {nullptr};
    int dev = -1;
    SEDMI_CHECK_HIP(hipGetDevice(&dev));
    SEDMI_REQUIRE(dev >= 0 && dev < kMaxDevices, "device index %d outside [0, %d)", dev, kMaxDevices);
    std::lock_guard<std::mutex> lock(mutex);
    if (dev_ptr[dev] == nullptr) {
        void *ptr = nullptr;
        SEDMI_CHECK_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        SEDMI_REQUIRE(ptr != nullptr, "hipGetSymbolAddress returned a null address for the static scratch");
        dev_ptr[dev] = ptr;
    }
    *out = dev_ptr[dev];
    return SEDMI_OK;
}

extern "C" {

int sedmi_threshold_i64(const float *p, int64_t n, float thr, int apply_sigmoid, int64_t *out, void *stream) {
    SEDMI_REQUIRE(n >= 0, "negative size");
    if (n == 0) return SEDMI_OK;
    SEDMI_REQUIRE(p && out, "null data pointer");
    const unsigned grid = static_cast<unsigned>(std::min<int64_t>(sedmi::ceil_div(n, 256), 4096));
    hipLaunchKernelGGL(threshold_i64_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), p,
                       static_cast<long>(n), thr, apply_sigmoid, reinterpret_cast<long long *>(out));
    SEDMI_CHECK_LAUNCH();
    return SEDMI_OK;
}

size_t sedmi_decode_workspace_bytes(int64_t n_seg, int n_class) {
    if (n_seg < 0 || n_class <= 0) return 0;
    const size_t pairs = static_cast<size_t>(n_seg) * n_class;
    return sedmi::align_up(pairs * sizeof(int), 256) + sedmi::align_up(pairs * sizeof(long long), 256) + 256;
}

// Argument checks shared by both decode entry points (everything but the scratch).
static int decode_check(const float *p, int64_t total_frames, int n_class, const int64_t *seg_start,
                        const int64_t *seg_cutoff, int64_t n_seg, int median_win, int32_t *ev_seg,
                        int32_t *ev_cls, int32_t *ev_on, int32_t *ev_off, int64_t capacity, int64_t *n_events) {
    SEDMI_REQUIRE(n_class > 0 && n_seg >= 0 && total_frames >= 0 && capacity >= 0, "bad shape");
    SEDMI_REQUIRE(median_win >= 1 && (median_win & 1) && median_win <= 2 * kMaxHalf + 1,
                  "median window must be odd and in [1, %d] (got %d)", 2 * kMaxHalf + 1, median_win);
    SEDMI_REQUIRE(total_frames < (1LL << 31), "more than 2^31 frames in one call");
    SEDMI_REQUIRE(n_events != nullptr, "n_events is null");
    if (n_seg == 0) return SEDMI_OK;
    SEDMI_REQUIRE(p || total_frames == 0, "null probability pointer");
    SEDMI_REQUIRE(seg_start && seg_cutoff, "null segment table");
    SEDMI_REQUIRE(capacity == 0 || (ev_seg && ev_cls && ev_on && ev_off), "null event buffer");
    return SEDMI_OK;
}

// count pass -> exclusive scan -> write pass; `work` (device memory: the caller's workspace or the looked-up
// address of the static array, never the array's name) holds the per-(recording, class) counters
// (int) followed, 256-byte aligned, by their scanned offsets (int64).  `timed` = bracket the launches
// with profiling events when profiling is on (creating events allocates, so the allocation-free entry
// point passes false).
static int decode_launch(const float *p, int64_t total_frames, int n_class, const int64_t *seg_start,
                         const int64_t *seg_cutoff, int64_t n_seg, float thr, int apply_sigmoid, int median_win,
                         int32_t *ev_seg, int32_t *ev_cls, int32_t *ev_on, int32_t *ev_off, int64_t capacity,
                         int64_t *n_events, void *work, bool timed, hipStream_t s) {
    const long pairs = static_cast<long>(n_seg) * n_class;
    int *counts = static_cast<int *>(work);
    long long *offsets = reinterpret_cast<long long *>(static_cast<char *>(work) +
                                                       sedmi::align_up(pairs * sizeof(int), 256));
    const unsigned grid = static_cast<unsigned>(sedmi::ceil_div(pairs, kWavesPerBlock));
    const int half = median_win / 2;
    const long long *ss = reinterpret_cast<const long long *>(seg_start);
    const long long *sc = reinterpret_cast<const long long *>(seg_cutoff);
    const double dbytes = 4.0 * total_frames * n_class;
    timed = timed && sedmi::profile_enabled();
    sedmi::ProfileTicket rec{-1, 0};
    if (timed) rec = sedmi::profile_begin("decode_kernel<false>", s, 0.0, dbytes);
    hipLaunchKernelGGL(decode_kernel<false>, dim3(grid), dim3(64 * kWavesPerBlock), 0, s, p, n_class, ss, sc,
                       static_cast<long>(n_seg), thr, apply_sigmoid, half, counts, offsets, ev_seg, ev_cls,
                       ev_on, ev_off, static_cast<long>(capacity));
    if (timed) sedmi::profile_end(rec, s);
    SEDMI_CHECK_LAUNCH();
    if (timed) rec = sedmi::profile_begin("scan_kernel", s, 0.0, 12.0 * pairs);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, counts, pairs, offsets,
                       reinterpret_cast<long long *>(n_events));
    if (timed) sedmi::profile_end(rec, s);
    SEDMI_CHECK_LAUNCH();
    if (timed) rec = sedmi::profile_begin("decode_kernel<true>", s, 0.0, dbytes);
    hipLaunchKernelGGL(decode_kernel<true>, dim3(grid), dim3(64 * kWavesPerBlock), 0, s, p, n_class, ss, sc,
                       static_cast<long>(n_seg), thr, apply_sigmoid, half, counts, offsets, ev_seg, ev_cls,
                       ev_on, ev_off, static_cast<long>(capacity));
    if (timed) sedmi::profile_end(rec, s);
    SEDMI_CHECK_LAUNCH();
    return SEDMI_OK;
}

int sedmi_decode_events(const float *p, int64_t total_frames, int n_class, const int64_t *seg_start,
                        const int64_t *seg_cutoff, int64_t n_seg, float thr, int apply_sigmoid,
                        int median_win, int32_t *ev_seg, int32_t *ev_cls, int32_t *ev_on, int32_t *ev_off,
                        int64_t capacity, int64_t *n_events, void *ws, size_t ws_bytes, void *stream) {
    const int rc = decode_check(p, total_frames, n_class, seg_start, seg_cutoff, n_seg, median_win, ev_seg, ev_cls,
                                ev_on, ev_off, capacity, n_events);
    if (rc != SEDMI_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n_seg == 0) {
        SEDMI_CHECK_HIP(hipMemsetAsync(n_events, 0, sizeof(int64_t), s));
        return SEDMI_OK;
    }
    SEDMI_REQUIRE(ws && ws_bytes >= sedmi_decode_workspace_bytes(n_seg, n_class),
                  "decode workspace too small: need %zu bytes", sedmi_decode_workspace_bytes(n_seg, n_class));
    return decode_launch(p, total_frames, n_class, seg_start, seg_cutoff, n_seg, thr, apply_sigmoid, median_win,
                         ev_seg, ev_cls, ev_on, ev_off, capacity, n_events, ws, true, s);
}

int sedmi_decode_events_static(const float *p, int64_t total_frames, int n_class, const int64_t *seg_start,
                               const int64_t *seg_cutoff, int64_t n_seg, float thr, int apply_sigmoid,
                               int median_win, int32_t *ev_seg, int32_t *ev_cls, int32_t *ev_on,
                               int32_t *ev_off, int64_t capacity, int64_t *n_events, void *stream) {
    const int rc = decode_check(p, total_frames, n_class, seg_start, seg_cutoff, n_seg, median_win, ev_seg, ev_cls,
                                ev_on, ev_off, capacity, n_events);
    if (rc != SEDMI_OK) return rc;
    SEDMI_REQUIRE(n_seg <= SEDMI_DECODE_STATIC_MAX_PAIRS / n_class,
                  "sedmi_decode_events_static: %lld recordings x %d classes exceeds the static scratch of %d "
                  "(recording, class) pairs; use sedmi_decode_events with a workspace",
                  static_cast<long long>(n_seg), n_class, SEDMI_DECODE_STATIC_MAX_PAIRS);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n_seg == 0) {
        SEDMI_CHECK_HIP(hipMemsetAsync(n_events, 0, sizeof(int64_t), s));
        return SEDMI_OK;
    }
    void *scratch_dev = nullptr;
    const int rs = static_scratch_address(&scratch_dev);
    if (rs != SEDMI_OK) return rs;
    // nothing in the scratch has to start at zero: the count pass writes counts[0 .. pairs) and the scan
    // writes offsets[0 .. pairs) before the write pass reads either
    return decode_launch(p, total_frames, n_class, seg_start, seg_cutoff, n_seg, thr, apply_sigmoid, median_win,
                         ev_seg, ev_cls, ev_on, ev_off, capacity, n_events, scratch_dev, false, s);
}

int sedmi_confusion_counts(const float *p, const float *labels, int64_t n_rows, int n_class, float thr,
                           int apply_sigmoid, int64_t *counts, void *stream) {
    SEDMI_REQUIRE(n_rows >= 0 && n_class > 0, "bad shape");
    if (n_rows == 0) return SEDMI_OK;
    SEDMI_REQUIRE(p && labels && counts, "null data pointer");
    const unsigned slabs = static_cast<unsigned>(std::min<int64_t>(sedmi::ceil_div(n_rows, 256), 256));
    hipLaunchKernelGGL(confusion_kernel, dim3(n_class, slabs), dim3(256), 0, static_cast<hipStream_t>(stream),
                       p, labels, static_cast<long>(n_rows), n_class, thr, apply_sigmoid,
                       reinterpret_cast<unsigned long long *>(counts));
    SEDMI_CHECK_LAUNCH();
    return SEDMI_OK;
}

}  // extern "C"
