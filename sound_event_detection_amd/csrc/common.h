// Shared host-side helpers of libsedmi: error reporting and small utilities.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/sedmi.h"

namespace sedmi {

// Thread-local message returned by sedmi_last_error().
char *error_buffer();
int set_error(int code, const char *fmt, ...);

#define SEDMI_CHECK_HIP(expr)                                                              \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return ::sedmi::set_error(SEDMI_EHIP, "%s failed: %s (%s:%d)", #expr,          \
                                      hipGetErrorString(e_), __FILE__, __LINE__);          \
    } while (0)

#define SEDMI_REQUIRE(cond, ...)                                                           \
    do {                                                                                   \
        if (!(cond)) return ::sedmi::set_error(SEDMI_EINVAL, __VA_ARGS__);                 \
    } while (0)

// Launch-error check that does not synchronise.
#define SEDMI_CHECK_LAUNCH() SEDMI_CHECK_HIP(hipGetLastError())

template <typename T>
int upload(T **dev, const std::vector<T> &host) {
    *dev = nullptr;
    size_t bytes = host.size() * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    SEDMI_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(dev), bytes));
    if (!host.empty())
        SEDMI_CHECK_HIP(hipMemcpy(*dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return SEDMI_OK;
}

// Optional per-launch timing (see sedmi_profile_enable in sedmi.h).
bool profile_enabled();
long profile_begin(const char *name, hipStream_t stream, double flops, double bytes);   // record index or -1
void profile_end(long index, hipStream_t stream);

struct ProfileScope {
    hipStream_t stream;
    long index;            // this scope's record; -1: profiling off or the events could not be created
    ProfileScope(const char *name, hipStream_t s, double flops, double bytes)
        : stream(s), index(profile_enabled() ? profile_begin(name, s, flops, bytes) : -1) {}
    ~ProfileScope() {
        if (index >= 0) profile_end(index, stream);
    }
};

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace sedmi
