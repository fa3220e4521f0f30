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
struct ProfileTicket {
    long index;                 // position of the record; -1: none
    unsigned long generation;   // of the record list at the time the record was pushed
};
ProfileTicket profile_begin(const char *name, hipStream_t stream, double flops, double bytes);
void profile_end(ProfileTicket ticket, hipStream_t stream);

struct ProfileScope {
    hipStream_t stream;
    ProfileTicket ticket;  // this scope's record; index -1: profiling off or the events could not be created
    ProfileScope(const char *name, hipStream_t s, double flops, double bytes)
        : stream(s), ticket(profile_enabled() ? profile_begin(name, s, flops, bytes) : ProfileTicket{-1, 0}) {}
    ~ProfileScope() {
        if (ticket.index >= 0) profile_end(ticket, stream);
    }
};

// Raises a kernel's dynamic-LDS ceiling (hipFuncAttributeMaxDynamicSharedMemorySize) to at least `bytes` on the
// current device.  The attribute belongs to the loaded function, not to a launch: the runtime is asked once per
// (kernel, device) and size, later calls are a table look-up.
int set_max_dynamic_lds(const void *kernel, size_t bytes);

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace sedmi
