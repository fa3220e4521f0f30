// libsedmi: error reporting, version, device probe.
#include "common.h"

#include <mutex>

namespace sedmi {

char *error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

int set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

namespace {
struct ProfileRecord {
    const char *name;
    double flops, bytes;
    hipEvent_t start, stop;
};
std::mutex g_prof_mutex;
bool g_prof_on = false;
std::vector<ProfileRecord> g_prof_records;
}  // namespace

bool profile_enabled() { return g_prof_on; }

void profile_begin(const char *name, hipStream_t stream, double flops, double bytes) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    ProfileRecord r{name, flops, bytes, nullptr, nullptr};
    if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return;
    (void)hipEventRecord(r.start, stream);
    g_prof_records.push_back(r);
}

void profile_end(hipStream_t stream) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    if (!g_prof_records.empty()) (void)hipEventRecord(g_prof_records.back().stop, stream);
}

}  // namespace sedmi

extern "C" {

int sedmi_profile_enable(int on) {
    std::lock_guard<std::mutex> lock(sedmi::g_prof_mutex);
    for (auto &r : sedmi::g_prof_records) {
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    sedmi::g_prof_records.clear();
    sedmi::g_prof_on = on != 0;
    return SEDMI_OK;
}

int sedmi_profile_collect(sedmi_kernel_stat *out, int capacity) {
    SEDMI_REQUIRE(out != nullptr && capacity >= 0, "bad argument");
    std::lock_guard<std::mutex> lock(sedmi::g_prof_mutex);
    int n = 0;
    for (auto &r : sedmi::g_prof_records) {
        float ms = 0.f;
        if (hipEventSynchronize(r.stop) == hipSuccess) (void)hipEventElapsedTime(&ms, r.start, r.stop);
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
        int k = 0;
        while (k < n && std::strncmp(out[k].name, r.name, sizeof(out[k].name) - 1) != 0) ++k;
        if (k == n) {
            if (n == capacity) continue;
            std::memset(&out[n], 0, sizeof(out[n]));
            std::strncpy(out[n].name, r.name, sizeof(out[n].name) - 1);
            ++n;
        }
        out[k].launches += 1;
        out[k].ms += ms;
        out[k].flops += r.flops;
        out[k].bytes += r.bytes;
    }
    sedmi::g_prof_records.clear();
    return n;
}

int sedmi_version(void) { return 100; }

const char *sedmi_last_error(void) { return sedmi::error_buffer(); }

int sedmi_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return sedmi::set_error(SEDMI_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    return n;
}

}  // extern "C"
