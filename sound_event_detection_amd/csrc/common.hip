// libsedmi: error reporting, version, device probe.
#include "common.h"

#include <map>
#include <mutex>
#include <utility>

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

int set_max_dynamic_lds(const void *kernel, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, size_t> granted;      // (kernel, device) -> ceiling already set
    int device = 0;
    SEDMI_CHECK_HIP(hipGetDevice(&device));
    std::lock_guard<std::mutex> lock(mu);
    size_t &have = granted[{kernel, device}];
    if (have >= bytes && have > 0) return SEDMI_OK;
    SEDMI_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes)));
    have = bytes;
    return SEDMI_OK;
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
unsigned long g_prof_generation = 0;      // bumped whenever g_prof_records is cleared
}  // namespace

bool profile_enabled() { return g_prof_on; }

// Returns the ticket of the new record; index -1 if its events could not be created (nothing is recorded
// then, and a half-created pair is destroyed).
ProfileTicket profile_begin(const char *name, hipStream_t stream, double flops, double bytes) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    ProfileRecord r{name, flops, bytes, nullptr, nullptr};
    if (hipEventCreate(&r.start) != hipSuccess) return {-1, g_prof_generation};
    if (hipEventCreate(&r.stop) != hipSuccess) {
        (void)hipEventDestroy(r.start);
        return {-1, g_prof_generation};
    }
    (void)hipEventRecord(r.start, stream);
    g_prof_records.push_back(r);
    return {static_cast<long>(g_prof_records.size()) - 1, g_prof_generation};
}

// Records the stop event of the ticket's record.  A ticket whose record a collect / enable has dropped in
// between carries an older generation and is ignored, so that it can never stop a record pushed afterwards
// that happens to sit at the same index.
void profile_end(ProfileTicket ticket, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    if (ticket.index >= 0 && ticket.generation == g_prof_generation &&
        static_cast<size_t>(ticket.index) < g_prof_records.size())
        (void)hipEventRecord(g_prof_records[static_cast<size_t>(ticket.index)].stop, stream);
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
    ++sedmi::g_prof_generation;
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
    ++sedmi::g_prof_generation;
    return n;
}

int sedmi_version(void) { return 300; }

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
