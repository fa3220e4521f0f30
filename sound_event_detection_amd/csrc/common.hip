// libsedmi: error reporting, version, device probe.
#include "common.h"

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

}  // namespace sedmi

extern "C" {

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
