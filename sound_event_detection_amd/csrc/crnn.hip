// placeholder while the CRNN kernels are being written
#include "common.h"
struct sedmi_crnn { int x; };
extern "C" {
int sedmi_crnn_create(sedmi_crnn **, const sedmi_crnn_weights *) { return sedmi::set_error(SEDMI_EINVAL, "CRNN not built yet"); }
int sedmi_crnn_destroy(sedmi_crnn *) { return SEDMI_OK; }
int64_t sedmi_crnn_out_frames(const sedmi_crnn *, int64_t) { return -1; }
size_t sedmi_crnn_workspace_bytes(const sedmi_crnn *, int64_t, int64_t) { return 0; }
int sedmi_crnn_forward_f32(const sedmi_crnn *, const float *, int64_t, int64_t, float *, void *, size_t, void *) { return SEDMI_EINVAL; }
int sedmi_crnn_cnn_f32(const sedmi_crnn *, const float *, int64_t, int64_t, float *, void *, size_t, void *) { return SEDMI_EINVAL; }
int sedmi_crnn_gru_f32(const sedmi_crnn *, const float *, int64_t, int64_t, float *, void *, size_t, void *) { return SEDMI_EINVAL; }
int sedmi_crnn_head_f32(const sedmi_crnn *, const float *, int64_t, float *, void *) { return SEDMI_EINVAL; }
}
