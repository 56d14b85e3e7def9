// Library-level entry points of the C ABI: error text, device info, K1 wrappers.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include "ap_common.h"

namespace ap {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap_;
    va_start(ap_, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap_);
    va_end(ap_);
}

}  // namespace ap

extern "C" {

int ap_abi_version(void) { return 1; }

const char* ap_last_error(void) { return ap::g_error; }

int ap_device_info(int device, char* name, int name_cap, int* cu_count, size_t* hbm_bytes) {
    hipDeviceProp_t prop;
    AP_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (name && name_cap > 0) {
        snprintf(name, (size_t)name_cap, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    return AP_OK;
}

int ap_preproc_u8hwc_to_chw(const uint8_t* src, int n, int h, int w, int crop_top, int crop_left,
                            int oh, int ow, const float mean[3], const float stdv[3], void* dst,
                            int dst_dtype, ap_stream_t stream) {
    return ap::preproc_chw(src, n, h, w, crop_top, crop_left, oh, ow, mean, stdv, dst, dst_dtype,
                           (hipStream_t)stream);
}

int ap_preproc_u8hwc_to_patchrows(const uint8_t* src, int n, int h, int w, int crop_top,
                                  int crop_left, int oh, int ow, int ps, const float mean[3],
                                  const float stdv[3], void* dst, int ld, int dst_dtype,
                                  ap_stream_t stream) {
    return ap::preproc_patchrows(src, n, h, w, crop_top, crop_left, oh, ow, ps, mean, stdv, dst, ld,
                                 dst_dtype, (hipStream_t)stream);
}

}  // extern "C"
