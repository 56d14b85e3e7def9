// Library-level entry points of the C ABI: error text, device info, K1 wrappers.
#include <cstdarg>
#include <cstdio>
#include <cstdio>
#include <cstring>
#include <vector>
#include <zlib.h>
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

int ap_abi_version(void) { return 8; }

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

int ap_host_gather_tiles(void* dst, const void* const* src, int n, size_t bytes_each) {
    AP_REQUIRE(dst && (src || n == 0) && n >= 0, "ap_host_gather_tiles: bad arguments");
    char* d = (char*)dst;
    for (int i = 0; i < n; ++i) {
        AP_REQUIRE(src[i], "ap_host_gather_tiles: null tile %d", i);
        memcpy(d + (size_t)i * bytes_each, src[i], bytes_each);
    }
    return AP_OK;
}

int ap_host_inflate_tiles(void* dst, const char* const* paths, int n, size_t bytes_each) {
    AP_REQUIRE(dst && (paths || n == 0) && n >= 0 && bytes_each > 0, "ap_host_inflate_tiles: bad arguments");
    std::vector<unsigned char> buf;
    for (int i = 0; i < n; ++i) {
        AP_REQUIRE(paths[i], "ap_host_inflate_tiles: null path %d", i);
        FILE* f = fopen(paths[i], "rb");
        AP_REQUIRE(f, "ap_host_inflate_tiles: cannot open %s", paths[i]);
        fseek(f, 0, SEEK_END);
        const long size = ftell(f);
        fseek(f, 0, SEEK_SET);
        buf.resize(size > 0 ? (size_t)size : 1);
        const size_t got = size > 0 ? fread(buf.data(), 1, (size_t)size, f) : 0;
        fclose(f);
        AP_REQUIRE(size > 0 && got == (size_t)size, "ap_host_inflate_tiles: short read of %s", paths[i]);
        uLongf len = (uLongf)bytes_each;
        const int zr = uncompress((Bytef*)dst + (size_t)i * bytes_each, &len, buf.data(), (uLong)size);
        AP_REQUIRE(zr == Z_OK && len == bytes_each, "ap_host_inflate_tiles: %s does not inflate to %zu bytes (zlib %d)",
                   paths[i], bytes_each, zr);
    }
    return AP_OK;
}

int ap_tile_content_counts(const uint8_t* tiles, int n, int h, int w, int black_thresh, int white_sat_thresh,
                           int white_value_thresh, uint32_t* counts, ap_stream_t stream) {
    return ap::tile_content_counts(tiles, n, h, w, black_thresh, white_sat_thresh, white_value_thresh,
                                   (unsigned*)counts, (hipStream_t)stream);
}

int ap_gemm(int dtype, int epilogue, const void* A, int lda, const void* W, int ldw, int M, int N,
            int K, const float* bias, const float* gamma, void* out, int ldo, int impl, int variant,
            ap_stream_t stream) {
    AP_REQUIRE(A && W && bias && out, "ap_gemm: null pointer");
    AP_REQUIRE(epilogue == AP_EPI_BIAS || epilogue == AP_EPI_BIAS_GELU || epilogue == AP_EPI_BIAS_RESID,
               "ap_gemm: unknown epilogue %d", epilogue);
    AP_REQUIRE(dtype == AP_F16 || dtype == AP_BF16 || dtype == AP_F32, "ap_gemm: unknown dtype %d", dtype);
    ap::GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.gamma = gamma; g.out = out; g.ldo = ldo;
    return ap::launch_gemm_impl(dtype, epilogue, g, impl, variant, (hipStream_t)stream);
}

int ap_gemm_trace(long long* device_buf, int tiles_per_workgroup) {
    ap::set_gemm_trace(device_buf, device_buf ? tiles_per_workgroup : 0);
    return AP_OK;
}

int ap_layernorm(int out_dtype, const float* x, long stride, int rows, int dim, const float* gamma,
                 const float* beta, float eps, void* out, ap_stream_t stream) {
    AP_REQUIRE(x && gamma && beta && out, "ap_layernorm: null pointer");
    return ap::launch_layernorm(out_dtype, x, stride, rows, dim, gamma, beta, eps, out, (hipStream_t)stream);
}

int ap_attention(int dtype, const void* qkv, void* out, int n, int tokens, int heads, int head_dim,
                 ap_stream_t stream) {
    AP_REQUIRE(qkv && out, "ap_attention: null pointer");
    return ap::launch_attention(dtype, qkv, out, n, tokens, heads, head_dim, (hipStream_t)stream);
}

}  // extern "C"
