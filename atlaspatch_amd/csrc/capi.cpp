// Library-level entry points of the C ABI: error text, device info, K1 wrappers.
#include <cstdarg>
#include <cstdio>
#include <cstdio>
#include <climits>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <zlib.h>
#include <cmath>
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

int ap_abi_version(void) { return AP_ABI_VERSION; }

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

int ap_host_format_passports(const int32_t* coords, int n, const char* prefix, const char* suffix, char* out, int width) {
    AP_REQUIRE((coords || n == 0) && prefix && suffix && out && n >= 0 && width > 0, "ap_host_format_passports: bad arguments");
    const size_t plen = strlen(prefix), slen = strlen(suffix);
    std::vector<char> line(plen + slen + 128);
    memcpy(line.data(), prefix, plen);
    for (int i = 0; i < n; ++i) {
        const int32_t* c = coords + (size_t)i * 5;
        char* p = line.data() + plen;
        // "__x{X}_y{Y}_rw{RW}_rh{RH}_lv{LV}": decimal int32 fields, '-' for negatives (Python's str(int))
        static const char* const tags[5] = {"__x", "_y", "_rw", "_rh", "_lv"};
        for (int f = 0; f < 5; ++f) {
            for (const char* t = tags[f]; *t; ++t) *p++ = *t;
            uint32_t v = (uint32_t)c[f];
            if (c[f] < 0) { *p++ = '-'; v = 0u - v; }
            char digits[12];
            int nd = 0;
            do { const uint32_t q = v / 10u; digits[nd++] = (char)('0' + (v - q * 10u)); v = q; } while (v);
            while (nd) *p++ = digits[--nd];
        }
        memcpy(p, suffix, slen);
        const size_t len = (size_t)(p - line.data()) + slen;
        char* o = out + (size_t)i * width;
        const size_t take = len < (size_t)width ? len : (size_t)width;      // NumPy's S<width> truncates longer strings
        memcpy(o, line.data(), take);
        memset(o + take, 0, (size_t)width - take);                           // ... and pads shorter ones with NUL
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

int ap_host_synth_tiles(void* dst, const int32_t* xy, int n, int side, int level_ds, int level, int64_t width,
                        int64_t height, uint32_t seed, const int64_t* ellipses, int k) {
    AP_REQUIRE(dst && (xy || n == 0) && (ellipses || k == 0) && n >= 0 && side > 0 && level_ds > 0 && k >= 0,
               "ap_host_synth_tiles: bad arguments");
    auto mix = [](uint32_t a) { a ^= a >> 16; a *= 0x7FEB352Du; a ^= a >> 15; a *= 0x846CA68Bu; a ^= a >> 16; return a; };
    uint8_t* o = (uint8_t*)dst;
    for (int t = 0; t < n; ++t) {
        for (int py = 0; py < side; ++py) {
            const long long gy = (long long)xy[2 * t + 1] + (long long)py * level_ds, uy = gy >> 4;
            long long last_ux = INT64_MIN;
            bool tissue = false;
            for (int px = 0; px < side; ++px, o += 3) {
                const long long gx = (long long)xy[2 * t] + (long long)px * level_ds, ux = gx >> 4;
                if (gx < 0 || gy < 0 || gx >= width || gy >= height) { o[0] = o[1] = o[2] = 0; continue; }
                if (ux != last_ux) {           // the ellipse test lives on a 16-pixel lattice: once per lattice cell
                    last_ux = ux;
                    tissue = false;
                    for (int e = 0; e < k && !tissue; ++e) {
                        const long long a = ellipses[4 * e + 2], b = ellipses[4 * e + 3];
                        const long long dx = (ux - ellipses[4 * e]) * b, dy = (uy - ellipses[4 * e + 1]) * a, ab = a * b;
                        tissue = dx * dx + dy * dy <= ab * ab;
                    }
                }
                const uint32_t h = mix((uint32_t)gx * 0x9E3779B1u + (uint32_t)gy * 0x85EBCA77u + seed + (uint32_t)level * 0xC2B2AE3Du);
                const int n0 = h & 0xFF, n1 = (h >> 8) & 0xFF, n2 = (h >> 16) & 0xFF, bg = 236 + (n0 & 7);
                o[0] = (uint8_t)(tissue ? 168 + (n0 >> 2) : bg);
                o[1] = (uint8_t)(tissue ? 72 + (n1 >> 1) : bg);
                o[2] = (uint8_t)(tissue ? 136 + (n2 >> 2) : bg);
            }
        }
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
    AP_REQUIRE(epilogue == AP_EPI_BIAS || epilogue == AP_EPI_BIAS_GELU || epilogue == AP_EPI_BIAS_RESID || epilogue == AP_EPI_BIAS_QUICK_GELU,
               "ap_gemm: unknown epilogue %d", epilogue);
    AP_REQUIRE(dtype == AP_F16 || dtype == AP_BF16 || dtype == AP_F32, "ap_gemm: unknown dtype %d", dtype);
    ap::GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.gamma = gamma; g.out = out; g.ldo = ldo;
    if (impl == 129) {                       // float32 buffers, split-f16 products: W = the rows ap_split_f16_weights made
        AP_REQUIRE(dtype == AP_F32, "ap_gemm: impl 129 (split-f16 products) takes float32 buffers");
        g.split = 1;
        impl = 128;
    }
    return ap::launch_gemm_impl(dtype, epilogue, g, impl, variant, (hipStream_t)stream);
}

int ap_split_f16_weights(const float* w32, void* out, size_t count, ap_stream_t stream) {
    return ap::launch_split_f16_weights(w32, out, count, (hipStream_t)stream);
}

static int gemm_split_f16(const float* A, int lda, const void* w_split, int M, int N, int K, const float* bias, int act,
                          const float* resid, int ldr, float* out, int ldo, int win_mode, int b, int h, int w, int ws, ap_stream_t stream) {
    AP_REQUIRE(A && w_split && out, "ap_gemm_split_f16: null pointer");
    AP_REQUIRE(act == 0 || act == 1, "ap_gemm_split_f16: act %d (0 none, 1 GELU)", act);
    AP_REQUIRE(M > 0 && N > 0 && N % 32 == 0 && K > 0 && K % 32 == 0, "ap_gemm_split_f16: %d x %d x %d (N and K multiples of 32)", M, N, K);
    AP_REQUIRE(lda >= K && lda % 4 == 0 && ldo >= N && ldo % 4 == 0 && (!resid || (ldr >= N && ldr % 4 == 0)), "ap_gemm_split_f16: strides");
    AP_REQUIRE(((uintptr_t)A | (uintptr_t)w_split | (uintptr_t)out | (uintptr_t)resid | (uintptr_t)bias) % 16 == 0, "ap_gemm_split_f16: 16-byte aligned pointers");
    ap::GemmArgs g{};
    g.A = A; g.lda = lda; g.W = w_split; g.ldw = K; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.out = out; g.ldo = ldo; g.resid = resid; g.ldr = ldr; g.split = 1;
    if (win_mode != 0) {
        AP_REQUIRE((win_mode == 1 || win_mode == 2) && b > 0 && h > 0 && w > 0 && ws > 0, "ap_gemm_split_f16_windows: mode %d, %d x %d x %d, window %d", win_mode, b, h, w, ws);
        g.win_mode = win_mode; g.win_ws = ws; g.win_H = h; g.win_W = w; g.win_nwy = (h + ws - 1) / ws; g.win_nwx = (w + ws - 1) / ws;
        AP_REQUIRE((long)b * g.win_nwy * g.win_nwx * ws * ws == (long)M, "ap_gemm_split_f16_windows: M = %d is not %d images x %d x %d windows of %d x %d", M, b,
                   g.win_nwy, g.win_nwx, ws, ws);
        if (win_mode == 1) {
            // padding rows of the gathered operand: one row of zeros per device, allocated on first use (outside any stream capture: the
            // predictor's warm-up forward comes first)
            constexpr int kZeroFloats = 16384;
            static float* zero_rows[64] = {};
            int dev = 0;
            AP_HIP_CHECK(hipGetDevice(&dev));
            AP_REQUIRE(dev >= 0 && dev < 64 && K <= kZeroFloats, "ap_gemm_split_f16_windows: device %d / K %d", dev, K);
            if (!zero_rows[dev]) {
                AP_HIP_CHECK(hipMalloc((void**)&zero_rows[dev], kZeroFloats * sizeof(float)));
                AP_HIP_CHECK(hipMemsetAsync(zero_rows[dev], 0, kZeroFloats * sizeof(float), (hipStream_t)stream));   // ordered before this launch
            }
            g.zero_row = zero_rows[dev];
        }
    }
    return ap::launch_gemm_impl(AP_F32, act == 1 ? ap::EPI_BIAS_GELU : ap::EPI_BIAS_STORE, g, 128, 0, (hipStream_t)stream);
}

int ap_gemm_split_f16(const float* A, int lda, const void* w_split, int M, int N, int K, const float* bias, int act,
                      const float* resid, int ldr, float* out, int ldo, ap_stream_t stream) {
    return gemm_split_f16(A, lda, w_split, M, N, K, bias, act, resid, ldr, out, ldo, 0, 0, 0, 0, 0, stream);
}

int ap_gemm_split_f16_windows(const float* A, int lda, const void* w_split, int M, int N, int K, const float* bias, int act,
                              const float* resid, int ldr, float* out, int ldo, int win_mode, int b, int h, int w, int ws,
                              ap_stream_t stream) {
    return gemm_split_f16(A, lda, w_split, M, N, K, bias, act, resid, ldr, out, ldo, win_mode, b, h, w, ws, stream);
}

int ap_gemm_fused(int dtype, int epilogue, const void* A, int lda, const void* W, int ldw, int M, int N, int K,
                  const float* bias, const float* colsum, const float* rowstats, float* partial, void* out, int ldo,
                  int impl, ap_stream_t stream) {
    AP_REQUIRE(A && W && bias && out, "ap_gemm_fused: null pointer");
    AP_REQUIRE(dtype == AP_F16 || dtype == AP_BF16, "ap_gemm_fused: f16 / bf16 only");
    int epi;
    if (epilogue == AP_EPI_NORM) epi = ap::EPI_NORM_STORE;
    else if (epilogue == AP_EPI_NORM_GELU) epi = ap::EPI_NORM_GELU;
    else if (epilogue == AP_EPI_NORM_SWIGLU) epi = ap::EPI_NORM_SWIGLU;
    else if (epilogue == AP_EPI_NORM_QUICK_GELU) epi = ap::EPI_NORM_QGELU;
    else if (epilogue == AP_EPI_RESID_STATS) epi = ap::EPI_RESID_STATS;
    else { ap::set_error("ap_gemm_fused: unknown epilogue %d", epilogue); return AP_ERR_INVALID; }
    AP_REQUIRE(epi == ap::EPI_RESID_STATS ? partial != nullptr : (colsum && rowstats), "ap_gemm_fused: missing operand for epilogue %d", epilogue);
    ap::GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.colsum = colsum; g.rowstats = rowstats; g.partial = partial; g.out = out; g.ldo = ldo;
    const int variant = (int)((unsigned)impl >> 12);      // bits 12..: kernel tuning variant (tools only), low 12 bits: implementation
    impl &= 0xfff;
    AP_REQUIRE(impl == 0 || impl == 128 || impl == 256 || impl == 257, "ap_gemm_fused: impl %d (0 = pick, 128, 256; 257 = the A/B twin)", impl);
    AP_REQUIRE(M > 0 && (impl == 128 || ap::gemm256_supports(dtype, epi, g)), "ap_gemm_fused: unsupported problem (N %% 256, K %% 128, 16-byte strides)");
    return ap::launch_gemm_impl(dtype, epi, g, impl, variant, (hipStream_t)stream);
}

int ap_stream_init(int dtype, const float* tok, int rows, int dim, float eps, void* x, float* rowstats, ap_stream_t stream) {
    AP_REQUIRE(tok && x && rowstats, "ap_stream_init: null pointer");
    return ap::launch_stream_init(dtype, tok, rows, dim, eps, x, rowstats, (hipStream_t)stream);
}

int ap_rowstats_finalize(const float* partial, int rows, int groups, int dim, float eps, float* rowstats, ap_stream_t stream) {
    AP_REQUIRE(partial && rowstats && groups > 0 && dim > 0, "ap_rowstats_finalize: bad arguments");
    return ap::launch_rowstats_finalize(partial, rows, groups, dim, eps, rowstats, (hipStream_t)stream);
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
    AP_REQUIRE(head_dim > 0, "ap_attention: head_dim %d", head_dim);
    return ap::launch_attention(dtype, qkv, out, n, tokens, heads, head_dim, 1.0f / sqrtf((float)head_dim), (hipStream_t)stream);
}

}  // extern "C"
