// Pillow-exact separable resampling of uint8 RGB tiles on the device (Image.resize with BICUBIC / BILINEAR):
// what timm's / open_clip's Resize does to every tile before ToTensor for uni_v1 (224) and conch_v1 (448)
// (models/patch/uni.py:48-49, conch.py:35-38 -> PIL), so those encoders can take tiles straight from the ring.
//
// Pillow (libImaging/Resample.c, 8 bits per channel): per output position a window [xmin, xmin + n) of input
// pixels and n fixed-point weights (22 fractional bits, computed in double by the host exactly as
// precompute_coeffs / normalize_coeffs_8bpc do); out = clip8((2^21 + sum pixel * weight) >> 22) in int32.
// Horizontal pass first into a uint8 intermediate, then the vertical pass -- the same two roundings.
//
// One thread per output pixel (3 channels); the tables are tiny and L1-resident.  HBM-bound:
// reads n*h*w*3 + n*h*ow*3, writes n*h*ow*3 + n*oh*ow*3 bytes.
#include "ap_common.h"

namespace ap {
namespace {

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= 22;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// src [n, h, w, 3] -> dst [n, h, ow, 3]
__global__ void resample_h_kernel(const uint8_t* __restrict__ src, int n, int h, int w, int ow,
                                  const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                  uint8_t* __restrict__ dst) {
    const size_t total = (size_t)n * h * ow;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int xx = (int)(i % ow);
    const size_t row = i / ow;                                  // (image, y)
    const int xmin = bounds[xx * 2], cnt = bounds[xx * 2 + 1];
    const int* k = kk + xx * ksize;
    const uint8_t* p = src + (row * w + xmin) * 3;
    int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
    for (int x = 0; x < cnt; ++x) {
        const int c = k[x];
        s0 += (int)p[x * 3] * c; s1 += (int)p[x * 3 + 1] * c; s2 += (int)p[x * 3 + 2] * c;
    }
    uint8_t* d = dst + i * 3;
    d[0] = clip8(s0); d[1] = clip8(s1); d[2] = clip8(s2);
}

// src [n, h, ow, 3] -> dst [n, oh, ow, 3]
__global__ void resample_v_kernel(const uint8_t* __restrict__ src, int n, int h, int ow, int oh,
                                  const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                  uint8_t* __restrict__ dst) {
    const size_t total = (size_t)n * oh * ow;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int xx = (int)(i % ow);
    const size_t r = i / ow;
    const int yy = (int)(r % oh);
    const size_t img = r / oh;
    const int ymin = bounds[yy * 2], cnt = bounds[yy * 2 + 1];
    const int* k = kk + yy * ksize;
    const uint8_t* p = src + ((img * h + ymin) * ow + xx) * 3;
    const size_t stride = (size_t)ow * 3;
    int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
    for (int y = 0; y < cnt; ++y) {
        const int c = k[y];
        s0 += (int)p[y * stride] * c; s1 += (int)p[y * stride + 1] * c; s2 += (int)p[y * stride + 2] * c;
    }
    uint8_t* d = dst + i * 3;
    d[0] = clip8(s0); d[1] = clip8(s1); d[2] = clip8(s2);
}

}  // namespace
}  // namespace ap

extern "C" int ap_resample_u8(const uint8_t* src, int n, int h, int w, uint8_t* dst, int oh, int ow,
                              const int32_t* bounds_x, const int32_t* coeffs_x, int ksize_x,
                              const int32_t* bounds_y, const int32_t* coeffs_y, int ksize_y,
                              uint8_t* tmp, ap_stream_t stream) {
    AP_REQUIRE(src && dst && tmp && bounds_x && coeffs_x && bounds_y && coeffs_y, "ap_resample_u8: null pointer");
    AP_REQUIRE(n >= 0 && h > 0 && w > 0 && oh > 0 && ow > 0 && ksize_x > 0 && ksize_y > 0, "ap_resample_u8: bad shape");
    if (n == 0) return AP_OK;
    hipStream_t s = (hipStream_t)stream;
    const size_t t1 = (size_t)n * h * ow, t2 = (size_t)n * oh * ow;
    ap::resample_h_kernel<<<(unsigned)((t1 + 255) / 256), 256, 0, s>>>(src, n, h, w, ow, bounds_x, coeffs_x, ksize_x, tmp);
    ap::resample_v_kernel<<<(unsigned)((t2 + 255) / 256), 256, 0, s>>>(tmp, n, h, ow, oh, bounds_y, coeffs_y, ksize_y, dst);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}
