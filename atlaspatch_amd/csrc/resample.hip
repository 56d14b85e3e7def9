// Pillow-exact separable resampling of uint8 RGB tiles on the device (Image.resize with BICUBIC / BILINEAR):
// what timm's / open_clip's Resize does to every tile before ToTensor for uni_v1 (224) and conch_v1 (448)
// (models/patch/uni.py:48-49, conch.py:35-38 -> PIL), so those encoders can take tiles straight from the ring.
//
// Pillow (libImaging/Resample.c, 8 bits per channel): per output position a window [xmin, xmin + n) of input
// pixels and n fixed-point weights (22 fractional bits, computed in double by the host exactly as
// precompute_coeffs / normalize_coeffs_8bpc do); out = clip8((2^21 + sum pixel * weight) >> 22) in int32.
// Horizontal pass first into a uint8 intermediate, then the vertical pass -- the same two roundings.
//
// Product path = resample_fused_kernel: one workgroup per (image, band of output rows).  The input rows the band
// needs are one contiguous byte range of the image: staged into LDS with 16-byte loads; horizontal pass LDS -> LDS
// (uint8 intermediate, Pillow's first rounding); vertical pass LDS -> HBM four bytes of a row per thread (the vertical
// filter does not care about channels), whole dwords stored.  HBM traffic = the input read once + the output written
// once (n*h*w*3 + n*oh*ow*3 bytes); the intermediate never leaves the CU.  Shapes it does not take (row bytes not a
// multiple of 16 / 4, bands that do not fit LDS) run the two one-thread-per-pixel kernels below through `tmp`.
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

// src [n, h, w, 3] -> dst [n, oh, ow, 3]; grid (bands, n); dynamic LDS = rin_cap * (w * 3 + ow * 3) bytes
__global__ __launch_bounds__(256)
void resample_fused_kernel(const uint8_t* __restrict__ src, int h, int w, int oh, int ow, int band_rows, int rin_cap,
                           const int* __restrict__ bounds_x, const int* __restrict__ kx, int ksx,
                           const int* __restrict__ bounds_y, const int* __restrict__ ky, int ksy,
                           uint8_t* __restrict__ dst) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x;
    const int oy0 = blockIdx.x * band_rows;
    const int rows = min(band_rows, oh - oy0);
    const int last = oy0 + rows - 1;
    const int iy0 = bounds_y[oy0 * 2];
    int rin = bounds_y[last * 2] + bounds_y[last * 2 + 1] - iy0;      // windows are monotone in the output row
    rin = rin < rin_cap ? rin : rin_cap;                              // (cannot exceed the host's bound)
    const int wb = w * 3, owb = ow * 3;
    uint8_t* in = lds;                        // [rin][wb]
    uint8_t* tmp = lds + (size_t)rin_cap * wb;        // [rin][owb]
    const uint8_t* img = src + (size_t)blockIdx.y * h * wb;

    // ---- stage: rows iy0 .. iy0 + rin are contiguous in the image
    {
        const uint4* g = (const uint4*)(img + (size_t)iy0 * wb);
        uint4* l = (uint4*)in;
        const int chunks = rin * wb / 16;
        for (int i = tid; i < chunks; i += 256) l[i] = g[i];
    }
    __syncthreads();
    // ---- horizontal pass: a thread keeps its output column (window start, taps) and walks the staged rows
    for (int xx = tid; xx < ow; xx += 256) {
        const int xmin = bounds_x[xx * 2], cnt = bounds_x[xx * 2 + 1];
        const int* k = kx + xx * ksx;
        for (int r = 0; r < rin; ++r) {
            // bytes are taken out of ALIGNED dword reads: hipcc merges adjacent uint8 LDS reads into 16-bit reads
            // at odd addresses, which return the wrong bytes
            const uint32_t* row32 = (const uint32_t*)(in + r * wb);
            const int o = xmin * 3;
            int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
            for (int x = 0; x < cnt; ++x) {
                const int c = k[x];
                const int b = o + x * 3;
                s0 += (int)((row32[b >> 2] >> ((b & 3) * 8)) & 255u) * c;
                s1 += (int)((row32[(b + 1) >> 2] >> (((b + 1) & 3) * 8)) & 255u) * c;
                s2 += (int)((row32[(b + 2) >> 2] >> (((b + 2) & 3) * 8)) & 255u) * c;
            }
            uint8_t* d = tmp + r * owb + xx * 3;
            d[0] = clip8(s0); d[1] = clip8(s1); d[2] = clip8(s2);
        }
    }
    __syncthreads();
    // ---- vertical pass: four bytes of an output row per thread
    const int dw = owb / 4;
    const uint32_t* t32 = (const uint32_t*)tmp;
    uint32_t* out = (uint32_t*)(dst + ((size_t)blockIdx.y * oh + oy0) * owb);
    for (int i = tid; i < rows * dw; i += 256) {
        const int ry = i / dw, d = i - ry * dw;
        const int yy = oy0 + ry;
        const int ymin = bounds_y[yy * 2], cnt = bounds_y[yy * 2 + 1];
        const int* k = ky + yy * ksy;
        const uint32_t* p = t32 + (ymin - iy0) * dw + d;
        int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21, s3 = 1 << 21;
        for (int y = 0; y < cnt; ++y) {
            const int c = k[y];
            const uint32_t v = p[y * dw];
            s0 += (int)(v & 255u) * c; s1 += (int)((v >> 8) & 255u) * c;
            s2 += (int)((v >> 16) & 255u) * c; s3 += (int)(v >> 24) * c;
        }
        out[i] = (uint32_t)clip8(s0) | ((uint32_t)clip8(s1) << 8) | ((uint32_t)clip8(s2) << 16) | ((uint32_t)clip8(s3) << 24);
    }
}

// Image.reduce((fx, fy), box) of Pillow (libImaging/Reduce.c) for uint8 RGB: out[oy][ox] = box average of the fx x fy block
// (partial blocks at the right / bottom edge of the box are averaged over the pixels they have), computed as
// ((sum + n / 2) * mult(n)) >> 24 in uint32 arithmetic with mult(n) = (uint32)(2^32 / (256 n)) evaluated in float32.
// One thread per output pixel; a wave reads fx * 64 consecutive pixels of each of its fy rows (HBM-bound: every source
// byte is read once).
__global__ __launch_bounds__(256)
void pillow_reduce_kernel(const uint8_t* __restrict__ src, int w, int x0, int y0, int bw, int bh, int fx, int fy,
                          int ow, int oh, uint8_t* __restrict__ dst) {
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ox >= ow || oy >= oh) return;
    const int xs = ox * fx, ys = oy * fy;
    const int nx = min(fx, bw - xs), ny = min(fy, bh - ys);
    uint32_t s0 = 0, s1 = 0, s2 = 0;
    for (int y = 0; y < ny; ++y) {
        const uint8_t* p = src + ((size_t)(y0 + ys + y) * w + (x0 + xs)) * 3;
        for (int x = 0; x < nx; ++x) { s0 += p[3 * x]; s1 += p[3 * x + 1]; s2 += p[3 * x + 2]; }
    }
    const uint32_t n = (uint32_t)(nx * ny);
    const uint32_t mult = (uint32_t)(4294967296.0f / (float)(256u * n));
    const uint32_t amend = n / 2;
    uint8_t* d = dst + ((size_t)oy * ow + ox) * 3;
    d[0] = (uint8_t)(((s0 + amend) * mult) >> 24);
    d[1] = (uint8_t)(((s1 + amend) * mult) >> 24);
    d[2] = (uint8_t)(((s2 + amend) * mult) >> 24);
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
    // fused LDS path: largest band (multiple of 8 output rows) whose input rows + intermediate fit 52 KiB (three workgroups / CU)
    // (16-byte loads of the staged rows, dword stores: only for pointers aligned accordingly -- the C ABI takes any)
    if ((w * 3) % 16 == 0 && (ow * 3) % 4 == 0 && n <= 65535 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 3) == 0) {
        int band = 0, cap = 0;
        for (int b = 64; b >= 8; b -= 8) {
            const int c = (int)(((long long)(b - 1) * h) / oh) + ksize_y + 2;
            if ((size_t)c * (size_t)(w * 3 + ow * 3) <= 52 * 1024) { band = b; cap = c; break; }
        }
        if (band > 0) {
            const size_t lds = (size_t)cap * (size_t)(w * 3 + ow * 3);
            dim3 grid((unsigned)((oh + band - 1) / band), (unsigned)n);
            ap::resample_fused_kernel<<<grid, 256, lds, s>>>(src, h, w, oh, ow, band, cap, bounds_x, coeffs_x, ksize_x,
                                                            bounds_y, coeffs_y, ksize_y, dst);
            AP_HIP_CHECK(hipGetLastError());
            return AP_OK;
        }
    }
    const size_t t1 = (size_t)n * h * ow, t2 = (size_t)n * oh * ow;
    ap::resample_h_kernel<<<(unsigned)((t1 + 255) / 256), 256, 0, s>>>(src, n, h, w, ow, bounds_x, coeffs_x, ksize_x, tmp);
    ap::resample_v_kernel<<<(unsigned)((t2 + 255) / 256), 256, 0, s>>>(tmp, n, h, ow, oh, bounds_y, coeffs_y, ksize_y, dst);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

extern "C" int ap_pillow_reduce_u8(const uint8_t* src, int h, int w, int box_x, int box_y, int box_w, int box_h, int fx, int fy,
                                   uint8_t* dst, ap_stream_t stream) {
    AP_REQUIRE(src && dst, "ap_pillow_reduce_u8: null pointer");
    AP_REQUIRE(h > 0 && w > 0 && fx >= 1 && fy >= 1 && box_x >= 0 && box_y >= 0 && box_w > 0 && box_h > 0 &&
               box_x + box_w <= w && box_y + box_h <= h, "ap_pillow_reduce_u8: box (%d, %d, %d, %d) / factor (%d, %d) outside %d x %d",
               box_x, box_y, box_w, box_h, fx, fy, w, h);
    AP_REQUIRE((long long)fx * fy <= 65536, "ap_pillow_reduce_u8: factor %d x %d too large", fx, fy);     // sum stays inside uint32
    const int ow = (box_w + fx - 1) / fx, oh = (box_h + fy - 1) / fy;
    dim3 grid((unsigned)((ow + 63) / 64), (unsigned)((oh + 3) / 4));
    ap::pillow_reduce_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(src, w, box_x, box_y, box_w, box_h, fx, fy, ow, oh, dst);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}
