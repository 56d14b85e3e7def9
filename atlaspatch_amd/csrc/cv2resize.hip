// cv2.resize for uint8 RGB on the device, arithmetic restated from OpenCV 4.x's 8-bit resize so that the result is
// the one the reference gets on its host:
//   * services/feature_embedding.py:94-95 / services/extraction.py:112-113: cv2.resize(patch, (ps, ps)) (INTER_LINEAR) on
//     every tile whose level read is not patch_size (a 40x slide at --target-mag 20 with levels 1/4/16 reads 512 x 512);
//   * core/wsi/iwsi.py:305-321: cv2.resize(level image, (out_w, out_h), INTER_AREA | INTER_CUBIC | INTER_LINEAR) for the
//     1.25x thumbnail.
// With this kernel those tiles cross PCIe at their read size and never touch a host resampler.
//
// The per-axis tables (offsets, 11-bit fixed-point weights, area-cell weights) are built on the host in the same
// float / double operation order as OpenCV builds them (this file is compiled with -ffp-contract=off), cached per
// (device, h, w, oh, ow, mode) and kept in HBM.  Arithmetic per mode:
//   area, integer ratio   2 x 2: (a + b + c + d + 2) >> 2;  else round_half_even(float(sum) * (1.f / area))
//   area, general         float32 cell weights: buf = sum_k S * alpha_k (table order), sum (+)= beta * buf, round_half_even
//   linear (and area when enlarging)   t = S[sx] a0 + S[sx+1] a1;  (((b0 (t0 >> 4)) >> 16) + ((b1 (t1 >> 4)) >> 16) + 2) >> 2
//   cubic (A = -0.75)     4 x 4 taps, replicate border, int32 horizontal sums; vertical either int32 (+2^21) >> 22 or, for the
//                         elements OpenCV's 128-bit vector loop serves, float32 S0 b0 + (S1 b1 + (S2 b2 + S3 b3)), round_half_even
// INTER_LINEAR at exactly 2 x 2 is the 2 x 2 area average (OpenCV re-routes it).  Bit-exact against oracle/cv2_resize.py
// (tests/test_gpu_ops.py); OpenCV itself is absent from the image: parity unpinned.
//
// HBM-bound: n (h w + oh ow) 3 bytes algorithmic; one thread per output pixel, neighbouring lanes read neighbouring windows.
#include <cmath>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>
#include "ap_common.h"
#include <memory>

namespace ap {
namespace {

constexpr int kCoefScale = 2048;           // INTER_RESIZE_COEF_SCALE (11 bits)

enum Mode { M_COPY = 0, M_AREA_FAST = 1, M_AREA_GEN = 2, M_LINEAR = 3, M_CUBIC = 4 };

struct Tables {
    int mode = M_COPY;
    int isx = 1, isy = 1;                  // area fast
    // linear / cubic: xofs [ow], xw [ow * k], yofs [oh], yw [oh * k]  (int32)
    // area general:   xbeg [ow + 1], xsi [nx], yb [oh + 1], ysi [ny] (int32);  xal [nx], yal [ny] (float)
    int32_t* i32 = nullptr;
    float* f32 = nullptr;
    size_t off_xw = 0, off_yofs = 0, off_yw = 0;                 // into i32
    size_t off_xsi = 0, off_ybeg = 0, off_ysi = 0, off_yal = 0;  // area general
};

inline short sat_short(float v) {
    const long r = lrintf(v);              // cvRound: current rounding mode = to nearest even
    return (short)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
}

void cubic_weights(float x, float* c) {
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

// offsets + fixed-point weights of one axis (clamp_edges: the x axis folds out-of-range taps of the 2-tap filter)
void axis_fixed(int ssize, int dsize, double scale, double inv_scale, bool area_mode, bool cubic, bool clamp_edges,
                std::vector<int32_t>& ofs, std::vector<int32_t>& wts) {
    const int k = cubic ? 4 : 2;
    ofs.resize(dsize);
    wts.resize((size_t)dsize * k);
    for (int d = 0; d < dsize; ++d) {
        int s;
        float f;
        if (!area_mode) {
            f = (float)((d + 0.5) * scale - 0.5);
            s = (int)std::floor(f);
            f -= s;
        } else {
            s = (int)std::floor(d * scale);
            f = (float)((d + 1) - (s + 1) * inv_scale);
            f = f <= 0 ? 0.f : f - std::floor(f);
        }
        if (clamp_edges && !cubic) {
            if (s < 0) { f = 0; s = 0; }
            if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        }
        float c[4];
        if (cubic) cubic_weights(f, c);
        else { c[0] = 1.f - f; c[1] = f; }
        ofs[d] = s;
        for (int j = 0; j < k; ++j) wts[(size_t)d * k + j] = sat_short(c[j] * kCoefScale);
    }
}

// area cells of one axis: for destination d the taps [beg[d], beg[d+1]) of (source index, float weight)
void axis_area(int ssize, int dsize, double scale, std::vector<int32_t>& beg, std::vector<int32_t>& si, std::vector<float>& al) {
    beg.assign(dsize + 1, 0);
    si.clear();
    al.clear();
    for (int d = 0; d < dsize; ++d) {
        beg[d] = (int32_t)si.size();
        const double f1 = d * scale, f2 = f1 + scale;
        const double cell = std::min(scale, ssize - f1);
        int s1 = (int)std::ceil(f1), s2 = (int)std::floor(f2);
        s2 = std::min(s2, ssize - 1);
        s1 = std::min(s1, s2);
        if (s1 - f1 > 1e-3) { si.push_back(s1 - 1); al.push_back((float)((s1 - f1) / cell)); }
        for (int s = s1; s < s2; ++s) { si.push_back(s); al.push_back((float)(1.0 / cell)); }
        if (f2 - s2 > 1e-3) { si.push_back(s2); al.push_back((float)(std::min(std::min(f2 - s2, 1.), cell) / cell)); }
    }
    beg[dsize] = (int32_t)si.size();
}

// Table cache: one entry per (device, shape, interpolation), at most kMaxCached of them (every slide's 1.25x thumbnail has
// its own shape, so an unbounded cache would grow with the cohort).  Entries are handed out as shared_ptr: an evicted
// entry's device tables are freed when its last user lets go (hipFree waits for the launches that still read them).
using Key = std::tuple<int, int, int, int, int, int>;
constexpr size_t kMaxCached = 64;
struct Entry {
    Tables t;
    uint64_t last_use = 0;
    ~Entry() { if (t.i32) (void)hipFree(t.i32); if (t.f32) (void)hipFree(t.f32); }
};
std::mutex g_mu;
// never destroyed: at process exit the HIP runtime may be gone before static destructors run, and ~Entry calls hipFree
std::map<Key, std::shared_ptr<Entry>>& g_tables = *new std::map<Key, std::shared_ptr<Entry>>();
uint64_t g_use_clock = 0;

int upload(const std::vector<int32_t>& ints, const std::vector<float>& flts, Tables& t) {
    if (!ints.empty()) {
        AP_HIP_CHECK(hipMalloc((void**)&t.i32, ints.size() * 4));
        AP_HIP_CHECK(hipMemcpy(t.i32, ints.data(), ints.size() * 4, hipMemcpyHostToDevice));
    }
    if (!flts.empty()) {
        AP_HIP_CHECK(hipMalloc((void**)&t.f32, flts.size() * 4));
        AP_HIP_CHECK(hipMemcpy(t.f32, flts.data(), flts.size() * 4, hipMemcpyHostToDevice));
    }
    return AP_OK;
}

int get_tables(int h, int w, int oh, int ow, int interp, std::shared_ptr<Entry>* out) {
    int dev = 0;
    AP_HIP_CHECK(hipGetDevice(&dev));
    const Key key{dev, h, w, oh, ow, interp};
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_tables.find(key);
    if (it != g_tables.end()) { it->second->last_use = ++g_use_clock; *out = it->second; return AP_OK; }
    auto entry = std::make_shared<Entry>();
    Tables& t = entry->t;
    const double inv_x = (double)ow / w, inv_y = (double)oh / h;
    const double scale_x = 1. / inv_x, scale_y = 1. / inv_y;
    const int isx = (int)lrint(scale_x), isy = (int)lrint(scale_y);
    const bool area_fast = std::abs(scale_x - isx) < 2.220446049250313e-16 && std::abs(scale_y - isy) < 2.220446049250313e-16;
    int mode = interp;
    if (h == oh && w == ow) {
        t.mode = M_COPY;
    } else {
        if (mode == AP_CV_INTER_LINEAR && area_fast && isx == 2 && isy == 2) mode = AP_CV_INTER_AREA;
        if (mode == AP_CV_INTER_AREA && scale_x >= 1 && scale_y >= 1) {
            if (area_fast) {
                t.mode = M_AREA_FAST; t.isx = isx; t.isy = isy;
            } else {
                t.mode = M_AREA_GEN;
                std::vector<int32_t> xb, xs, yb, ys, ints;
                std::vector<float> xa, ya, flts;
                axis_area(w, ow, scale_x, xb, xs, xa);
                axis_area(h, oh, scale_y, yb, ys, ya);
                ints = xb; t.off_xsi = ints.size();
                ints.insert(ints.end(), xs.begin(), xs.end()); t.off_ybeg = ints.size();
                ints.insert(ints.end(), yb.begin(), yb.end()); t.off_ysi = ints.size();
                ints.insert(ints.end(), ys.begin(), ys.end());
                flts = xa; t.off_yal = flts.size();
                flts.insert(flts.end(), ya.begin(), ya.end());
                const int rc = upload(ints, flts, t);
                if (rc != AP_OK) return rc;
            }
        } else {
            const bool cubic = mode == AP_CV_INTER_CUBIC, area_mode = mode == AP_CV_INTER_AREA;
            t.mode = cubic ? M_CUBIC : M_LINEAR;
            std::vector<int32_t> xo, xw, yo, yw, ints;
            axis_fixed(w, ow, scale_x, inv_x, area_mode, cubic, true, xo, xw);
            axis_fixed(h, oh, scale_y, inv_y, area_mode, cubic, false, yo, yw);
            ints = xo; t.off_xw = ints.size();
            ints.insert(ints.end(), xw.begin(), xw.end()); t.off_yofs = ints.size();
            ints.insert(ints.end(), yo.begin(), yo.end()); t.off_yw = ints.size();
            ints.insert(ints.end(), yw.begin(), yw.end());
            const int rc = upload(ints, {}, t);
            if (rc != AP_OK) return rc;
        }
    }
    if (g_tables.size() >= kMaxCached) {          // drop the least recently used entry
        auto lru = g_tables.begin();
        for (auto i = g_tables.begin(); i != g_tables.end(); ++i)
            if (i->second->last_use < lru->second->last_use) lru = i;
        g_tables.erase(lru);
    }
    entry->last_use = ++g_use_clock;
    g_tables[key] = entry;
    *out = entry;
    return AP_OK;
}

__device__ __forceinline__ uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

#define AP_RS_PIXEL_INDEX                                                      \
    const size_t total = (size_t)n * oh * ow;                                  \
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;            \
    if (i >= total) return;                                                    \
    const int dx = (int)(i % ow);                                              \
    const size_t r_ = i / ow;                                                  \
    const int dy = (int)(r_ % oh);                                             \
    const uint8_t* img = src + (r_ / oh) * (size_t)h * w * 3;                  \
    uint8_t* d = dst + i * 3

__global__ void area_fast_kernel(const uint8_t* __restrict__ src, int n, int h, int w, int oh, int ow, int isx, int isy,
                                 uint8_t* __restrict__ dst) {
    AP_RS_PIXEL_INDEX;
    int s0 = 0, s1 = 0, s2 = 0;
    for (int y = 0; y < isy; ++y) {
        const uint8_t* p = img + ((size_t)(dy * isy + y) * w + (size_t)dx * isx) * 3;
        for (int x = 0; x < isx; ++x) { s0 += p[x * 3]; s1 += p[x * 3 + 1]; s2 += p[x * 3 + 2]; }
    }
    if (isx == 2 && isy == 2) {
        d[0] = (uint8_t)((s0 + 2) >> 2); d[1] = (uint8_t)((s1 + 2) >> 2); d[2] = (uint8_t)((s2 + 2) >> 2);
    } else {
        const float scale = 1.f / (float)(isx * isy);
        d[0] = sat_u8(__float2int_rn(__fmul_rn((float)s0, scale)));
        d[1] = sat_u8(__float2int_rn(__fmul_rn((float)s1, scale)));
        d[2] = sat_u8(__float2int_rn(__fmul_rn((float)s2, scale)));
    }
}

// The exact 2 x 2 shrink (what a 40x slide at --target-mag 20 sends every tile through), four output pixels per thread:
// two rows of 24 input bytes arrive as three 8-byte loads each, the 12 output bytes leave as three dwords.
// out = (a + b + c + d + 2) >> 2 per channel, as area_fast_kernel.  Requires ow % 4 == 0 (rows stay 8-byte aligned).
__global__ void area_2x2_quad_kernel(const uint8_t* __restrict__ src, size_t quads, int oh, int ow, uint8_t* __restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;           // quad index over [n][oh][ow / 4]
    if (i >= quads) return;
    const int qpr = ow >> 2;
    const size_t row = i / qpr;                                                  // (image, output row)
    const int q = (int)(i - row * qpr);
    const size_t in_stride = (size_t)ow * 6;                                     // w * 3 with w = 2 ow
    const uint8_t* p0 = src + (row * 2) * in_stride + (size_t)q * 24;            // image rows are contiguous: row * 2 spans images
    const uint2* a = (const uint2*)p0;
    const uint2* b = (const uint2*)(p0 + in_stride);
    uint32_t ra[6], rb[6];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint2 va = a[k], vb = b[k];
        ra[2 * k] = va.x; ra[2 * k + 1] = va.y; rb[2 * k] = vb.x; rb[2 * k + 1] = vb.y;
    }
    auto byte = [](const uint32_t (&r)[6], int idx) { return (r[idx >> 2] >> ((idx & 3) * 8)) & 255u; };
    uint32_t o[3] = {0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 12; ++j) {                                               // output byte j = pixel j / 3, channel j % 3
        const int px = j / 3, c = j - px * 3;
        const uint32_t s = byte(ra, 6 * px + c) + byte(ra, 6 * px + 3 + c) + byte(rb, 6 * px + c) + byte(rb, 6 * px + 3 + c);
        o[j >> 2] |= ((s + 2u) >> 2) << ((j & 3) * 8);
    }
    uint32_t* d = (uint32_t*)(dst + row * (size_t)ow * 3 + (size_t)q * 12);
    d[0] = o[0]; d[1] = o[1]; d[2] = o[2];
}

__global__ void area_general_kernel(const uint8_t* __restrict__ src, int n, int h, int w, int oh, int ow,
                                    const int32_t* __restrict__ xbeg, const int32_t* __restrict__ xsi,
                                    const float* __restrict__ xal, const int32_t* __restrict__ ybeg,
                                    const int32_t* __restrict__ ysi, const float* __restrict__ yal,
                                    uint8_t* __restrict__ dst) {
    AP_RS_PIXEL_INDEX;
    const int x0 = xbeg[dx], x1 = xbeg[dx + 1], y0 = ybeg[dy], y1 = ybeg[dy + 1];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int j = y0; j < y1; ++j) {
        const uint8_t* row = img + (size_t)ysi[j] * w * 3;
        float b0 = 0.f, b1 = 0.f, b2 = 0.f;
        for (int k = x0; k < x1; ++k) {
            const uint8_t* p = row + (size_t)xsi[k] * 3;
            const float al = xal[k];
            b0 = __fadd_rn(b0, __fmul_rn((float)p[0], al));
            b1 = __fadd_rn(b1, __fmul_rn((float)p[1], al));
            b2 = __fadd_rn(b2, __fmul_rn((float)p[2], al));
        }
        const float be = yal[j];
        a0 = __fadd_rn(a0, __fmul_rn(be, b0));
        a1 = __fadd_rn(a1, __fmul_rn(be, b1));
        a2 = __fadd_rn(a2, __fmul_rn(be, b2));
    }
    d[0] = sat_u8(__float2int_rn(a0)); d[1] = sat_u8(__float2int_rn(a1)); d[2] = sat_u8(__float2int_rn(a2));
}

__global__ void linear_kernel(const uint8_t* __restrict__ src, int n, int h, int w, int oh, int ow,
                              const int32_t* __restrict__ xofs, const int32_t* __restrict__ xw,
                              const int32_t* __restrict__ yofs, const int32_t* __restrict__ yw, uint8_t* __restrict__ dst) {
    AP_RS_PIXEL_INDEX;
    const int sx = xofs[dx], sx1 = sx + 1 < w ? sx + 1 : w - 1;
    const int a0 = xw[dx * 2], a1 = xw[dx * 2 + 1];
    const int sy = yofs[dy];
    const int b0 = yw[dy * 2], b1 = yw[dy * 2 + 1];
    const uint8_t* r0 = img + (size_t)clampi(sy, 0, h - 1) * w * 3;
    const uint8_t* r1 = img + (size_t)clampi(sy + 1, 0, h - 1) * w * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int t0 = r0[sx * 3 + c] * a0 + r0[sx1 * 3 + c] * a1;
        const int t1 = r1[sx * 3 + c] * a0 + r1[sx1 * 3 + c] * a1;
        d[c] = (uint8_t)((((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2);
    }
}

__global__ void cubic_kernel(const uint8_t* __restrict__ src, int n, int h, int w, int oh, int ow,
                             const int32_t* __restrict__ xofs, const int32_t* __restrict__ xw,
                             const int32_t* __restrict__ yofs, const int32_t* __restrict__ yw, int scalar_only,
                             uint8_t* __restrict__ dst) {
    AP_RS_PIXEL_INDEX;
    const int sx = xofs[dx], sy = yofs[dy];
    int xs[4], al[4], be[4];
    const uint8_t* rows[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        xs[j] = clampi(sx - 1 + j, 0, w - 1) * 3;
        al[j] = xw[dx * 4 + j];
        be[j] = yw[dy * 4 + j];
        rows[j] = img + (size_t)clampi(sy - 1 + j, 0, h - 1) * w * 3;
    }
    const int nvec = (ow * 3 / 8) * 8;     // elements of a row served by the 8-lane vector loop
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int S[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            S[k] = rows[k][xs[0] + c] * al[0] + rows[k][xs[1] + c] * al[1] + rows[k][xs[2] + c] * al[2] + rows[k][xs[3] + c] * al[3];
        if (scalar_only || dx * 3 + c >= nvec) {
            const unsigned acc = (unsigned)S[0] * (unsigned)be[0] + (unsigned)S[1] * (unsigned)be[1] +
                                 (unsigned)S[2] * (unsigned)be[2] + (unsigned)S[3] * (unsigned)be[3] + (1u << 21);
            d[c] = sat_u8((int)acc >> 22);
        } else {
            const float sc = 1.f / (float)(kCoefScale * kCoefScale);
            float v = __fmul_rn((float)S[3], __fmul_rn((float)be[3], sc));
            v = __fadd_rn(__fmul_rn((float)S[2], __fmul_rn((float)be[2], sc)), v);
            v = __fadd_rn(__fmul_rn((float)S[1], __fmul_rn((float)be[1], sc)), v);
            v = __fadd_rn(__fmul_rn((float)S[0], __fmul_rn((float)be[0], sc)), v);
            d[c] = sat_u8(__float2int_rn(v));
        }
    }
}

}  // namespace
}  // namespace ap

extern "C" int ap_cv2_resize_u8(const uint8_t* src, int n, int h, int w, uint8_t* dst, int oh, int ow, int interpolation,
                                int flags, ap_stream_t stream) {
    using namespace ap;
    AP_REQUIRE(n >= 0 && h > 0 && w > 0 && oh > 0 && ow > 0, "ap_cv2_resize_u8: bad shape");
    AP_REQUIRE(interpolation == AP_CV_INTER_LINEAR || interpolation == AP_CV_INTER_CUBIC || interpolation == AP_CV_INTER_AREA,
               "ap_cv2_resize_u8: unsupported interpolation %d", interpolation);
    if (n == 0) return AP_OK;                 // an empty batch has no buffers to check
    AP_REQUIRE(src && dst, "ap_cv2_resize_u8: null pointer");
    std::shared_ptr<Entry> entry;             // keeps the tables alive through the launch below
    const int rc = get_tables(h, w, oh, ow, interpolation, &entry);
    if (rc != AP_OK) return rc;
    const Tables* t = &entry->t;
    hipStream_t s = (hipStream_t)stream;
    const size_t total = (size_t)n * oh * ow;
    const unsigned grid = (unsigned)((total + 255) / 256);
    switch (t->mode) {
    case M_COPY:
        AP_HIP_CHECK(hipMemcpyAsync(dst, src, total * 3, hipMemcpyDeviceToDevice, s));
        return AP_OK;
    case M_AREA_FAST:
        // the quad kernel reads 8 bytes and stores 4 at a time: only for pointers aligned accordingly (the C ABI takes any)
        if (t->isx == 2 && t->isy == 2 && ow % 4 == 0 && w == 2 * ow && h == 2 * oh && ((uintptr_t)src & 7) == 0 &&
            ((uintptr_t)dst & 3) == 0) {
            const size_t quads = (size_t)n * oh * (ow / 4);
            area_2x2_quad_kernel<<<(unsigned)((quads + 255) / 256), 256, 0, s>>>(src, quads, oh, ow, dst);
            break;
        }
        area_fast_kernel<<<grid, 256, 0, s>>>(src, n, h, w, oh, ow, t->isx, t->isy, dst);
        break;
    case M_AREA_GEN:
        area_general_kernel<<<grid, 256, 0, s>>>(src, n, h, w, oh, ow, t->i32, t->i32 + t->off_xsi, t->f32,
                                                 t->i32 + t->off_ybeg, t->i32 + t->off_ysi, t->f32 + t->off_yal, dst);
        break;
    case M_LINEAR:
        linear_kernel<<<grid, 256, 0, s>>>(src, n, h, w, oh, ow, t->i32, t->i32 + t->off_xw, t->i32 + t->off_yofs,
                                           t->i32 + t->off_yw, dst);
        break;
    default:
        cubic_kernel<<<grid, 256, 0, s>>>(src, n, h, w, oh, ow, t->i32, t->i32 + t->off_xw, t->i32 + t->off_yofs,
                                          t->i32 + t->off_yw, (flags & AP_CV_CUBIC_SCALAR) ? 1 : 0, dst);
        break;
    }
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}
