// float32 operator set of the SAM2 (Hiera-T) image path: segmentation runs once per slide on one 1024 x 1024
// thumbnail, so these kernels favour exact f32 arithmetic and generality (any M, N, K; any window size) over peak
// rate: ~175 GFLOP per slide, a few tens of milliseconds.  The host (atlaspatch_amd/services/sam2_hip.py) chains
// them through the C ABI; every tensor is channels-last ([tokens, C]).
//
//   ap_sgemm            out = act(alpha * A W^T + bias) + resid, batched with strides; W either [N, K] ("NT",
//                       nn.Linear / 1x1 conv layout) or [K, N] ("NN", for P V).  64 x 64 tile, 16-deep K step
//                       through LDS, 4 x 4 outputs per thread, fully bounds-checked.
//   ap_softmax_rows     in-place row softmax (one wave per row)
//   ap_sam2_patchify    uint8 HWC image -> ImageNet normalise -> im2col rows of the 7x7 stride-4 pad-3 patch embed
//   ap_window_partition / ap_window_unpartition   (zero padded, hieradet.py window_partition semantics)
//   ap_maxpool2x2       2x2 / stride 2 max pool on [B, H, W, C] with an input row stride (q pooling, shortcut)
//   ap_add / ap_add_rowvec / ap_gelu / ap_upsample2x_add / ap_convt2x2_shuffle / ap_bilinear_up4_threshold
#include "ap_common.h"

namespace ap {
namespace {

constexpr int kGT = 64;     // GEMM tile
constexpr int kGK = 16;

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == 1) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    if (act == 2) return v > 0.f ? v : 0.f;
    return v;
}

struct SgemmArgs {
    const float* A; long lda, sA;
    const float* W; long ldw, sW; int w_kn;
    int M, N, K;
    float alpha;
    const float* bias; int act;
    const float* resid; long ldr, sR;
    float* out; long ldo, sO;
};

__global__ __launch_bounds__(256) void sgemm_kernel(SgemmArgs g) {
    __shared__ float As[kGK][kGT + 4];
    __shared__ float Ws[kGK][kGT + 4];
    const int b = blockIdx.z;
    const float* A = g.A + (size_t)b * g.sA;
    const float* W = g.W + (size_t)b * g.sW;
    const int m0 = blockIdx.y * kGT, n0 = blockIdx.x * kGT;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;      // 16 x 16 threads, 4 x 4 outputs each
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    for (int k0 = 0; k0 < g.K; k0 += kGK) {
        // A tile: 64 rows x 16 k  (thread -> row = tid / 4, 4 consecutive k)
        {
            const int r = threadIdx.x >> 2, kk = (threadIdx.x & 3) * 4;
            const int m = m0 + r;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = k0 + kk + e;
                As[kk + e][r] = (m < g.M && k < g.K) ? A[(size_t)m * g.lda + k] : 0.f;
            }
        }
        if (!g.w_kn) {
            const int r = threadIdx.x >> 2, kk = (threadIdx.x & 3) * 4;
            const int n = n0 + r;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = k0 + kk + e;
                Ws[kk + e][r] = (n < g.N && k < g.K) ? W[(size_t)n * g.ldw + k] : 0.f;
            }
        } else {
            const int kk = threadIdx.x >> 4, c = (threadIdx.x & 15) * 4;      // W[k][n]: 16 k x 64 n
            const int k = k0 + kk;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + c + e;
                Ws[kk][c + e] = (n < g.N && k < g.K) ? W[(size_t)k * g.ldw + n] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < kGK; ++kk) {
            float a[4], w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = Ws[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fmaf(a[i], w[j], acc[i][j]);
        }
        __syncthreads();
    }
    float* out = g.out + (size_t)b * g.sO;
    const float* resid = g.resid ? g.resid + (size_t)b * g.sR : nullptr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= g.N) continue;
            float v = acc[i][j] * g.alpha;
            if (g.bias) v += g.bias[n];
            v = act_apply(v, g.act);
            if (resid) v += resid[(size_t)m * g.ldr + n];
            out[(size_t)m * g.ldo + n] = v;
        }
    }
}

__global__ __launch_bounds__(256) void softmax_rows_kernel(float* x, long ld, int rows, int cols) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float* p = x + (size_t)row * ld;
    float mx = -INFINITY;
    for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, p[c]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) {
        const float e = expf(p[c] - mx);
        p[c] = e;
        s += e;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const float inv = 1.0f / s;
    for (int c = lane; c < cols; c += 64) p[c] *= inv;
}

// image u8 [H, W, 3] -> rows [(oy * OW + ox)][c * 49 + ky * 7 + kx] of ((x / 255) - mean) / std, zero padded
__global__ void patchify_kernel(const uint8_t* img, int H, int W, float* out, int OH, int OW,
                                float m0, float m1, float m2, float s0, float s1, float s2) {
    const size_t total = (size_t)OH * OW * 147;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int col = (int)(i % 147);
    const size_t pix = i / 147;
    const int ox = (int)(pix % OW), oy = (int)(pix / OW);
    const int c = col / 49, ky = (col % 49) / 7, kx = col % 7;
    const int y = oy * 4 - 3 + ky, x = ox * 4 - 3 + kx;
    float v = 0.f;
    if (y >= 0 && y < H && x >= 0 && x < W) {
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
        v = (((float)img[((size_t)y * W + x) * 3 + c] / 255.0f) - mean) / sd;
    }
    out[i] = v;
}

// x [B, H, W, C] -> win [(b, wy, wx)][ws * ws][C], zero padded to multiples of ws
__global__ void window_partition_kernel(const float* x, int B, int H, int W, int C, int ws, int nwy, int nwx, float* win) {
    const size_t total = (size_t)B * nwy * nwx * ws * ws * C;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C); size_t r = i / C;
    const int ix = (int)(r % ws); r /= ws;
    const int iy = (int)(r % ws); r /= ws;
    const int wx = (int)(r % nwx); r /= nwx;
    const int wy = (int)(r % nwy); const int b = (int)(r / nwy);
    const int y = wy * ws + iy, xx = wx * ws + ix;
    win[i] = (y < H && xx < W) ? x[(((size_t)b * H + y) * W + xx) * C + c] : 0.f;
}

__global__ void window_unpartition_kernel(const float* win, int B, int H, int W, int C, int ws, int nwy, int nwx, float* x) {
    const size_t total = (size_t)B * H * W * C;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C); size_t r = i / C;
    const int xx = (int)(r % W); r /= W;
    const int y = (int)(r % H); const int b = (int)(r / H);
    const int wy = y / ws, iy = y % ws, wx = xx / ws, ix = xx % ws;
    x[i] = win[((((size_t)(b * nwy + wy) * nwx + wx) * ws + iy) * ws + ix) * C + c];
}

// in [B, H, W, C] with row stride ld_in (elements per pixel) -> out [B, H/2, W/2, C] dense
__global__ void maxpool2x2_kernel(const float* in, long ld_in, int B, int H, int W, int C, float* out) {
    const int OH = H / 2, OW = W / 2;
    const size_t total = (size_t)B * OH * OW * C;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C); size_t r = i / C;
    const int ox = (int)(r % OW); r /= OW;
    const int oy = (int)(r % OH); const int b = (int)(r / OH);
    const float* p = in + (((size_t)b * H + oy * 2) * W + ox * 2) * ld_in + c;
    const float a = p[0], bb = p[ld_in], cc = p[(size_t)W * ld_in], d = p[(size_t)W * ld_in + ld_in];
    out[i] = fmaxf(fmaxf(a, bb), fmaxf(cc, d));
}

__global__ void add_kernel(float* out, const float* a, const float* b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}
__global__ void add_rowvec_kernel(float* out, const float* a, const float* vec, size_t rows, int cols) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * cols) out[i] = a[i] + vec[i % cols];
}
__global__ void gelu_kernel(float* x, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = act_apply(x[i], 1);
}
// out [2H, 2W, C] = lat [2H, 2W, C] + nearest-upsampled prev [H, W, C]
__global__ void upsample2x_add_kernel(float* out, const float* lat, const float* prev, int H, int W, int C) {
    const size_t total = (size_t)4 * H * W * C;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C); size_t r = i / C;
    const int x = (int)(r % (2 * W)); const int y = (int)(r / (2 * W));
    out[i] = lat[i] + prev[((size_t)(y / 2) * W + x / 2) * C + c];
}
// ConvTranspose2d(k 2, s 2) as a GEMM: g [H * W, Cout * 4] with column co * 4 + dy * 2 + dx
// -> out [2H, 2W, Cout] = g + bias[co] + skip, optional GELU
__global__ void convt_shuffle_kernel(const float* g, const float* bias, const float* skip, float* out, int H, int W,
                                     int Cout, int act) {
    const size_t total = (size_t)4 * H * W * Cout;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int co = (int)(i % Cout); size_t r = i / Cout;
    const int x = (int)(r % (2 * W)); const int y = (int)(r / (2 * W));
    float v = g[((size_t)(y / 2) * W + x / 2) * (Cout * 4) + co * 4 + (y & 1) * 2 + (x & 1)] + bias[co];
    if (skip) v += skip[i];
    out[i] = act_apply(v, act);
}
// F.interpolate(bilinear, align_corners=False) x4 of logits [S, S] and `> threshold` -> float {0, 1}
__global__ void bilinear_up4_threshold_kernel(const float* lg, int S, float thr, float* mask) {
    const int OS = S * 4;
    const size_t total = (size_t)OS * OS;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ox = (int)(i % OS), oy = (int)(i / OS);
    float sy = ((float)oy + 0.5f) * 0.25f - 0.5f, sx = ((float)ox + 0.5f) * 0.25f - 0.5f;
    sy = sy < 0.f ? 0.f : sy; sx = sx < 0.f ? 0.f : sx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < S - 1 ? 1 : 0), x1 = x0 + (x0 < S - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float hy = 1.0f - ly, hx = 1.0f - lx;
    const float v = hy * (hx * lg[(size_t)y0 * S + x0] + lx * lg[(size_t)y0 * S + x1]) +
                    ly * (hx * lg[(size_t)y1 * S + x0] + lx * lg[(size_t)y1 * S + x1]);
    mask[i] = v > thr ? 1.0f : 0.0f;
}

inline dim3 grid1(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace
}  // namespace ap

extern "C" {

int ap_sgemm(const float* A, long lda, long strideA, const float* W, long ldw, long strideW, int w_is_kn,
             int batch, int M, int N, int K, float alpha, const float* bias, int act,
             const float* resid, long ldr, long strideR, float* out, long ldo, long strideO, ap_stream_t stream) {
    AP_REQUIRE(A && W && out, "ap_sgemm: null pointer");
    AP_REQUIRE(batch > 0 && M > 0 && N > 0 && K > 0 && batch <= 65535, "ap_sgemm: bad problem %d x %d x %d x %d", batch, M, N, K);
    AP_REQUIRE(act >= 0 && act <= 2, "ap_sgemm: activation %d", act);
    ap::SgemmArgs g{A, lda, strideA, W, ldw, strideW, w_is_kn, M, N, K, alpha, bias, act, resid, ldr, strideR, out, ldo, strideO};
    const int gy = (M + ap::kGT - 1) / ap::kGT;
    AP_REQUIRE(gy <= 65535, "ap_sgemm: M %d too large", M);
    dim3 grid((N + ap::kGT - 1) / ap::kGT, gy, batch);
    ap::sgemm_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(g);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int ap_softmax_rows(float* x, long ld, int rows, int cols, ap_stream_t stream) {
    AP_REQUIRE(x && rows > 0 && cols > 0, "ap_softmax_rows: bad arguments");
    ap::softmax_rows_kernel<<<(rows + 3) / 4, 256, 0, (hipStream_t)stream>>>(x, ld, rows, cols);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int ap_sam2_patchify(const uint8_t* image, int h, int w, const float mean[3], const float stdv[3], float* out,
                     ap_stream_t stream) {
    AP_REQUIRE(image && out && mean && stdv && h > 0 && w > 0 && h % 4 == 0 && w % 4 == 0, "ap_sam2_patchify: bad arguments");
    const int OH = h / 4, OW = w / 4;
    ap::patchify_kernel<<<ap::grid1((size_t)OH * OW * 147), 256, 0, (hipStream_t)stream>>>(
        image, h, w, out, OH, OW, mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2]);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int ap_window_partition(const float* x, int b, int h, int w, int c, int ws, float* win, ap_stream_t stream) {
    AP_REQUIRE(x && win && ws > 0, "ap_window_partition: bad arguments");
    const int nwy = (h + ws - 1) / ws, nwx = (w + ws - 1) / ws;
    ap::window_partition_kernel<<<ap::grid1((size_t)b * nwy * nwx * ws * ws * c), 256, 0, (hipStream_t)stream>>>(
        x, b, h, w, c, ws, nwy, nwx, win);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int ap_window_unpartition(const float* win, int b, int h, int w, int c, int ws, float* x, ap_stream_t stream) {
    AP_REQUIRE(x && win && ws > 0, "ap_window_unpartition: bad arguments");
    const int nwy = (h + ws - 1) / ws, nwx = (w + ws - 1) / ws;
    ap::window_unpartition_kernel<<<ap::grid1((size_t)b * h * w * c), 256, 0, (hipStream_t)stream>>>(
        win, b, h, w, c, ws, nwy, nwx, x);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int ap_maxpool2x2(const float* in, long ld_in, int b, int h, int w, int c, float* out, ap_stream_t stream) {
    AP_REQUIRE(in && out && h % 2 == 0 && w % 2 == 0 && ld_in >= c, "ap_maxpool2x2: bad arguments");
    ap::maxpool2x2_kernel<<<ap::grid1((size_t)b * (h / 2) * (w / 2) * c), 256, 0, (hipStream_t)stream>>>(in, ld_in, b, h, w, c, out);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int ap_add(float* out, const float* a, const float* b, size_t n, ap_stream_t stream) {
    AP_REQUIRE(out && a && b, "ap_add: null pointer");
    if (n) ap::add_kernel<<<ap::grid1(n), 256, 0, (hipStream_t)stream>>>(out, a, b, n);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int ap_add_rowvec(float* out, const float* a, const float* vec, size_t rows, int cols, ap_stream_t stream) {
    AP_REQUIRE(out && a && vec && cols > 0, "ap_add_rowvec: bad arguments");
    if (rows) ap::add_rowvec_kernel<<<ap::grid1(rows * cols), 256, 0, (hipStream_t)stream>>>(out, a, vec, rows, cols);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int ap_gelu(float* x, size_t n, ap_stream_t stream) {
    AP_REQUIRE(x, "ap_gelu: null pointer");
    if (n) ap::gelu_kernel<<<ap::grid1(n), 256, 0, (hipStream_t)stream>>>(x, n);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int ap_upsample2x_add(float* out, const float* lateral, const float* prev, int h, int w, int c, ap_stream_t stream) {
    AP_REQUIRE(out && lateral && prev, "ap_upsample2x_add: null pointer");
    ap::upsample2x_add_kernel<<<ap::grid1((size_t)4 * h * w * c), 256, 0, (hipStream_t)stream>>>(out, lateral, prev, h, w, c);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int ap_convt2x2_shuffle(const float* g, const float* bias, const float* skip, float* out, int h, int w, int cout,
                        int act, ap_stream_t stream) {
    AP_REQUIRE(g && bias && out, "ap_convt2x2_shuffle: null pointer");
    ap::convt_shuffle_kernel<<<ap::grid1((size_t)4 * h * w * cout), 256, 0, (hipStream_t)stream>>>(g, bias, skip, out, h, w, cout, act);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int ap_bilinear_up4_threshold(const float* logits, int size, float threshold, float* mask, ap_stream_t stream) {
    AP_REQUIRE(logits && mask && size > 1, "ap_bilinear_up4_threshold: bad arguments");
    ap::bilinear_up4_threshold_kernel<<<ap::grid1((size_t)16 * size * size), 256, 0, (hipStream_t)stream>>>(logits, size, threshold, mask);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

}  // extern "C"
