// float32 operator set of the SAM2 (Hiera-T) image path: segmentation runs once per slide on one 1024 x 1024
// thumbnail, so these kernels favour exact f32 arithmetic and generality (any M, N, K; any window size) over peak
// rate: ~175 GFLOP per slide, a few tens of milliseconds.  The host (atlaspatch_amd/services/sam2_hip.py) chains
// them through the C ABI; every tensor is channels-last ([tokens, C]).
//
//   ap_sgemm            out = act(alpha * A W^T + bias) + resid, batched with strides; W either [N, K] ("NT",
//                       nn.Linear / 1x1 conv layout) or [K, N] ("NN", for P V).  Exact-f32 MFMA
//                       (v_mfma_f32_32x32x2_f32), 128 x 128 or 64 x 64 tile, 16-deep K tiles, fully bounds-checked.
//   ap_softmax_rows     in-place row softmax (one wave per row)
//   ap_sam2_patchify    uint8 HWC image -> ImageNet normalise -> im2col rows of the 7x7 stride-4 pad-3 patch embed
//   ap_window_partition / ap_window_unpartition   (zero padded, hieradet.py window_partition semantics)
//   ap_maxpool2x2       2x2 / stride 2 max pool on [B, H, W, C] with an input row stride (q pooling, shortcut)
//   ap_add / ap_add_rowvec / ap_gelu / ap_upsample2x_add / ap_convt2x2_shuffle / ap_bilinear_up4_threshold
#include <algorithm>
#include <map>
#include <mutex>
#include "ap_common.h"

namespace ap {
namespace {

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
    int vec;                 // operands allow 16-byte loads (alignment and K / N multiples checked by the host)
    int splits, k_chunk;     // split-K: blockIdx.z = batch * splits + s works on k in [s * k_chunk, (s + 1) * k_chunk)
    float* partial;          //   and stores its raw accumulators to partial[blockIdx.z][M][N] (reduced by splitk_reduce_kernel)
};

// Exact-f32 MFMA GEMM (v_mfma_f32_32x32x2_f32), any M / N / K, batched, NT or NN.
//   T x T output tile per 256-thread workgroup (T = 128: 2 x 2 waves of 64 x 64 = four 32 x 32 MFMA blocks each;
//   T = 64: one block per wave for the small batched attention products), 16-deep K tiles.
//   Global -> registers (float4 per thread, next K tile in flight under the MFMAs) -> LDS (two buffers, one barrier per
//   K tile).  LDS rows hold [row][16 k] padded to 20 floats: a lane reads its 8 k values with two ds_read_b128 and rows
//   80 B apart spread over all 32 banks.  The MFMA's two k slots are fed k = e and k = 8 + e (lane halves), the same
//   mapping on both operands, so the sum runs over all 16 k of the tile.  The NN weight ([k][n], P V products) keeps its
//   [k][n] layout in LDS and is read with 8 conflict-free ds_read_b32.
//   D[i = m][j = n]: a lane owns column n = lane % 32 and 16 rows -> stores / residual reads are 128-byte row segments.
template <int T, bool WKN>
__global__ __launch_bounds__(256) void sgemm_mfma_kernel(SgemmArgs g) {
    constexpr int BLK = T / 64, LDR = 20, LDN = T + 4, PER = T / 64;
    __shared__ __attribute__((aligned(16))) float As[2][T * LDR];
    __shared__ __attribute__((aligned(16))) float Ws[2][WKN ? 16 * LDN : T * LDR];
    const int b = blockIdx.z / g.splits, split = blockIdx.z % g.splits;
    const float* A = g.A + (size_t)b * g.sA;
    const float* W = g.W + (size_t)b * g.sW;
    const int m0 = blockIdx.y * T, n0 = blockIdx.x * T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 1, wn = wave >> 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int kbeg = split * g.k_chunk, kend = (kbeg + g.k_chunk < g.K) ? kbeg + g.k_chunk : g.K;

    f32x4 ra[PER], rw[PER];
    auto load_rows = [&](const float* base, long ld, int row0, int rows, int k0, f32x4* dst) {     // [row][k] operand
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int f = tid + 256 * j, r = f >> 2, k = k0 + (f & 3) * 4;
            int gr = row0 + r;
            gr = gr < rows ? gr : rows - 1;
            const float* p = base + (size_t)gr * ld + k;
            if (g.vec) {
                dst[j] = k < kend ? *(const f32x4*)p : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[j][e] = (k + e < kend) ? p[e] : 0.f;
            }
        }
    };
    auto load_kn = [&](int k0, f32x4* dst) {                                                        // W[k][n]
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int f = tid + 256 * j, kr = f / (T / 4), n = n0 + (f % (T / 4)) * 4, k = k0 + kr;
            const float* p = W + (size_t)k * g.ldw + n;
            if (k < kend && g.vec && n + 3 < g.N) {
                dst[j] = *(const f32x4*)p;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[j][e] = (k < kend && n + e < g.N) ? p[e] : 0.f;
            }
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int f = tid + 256 * j;
            *(f32x4*)&As[buf][(f >> 2) * LDR + (f & 3) * 4] = ra[j];
            if constexpr (WKN) *(f32x4*)&Ws[buf][(f / (T / 4)) * LDN + (f % (T / 4)) * 4] = rw[j];
            else *(f32x4*)&Ws[buf][(f >> 2) * LDR + (f & 3) * 4] = rw[j];
        }
    };
    auto load_tile = [&](int k0) {
        load_rows(A, g.lda, m0, g.M, k0, ra);
        if constexpr (WKN) load_kn(k0, rw);
        else load_rows(W, g.ldw, n0, g.N, k0, rw);
    };

    f32x16 acc[BLK][BLK];
#pragma unroll
    for (int i = 0; i < BLK; ++i)
#pragma unroll
        for (int j = 0; j < BLK; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (kend - kbeg + 15) / 16;
    load_tile(kbeg);
    store_lds(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(kbeg + (kt + 1) * 16);
        float a[BLK][8], w[BLK][8];
#pragma unroll
        for (int i = 0; i < BLK; ++i) {
            const float* pa = &As[cur][(wm * (T / 2) + i * 32 + l31) * LDR + hi * 8];
            const f32x4 a0 = *(const f32x4*)pa, a1 = *(const f32x4*)(pa + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[i][e] = a0[e]; a[i][4 + e] = a1[e]; }
            if constexpr (WKN) {
#pragma unroll
                for (int e = 0; e < 8; ++e) w[i][e] = Ws[cur][(hi * 8 + e) * LDN + wn * (T / 2) + i * 32 + l31];
            } else {
                const float* pw = &Ws[cur][(wn * (T / 2) + i * 32 + l31) * LDR + hi * 8];
                const f32x4 w0 = *(const f32x4*)pw, w1 = *(const f32x4*)(pw + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { w[i][e] = w0[e]; w[i][4 + e] = w1[e]; }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int i = 0; i < BLK; ++i)
#pragma unroll
                for (int j = 0; j < BLK; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], w[j][e], acc[i][j], 0, 0, 0);
        if (kt + 1 < nk) store_lds(cur ^ 1);
        __syncthreads();
    }

    if (g.splits > 1) {                 // raw partial sums; alpha / bias / activation / residual in the reduce kernel
        float* part = g.partial + (size_t)blockIdx.z * g.M * g.N;
#pragma unroll
        for (int j = 0; j < BLK; ++j) {
            const int n = n0 + wn * (T / 2) + j * 32 + l31;
            if (n >= g.N) continue;
#pragma unroll
            for (int i = 0; i < BLK; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * (T / 2) + i * 32 + (r >> 2) * 8 + hi * 4 + (r & 3);
                    if (m < g.M) part[(size_t)m * g.N + n] = acc[i][j][r];
                }
        }
        return;
    }
    float* out = g.out + (size_t)b * g.sO;
    const float* resid = g.resid ? g.resid + (size_t)b * g.sR : nullptr;
#pragma unroll
    for (int j = 0; j < BLK; ++j) {
        const int n = n0 + wn * (T / 2) + j * 32 + l31;
        if (n >= g.N) continue;
        const float bias = g.bias ? g.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < BLK; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (T / 2) + i * 32 + (r >> 2) * 8 + hi * 4 + (r & 3);
                if (m >= g.M) continue;
                float v = acc[i][j][r] * g.alpha + bias;
                v = act_apply(v, g.act);
                if (resid) v += resid[(size_t)m * g.ldr + n];
                out[(size_t)m * g.ldo + n] = v;
            }
    }
}

// out[b][m][n] = act(alpha * sum_s partial[b * splits + s][m][n] + bias[n]) + resid[b][m][n], s in ascending order
__global__ __launch_bounds__(256) void splitk_reduce_kernel(SgemmArgs g, int batch) {
    const size_t mn = (size_t)g.M * g.N, total = mn * batch;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int b = (int)(i / mn);
    const size_t r = i - (size_t)b * mn;
    const int m = (int)(r / g.N), n = (int)(r - (size_t)m * g.N);
    const float* p = g.partial + (size_t)b * g.splits * mn + r;
    float v = 0.f;
    for (int s = 0; s < g.splits; ++s) v += p[(size_t)s * mn];
    v = v * g.alpha + (g.bias ? g.bias[n] : 0.f);
    v = act_apply(v, g.act);
    if (g.resid) v += g.resid[(size_t)b * g.sR + (size_t)m * g.ldr + n];
    g.out[(size_t)b * g.sO + (size_t)m * g.ldo + n] = v;
}

// rows of up to 64 * NV columns: the row lives in registers between the max, the exponentials and the scaling --
// one read and one write of the score matrix instead of three reads and two writes
template <int NV>
__global__ __launch_bounds__(256) void softmax_rows_reg_kernel(float* x, long ld, int rows, int cols) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float* p = x + (size_t)row * ld;
    float v[NV];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < cols ? p[c] : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = expf(v[i] - mx);           // exp(-inf) = 0 for the padding lanes
        s += v[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const float inv = 1.0f / s;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < cols) p[c] = v[i] * inv;
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
    ap::SgemmArgs g{A, lda, strideA, W, ldw, strideW, w_is_kn, M, N, K, alpha, bias, act, resid, ldr, strideR, out, ldo, strideO,
                    0, 1, K, nullptr};
    const bool al16 = (((uintptr_t)A | (uintptr_t)W) & 15) == 0 && lda % 4 == 0 && ldw % 4 == 0 && strideA % 4 == 0 && strideW % 4 == 0;
    g.vec = al16 && (w_is_kn || K % 4 == 0) ? 1 : 0;
    // tile: 128 x 128 only when that still gives every CU two workgroups; else 64 x 64
    auto wgs = [&](int t) { return (long)((M + t - 1) / t) * ((N + t - 1) / t) * batch; };
    const int T = (M > 64 && N > 64 && wgs(128) >= 512) ? 128 : 64;
    // split-K: few output tiles over a long K (P V of the global / token-to-image attention, the late MLPs) would leave
    // most CUs idle behind serial K loops -> K chunks of >= 128 on separate workgroups, partial sums reduced in order
    int splits = 1;
    if (wgs(T) < 256 && K >= 512) {
        splits = (int)std::min<long>(std::min<long>((256 + wgs(T) - 1) / wgs(T), K / 128), 32);
        if (splits < 2) splits = 1;
    }
    hipStream_t s = (hipStream_t)stream;
    if (splits > 1) {
        g.k_chunk = (int)ap::align_up((size_t)(K + splits - 1) / splits, 16);
        splits = (K + g.k_chunk - 1) / g.k_chunk;
        g.splits = splits;
        const size_t need = (size_t)splits * batch * M * N * sizeof(float);
        int dev = 0;
        AP_HIP_CHECK(hipGetDevice(&dev));
        static std::mutex mu;
        static std::map<int, std::pair<float*, size_t>> scratch;         // per device, grow-only
        std::lock_guard<std::mutex> lock(mu);
        auto& sc = scratch[dev];
        if (sc.second < need) {
            // An outgrown buffer is kept alive (captured graphs may still launch kernels that point at it); sizes double,
            // so the total stays below twice the largest request.  hipMalloc is not legal inside a stream capture: callers
            // that capture (services/sam2_hip.py) run every shape once before capturing.
            const size_t bytes = std::max(need, sc.second * 2);
            float* fresh = nullptr;
            AP_HIP_CHECK(hipMalloc((void**)&fresh, bytes));
            sc = {fresh, bytes};
        }
        g.partial = sc.first;
    }
    const int gy = (M + T - 1) / T;
    AP_REQUIRE(gy <= 65535 && (long)batch * splits <= 65535, "ap_sgemm: problem %d x %d x %d x %d too large", batch, M, N, K);
    dim3 grid((N + T - 1) / T, gy, batch * splits);
    if (T == 128) {
        if (w_is_kn) ap::sgemm_mfma_kernel<128, true><<<grid, 256, 0, s>>>(g);
        else ap::sgemm_mfma_kernel<128, false><<<grid, 256, 0, s>>>(g);
    } else {
        if (w_is_kn) ap::sgemm_mfma_kernel<64, true><<<grid, 256, 0, s>>>(g);
        else ap::sgemm_mfma_kernel<64, false><<<grid, 256, 0, s>>>(g);
    }
    if (splits > 1)
        ap::splitk_reduce_kernel<<<ap::grid1((size_t)batch * M * N), 256, 0, s>>>(g, batch);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int ap_softmax_rows(float* x, long ld, int rows, int cols, ap_stream_t stream) {
    AP_REQUIRE(x && rows > 0 && cols > 0, "ap_softmax_rows: bad arguments");
    const dim3 grid((rows + 3) / 4);
    hipStream_t s = (hipStream_t)stream;
    if (cols <= 64) ap::softmax_rows_reg_kernel<1><<<grid, 256, 0, s>>>(x, ld, rows, cols);
    else if (cols <= 256) ap::softmax_rows_reg_kernel<4><<<grid, 256, 0, s>>>(x, ld, rows, cols);
    else if (cols <= 1024) ap::softmax_rows_reg_kernel<16><<<grid, 256, 0, s>>>(x, ld, rows, cols);
    else if (cols <= 4096) ap::softmax_rows_reg_kernel<64><<<grid, 256, 0, s>>>(x, ld, rows, cols);
    else ap::softmax_rows_kernel<<<grid, 256, 0, s>>>(x, ld, rows, cols);
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
