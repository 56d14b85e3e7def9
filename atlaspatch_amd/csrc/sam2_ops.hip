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
#include <type_traits>
#include <cstdlib>
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
//   T = 64: one block per wave for the small batched attention products), 32-deep K tiles.
//   Global -> registers (float4 per thread, next K tile in flight under the MFMAs) -> LDS (two buffers, one barrier per
//   K tile).  LDS rows hold [row][32 k] padded to 36 floats: a lane reads its 16 k values with four ds_read_b128 and rows
//   144 B apart put every lane group of a ds_read_b128 on 64 distinct banks.  The MFMA's two k slots are fed k = e and
//   k = 16 + e (lane halves), the same mapping on both operands, so the sum runs over all 32 k of the tile.  The NN weight
//   ([k][n], P V products) keeps its [k][n] layout in LDS and is read with 16 conflict-free ds_read_b32.
//   D[i = m][j = n]: a lane owns column n = lane % 32 and 16 rows -> stores / residual reads are 128-byte row segments.
template <int TM, int TN, bool WKN, bool VEC>
__global__ __launch_bounds__(256) void sgemm_mfma_kernel(SgemmArgs g) {
    // KT-deep K tiles: one barrier per 16 MFMA steps and whole 128-byte lines per operand row.  Measured
    // (tools/archive/sgemm_probe.py): SAM2's mid-size shapes run at 42 - 49 % of the 157 TF/s f32 MFMA peak with either tile size;
    // large problems reach 67 % (64 x 64) / 77 % (128 x 128).  Tried without effect on the mid sizes: two accumulators per
    // wave instead of one dependent chain (+-0), 16- vs 32-deep K tiles (+3 %), a 128 x 64 tile (-19 %: occupancy), loads
    // hoisted out of branches (+6 %).  MFMA busy 51 % (PMC) with LDS 29 % busy: what is left is the short K loop itself --
    // 3 to 12 K tiles per workgroup leave prologue / epilogue and the ramp of co-resident workgroups a large share.
    constexpr int KT = 32, Q = KT / 4, HALF = KT / 2;
    constexpr int BM = TM / 64, BN = TN / 64, LDR = KT + 4, LDN = TN + 4;
    constexpr int PA = TM * Q / 256, PW = WKN ? KT * (TN / 4) / 256 : TN * Q / 256;      // float4 loads per thread and K tile
    __shared__ __attribute__((aligned(16))) float As[2][TM * LDR];
    __shared__ __attribute__((aligned(16))) float Ws[2][WKN ? KT * LDN : TN * LDR];
    const int b = blockIdx.z / g.splits, split = blockIdx.z % g.splits;
    const float* A = g.A + (size_t)b * g.sA;
    const float* W = g.W + (size_t)b * g.sW;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 1, wn = wave >> 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int kbeg = split * g.k_chunk, kend = (kbeg + g.k_chunk < g.K) ? kbeg + g.k_chunk : g.K;

    // Every global load is UNCONDITIONAL (addresses clamped into the operand, validity kept as a 4-bit mask and applied
    // when the registers go to LDS): a load behind a branch makes hipcc wait for it at the join, i.e. before the MFMAs
    // it was meant to run under.
    f32x4 ra[PA], rw[PW];
    unsigned ma[PA], mw[PW];
    auto load_rows = [&](const float* base, long ld, int row0, int rows, int k0, f32x4* dst, unsigned* msk, auto per) {   // [row][k]
#pragma unroll
        for (int j = 0; j < decltype(per)::value; ++j) {
            const int f = tid + 256 * j, r = f / Q, k = k0 + (f % Q) * 4;
            int gr = row0 + r;
            gr = gr < rows ? gr : rows - 1;
            const float* p = base + (size_t)gr * ld;
            if constexpr (VEC) {                      // K % 4 == 0: a float4 is wholly inside or wholly outside [kbeg, kend)
                const bool ok = k < kend;
                dst[j] = *(const f32x4*)(p + (ok ? k : kbeg));
                msk[j] = ok ? 15u : 0u;
            } else {
                unsigned m = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool ok = k + e < kend;
                    dst[j][e] = p[ok ? k + e : kend - 1];
                    m |= (ok ? 1u : 0u) << e;
                }
                msk[j] = m;
            }
        }
    };
    auto load_kn = [&](int k0, f32x4* dst, unsigned* msk) {                                                     // W[k][n]
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const int f = tid + 256 * j, kr = f / (TN / 4), n = n0 + (f % (TN / 4)) * 4, k = k0 + kr;
            const bool kok = k < kend;
            const float* p = W + (size_t)(kok ? k : kend - 1) * g.ldw;
            if (VEC && n + 3 < g.N) {                 // uniform per (thread, launch): N and n0 do not change in the K loop
                dst[j] = *(const f32x4*)(p + n);
                msk[j] = kok ? 15u : 0u;
            } else {
                unsigned m = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool ok = kok && n + e < g.N;
                    dst[j][e] = p[n + e < g.N ? n + e : g.N - 1];
                    m |= (ok ? 1u : 0u) << e;
                }
                msk[j] = m;
            }
        }
    };
    auto masked = [](f32x4 v, unsigned m) {
        return f32x4{(m & 1u) ? v[0] : 0.f, (m & 2u) ? v[1] : 0.f, (m & 4u) ? v[2] : 0.f, (m & 8u) ? v[3] : 0.f};
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const int f = tid + 256 * j;
            *(f32x4*)&As[buf][(f / Q) * LDR + (f % Q) * 4] = masked(ra[j], ma[j]);
        }
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const int f = tid + 256 * j;
            if constexpr (WKN) *(f32x4*)&Ws[buf][(f / (TN / 4)) * LDN + (f % (TN / 4)) * 4] = masked(rw[j], mw[j]);
            else *(f32x4*)&Ws[buf][(f / Q) * LDR + (f % Q) * 4] = masked(rw[j], mw[j]);
        }
    };
    auto load_tile = [&](int k0) {
        load_rows(A, g.lda, m0, g.M, k0, ra, ma, std::integral_constant<int, PA>{});
        if constexpr (WKN) load_kn(k0, rw, mw);
        else load_rows(W, g.ldw, n0, g.N, k0, rw, mw, std::integral_constant<int, PW>{});
    };

    f32x16 acc[BM][BN];
#pragma unroll
    for (int i = 0; i < BM; ++i)
#pragma unroll
        for (int j = 0; j < BN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (kend - kbeg + KT - 1) / KT;
    load_tile(kbeg);
    store_lds(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(kbeg + (kt + 1) * KT);
        // the MFMA's two k slots take k = e (lanes 0-31) and k = HALF + e (lanes 32-63), the same mapping on both operands
        float a[BM][HALF], w[BN][HALF];
#pragma unroll
        for (int i = 0; i < BM; ++i) {
            const float* pa = &As[cur][(wm * (TM / 2) + i * 32 + l31) * LDR + hi * HALF];
#pragma unroll
            for (int q = 0; q < HALF / 4; ++q) {
                const f32x4 v = *(const f32x4*)(pa + q * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) a[i][q * 4 + e] = v[e];
            }
        }
#pragma unroll
        for (int i = 0; i < BN; ++i) {
            if constexpr (WKN) {
#pragma unroll
                for (int e = 0; e < HALF; ++e) w[i][e] = Ws[cur][(hi * HALF + e) * LDN + wn * (TN / 2) + i * 32 + l31];
            } else {
                const float* pw = &Ws[cur][(wn * (TN / 2) + i * 32 + l31) * LDR + hi * HALF];
#pragma unroll
                for (int q = 0; q < HALF / 4; ++q) {
                    const f32x4 v = *(const f32x4*)(pw + q * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[i][q * 4 + e] = v[e];
                }
            }
        }
#pragma unroll
        for (int e = 0; e < HALF; ++e)
#pragma unroll
            for (int i = 0; i < BM; ++i)
#pragma unroll
                for (int j = 0; j < BN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], w[j][e], acc[i][j], 0, 0, 0);
        if (kt + 1 < nk) store_lds(cur ^ 1);
        __syncthreads();
    }

    if (g.splits > 1) {                 // raw partial sums; alpha / bias / activation / residual in the reduce kernel
        float* part = g.partial + (size_t)blockIdx.z * g.M * g.N;
#pragma unroll
        for (int j = 0; j < BN; ++j) {
            const int n = n0 + wn * (TN / 2) + j * 32 + l31;
            if (n >= g.N) continue;
#pragma unroll
            for (int i = 0; i < BM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * (TM / 2) + i * 32 + (r >> 2) * 8 + hi * 4 + (r & 3);
                    if (m < g.M) part[(size_t)m * g.N + n] = acc[i][j][r];
                }
        }
        return;
    }
    float* out = g.out + (size_t)b * g.sO;
    const float* resid = g.resid ? g.resid + (size_t)b * g.sR : nullptr;
#pragma unroll
    for (int j = 0; j < BN; ++j) {
        const int n = n0 + wn * (TN / 2) + j * 32 + l31;
        const bool nok = n < g.N;
        const int nn = nok ? n : g.N - 1;
        const float bias = g.bias ? g.bias[nn] : 0.f;
#pragma unroll
        for (int i = 0; i < BM; ++i) {
            const int mb = m0 + wm * (TM / 2) + i * 32 + hi * 4;
            float rv[16];
            if (resid) {                              // all 16 residual loads in flight before the first is used
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r >> 2) * 8 + (r & 3);
                    rv[r] = resid[(size_t)(m < g.M ? m : g.M - 1) * g.ldr + nn];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r >> 2) * 8 + (r & 3);
                float v = acc[i][j][r] * g.alpha + bias;
                v = act_apply(v, g.act);
                if (resid) v += rv[r];
                if (nok && m < g.M) out[(size_t)m * g.ldo + n] = v;
            }
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

__global__ void window_unpartition_kernel(const float* win, const float* resid, int B, int H, int W, int C, int ws, int nwy, int nwx,
                                          float* x) {
    const size_t total = (size_t)B * H * W * C;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C); size_t r = i / C;
    const int xx = (int)(r % W); r /= W;
    const int y = (int)(r % H); const int b = (int)(r / H);
    const int wy = y / ws, iy = y % ws, wx = xx / ws, ix = xx % ws;
    const float v = win[((((size_t)(b * nwy + wy) * nwx + wx) * ws + iy) * ws + ix) * C + c];
    x[i] = resid ? resid[i] + v : v;
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
// dst[y][x] = src[yidx[y]][xidx[x]]: the mask's way back to the thumbnail's shape (PIL NEAREST resize = a gather through
// per-axis index tables; services/segmentation.py:112-118).
__global__ void gather2d_kernel(const float* __restrict__ src, int src_w, const int* __restrict__ yidx,
                                const int* __restrict__ xidx, int oh, int ow, float* __restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)oh * ow) return;
    const int y = (int)(i / ow), x = (int)(i - (size_t)y * ow);
    dst[i] = src[(size_t)yidx[y] * src_w + xidx[x]];
}

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

// stack > 1: the M rows are `stack` equally shaped problems stacked along M (a batch of images through a row-wise layer).
// The split-K plan is then the single problem's, so every output row goes through exactly the arithmetic it would go
// through alone (results independent of the batch size); only the tile size follows the whole problem.
static int sgemm_impl(const float* A, long lda, long strideA, const float* W, long ldw, long strideW, int w_is_kn,
                      int batch, int M, int N, int K, float alpha, const float* bias, int act,
                      const float* resid, long ldr, long strideR, float* out, long ldo, long strideO, int stack,
                      ap_stream_t stream) {
    AP_REQUIRE(A && W && out, "ap_sgemm: null pointer");
    AP_REQUIRE(stack >= 1 && M % stack == 0, "ap_sgemm: %d rows are not %d stacked problems", M, stack);
    AP_REQUIRE(batch > 0 && M > 0 && N > 0 && K > 0 && batch <= 65535, "ap_sgemm: bad problem %d x %d x %d x %d", batch, M, N, K);
    AP_REQUIRE(act >= 0 && act <= 2, "ap_sgemm: activation %d", act);
    ap::SgemmArgs g{A, lda, strideA, W, ldw, strideW, w_is_kn, M, N, K, alpha, bias, act, resid, ldr, strideR, out, ldo, strideO,
                    0, 1, K, nullptr};
    const bool al16 = (((uintptr_t)A | (uintptr_t)W) & 15) == 0 && lda % 4 == 0 && ldw % 4 == 0 && strideA % 4 == 0 && strideW % 4 == 0;
    g.vec = al16 && (w_is_kn || K % 4 == 0) ? 1 : 0;
    // tile (M x N): 128 x 128 when that still gives every CU two workgroups, else 64 x 64.  (A 128 x 64 tile was measured
    // too: M = 4096, N = 1536, K = 384 went from 62.9 to 74.6 us -- two workgroups per CU instead of four cost more than
    // the 25 % fewer operand bytes returned.)
    auto wgs_m = [&](int m, int tm, int tn) { return (long)((m + tm - 1) / tm) * ((N + tn - 1) / tn) * batch; };
    auto wgs = [&](int tm, int tn) { return wgs_m(M, tm, tn); };
    // split-K: few output tiles over a long K (P V of the global / token-to-image attention, the late MLPs) would leave
    // most CUs idle behind serial K loops -> K chunks of >= 128 on separate workgroups, partial sums reduced in order.
    // Planned on ONE of the stacked problems (its own tile size included), see above.
    const int M1 = M / stack;
    int TM1 = 64;
    if (M1 > 64 && N > 64 && wgs_m(M1, 128, 128) >= 512) TM1 = 128;
    int splits = 1;
    if (wgs_m(M1, TM1, TM1) < 256 && K >= 512) {
        splits = (int)std::min<long>(std::min<long>((256 + wgs_m(M1, TM1, TM1) - 1) / wgs_m(M1, TM1, TM1), K / 128), 32);
        if (splits < 2) splits = 1;
    }
    int TM = 64, TN = 64;
    if (M > 64 && N > 64 && wgs(128, 128) >= 512) { TM = 128; TN = 128; }
    hipStream_t s = (hipStream_t)stream;
    if (splits > 1) {
        g.k_chunk = (int)ap::align_up((size_t)(K + splits - 1) / splits, 32);
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
    const int gy = (M + TM - 1) / TM;
    AP_REQUIRE(gy <= 65535 && (long)batch * splits <= 65535, "ap_sgemm: problem %d x %d x %d x %d too large", batch, M, N, K);
    dim3 grid((N + TN - 1) / TN, gy, batch * splits);
#define AP_SGEMM_LAUNCH(A_, B_, KN, VV) ap::sgemm_mfma_kernel<A_, B_, KN, VV><<<grid, 256, 0, s>>>(g)
#define AP_SGEMM_TILE(A_, B_)                                                                              \
    do {                                                                                                   \
        if (w_is_kn) { if (g.vec) AP_SGEMM_LAUNCH(A_, B_, true, true); else AP_SGEMM_LAUNCH(A_, B_, true, false); }   \
        else { if (g.vec) AP_SGEMM_LAUNCH(A_, B_, false, true); else AP_SGEMM_LAUNCH(A_, B_, false, false); }         \
    } while (0)
    if (TM == 128) AP_SGEMM_TILE(128, 128);
    else AP_SGEMM_TILE(64, 64);
#undef AP_SGEMM_TILE
#undef AP_SGEMM_LAUNCH
    if (splits > 1)
        ap::splitk_reduce_kernel<<<ap::grid1((size_t)batch * M * N), 256, 0, s>>>(g, batch);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int ap_sgemm(const float* A, long lda, long strideA, const float* W, long ldw, long strideW, int w_is_kn,
             int batch, int M, int N, int K, float alpha, const float* bias, int act,
             const float* resid, long ldr, long strideR, float* out, long ldo, long strideO, ap_stream_t stream) {
    return sgemm_impl(A, lda, strideA, W, ldw, strideW, w_is_kn, batch, M, N, K, alpha, bias, act, resid, ldr, strideR, out, ldo,
                      strideO, 1, stream);
}

int ap_sgemm_stacked(const float* A, long lda, const float* W, long ldw, int stack, int M, int N, int K, const float* bias, int act,
                     const float* resid, long ldr, float* out, long ldo, ap_stream_t stream) {
    return sgemm_impl(A, lda, 0, W, ldw, 0, 0, 1, M, N, K, 1.0f, bias, act, resid, ldr, 0, out, ldo, 0, stack, stream);
}

int ap_sattention_f32(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, int batch, int heads, int tq,
                      int tk, int d, float scale, float* out, long ldo, ap_stream_t stream) {
    return ap::launch_sattention(q, ldq, k, ldk, v, ldv, batch, heads, tq, tk, d, scale, out, ldo, (hipStream_t)stream, /*exact=*/true);
}

int ap_sattention_split_f16(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, int batch, int heads, int tq,
                            int tk, int d, float scale, float* out, long ldo, ap_stream_t stream) {
    return ap::launch_sattention(q, ldq, k, ldk, v, ldv, batch, heads, tq, tk, d, scale, out, ldo, (hipStream_t)stream, /*exact=*/false);
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
        win, nullptr, b, h, w, c, ws, nwy, nwx, x);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int ap_window_unpartition_add(const float* win, const float* resid, int b, int h, int w, int c, int ws, float* x,
                              ap_stream_t stream) {
    AP_REQUIRE(x && win && resid && ws > 0, "ap_window_unpartition_add: bad arguments");
    const int nwy = (h + ws - 1) / ws, nwx = (w + ws - 1) / ws;
    ap::window_unpartition_kernel<<<ap::grid1((size_t)b * h * w * c), 256, 0, (hipStream_t)stream>>>(
        win, resid, b, h, w, c, ws, nwy, nwx, x);
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

int ap_gather2d_f32(const float* src, int src_h, int src_w, const int32_t* yidx, const int32_t* xidx, int oh, int ow, float* dst,
                    ap_stream_t stream) {
    AP_REQUIRE(src && yidx && xidx && dst && src_h > 0 && src_w > 0 && oh > 0 && ow > 0, "ap_gather2d_f32: bad arguments");
    ap::gather2d_kernel<<<ap::grid1((size_t)oh * ow), 256, 0, (hipStream_t)stream>>>(src, src_w, yidx, xidx, oh, ow, dst);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

}  // extern "C"
