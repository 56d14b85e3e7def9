// HBM-bound glue kernels of the ViT encoder: LayerNorm, CLS/pos init, dtype convert and
// CHW -> patch-row gather.  All are streaming kernels: 16-byte loads per lane, one wave
// per row where a row reduction is needed (no LDS, no block barrier).
//
// Algorithmic bytes per row: LayerNorm reads 4*dim and writes sizeof(T)*dim; with the fused residual
// add it also reads sizeof(T)*dim (delta) and writes 4*dim (the updated stream).
#include "ap_common.h"
#include <type_traits>

namespace ap {
namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <typename T> __device__ __forceinline__ void store_vec4(T* p, f32x4 v);
template <> __device__ __forceinline__ void store_vec4<float>(float* p, f32x4 v) { *(f32x4*)p = v; }
template <> __device__ __forceinline__ void store_vec4<f16>(f16* p, f32x4 v) {
    f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
    *(f16x4*)p = h;
}
template <> __device__ __forceinline__ void store_vec4<bf16>(bf16* p, f32x4 v) {
    bf16x4 h = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    *(bf16x4*)p = h;
}

template <typename T> __device__ __forceinline__ f32x4 load_vec4(const T* p);
template <> __device__ __forceinline__ f32x4 load_vec4<float>(const float* p) { return *(const f32x4*)p; }
template <> __device__ __forceinline__ f32x4 load_vec4<f16>(const f16* p) {
    const f16x4 h = *(const f16x4*)p;
    return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}
template <> __device__ __forceinline__ f32x4 load_vec4<bf16>(const bf16* p) {
    const bf16x4 h = *(const bf16x4*)p;
    return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}

// ---- (residual add +) LayerNorm.  x: f32 rows (the residual stream); when `delta` is given the
// row is first updated, x += delta * ls (the branch output a GEMM stored in TD, times the optional
// LayerScale vector ls; written back unless the launch says otherwise), then
// normalised:  out = (x - mean) * rstd * gamma + beta  in TO.  Two-pass mean / variance on the
// register-resident row, biased variance, eps inside the sqrt (nn.LayerNorm).
//
// Up to two pending branch outputs are folded in one pass (d[0] first, then d[1]: the same f32 operation order as
// folding them in two launches); `store` = 0 leaves the stream untouched (the value is only needed for this
// normalisation and will be folded again, together with the next branch, by the launch that does store).
struct LnAdds {
    const void* d[2];
    long ds[2];
    const float* ls[2];
    int store;
};

// Main kernel: 16 lanes per row, 4 rows per wave: reductions stay inside a DPP row (no LDS
// permutes) and four rows in flight per wave hide the HBM latency.  dim = 64 * NV.


__device__ __forceinline__ float row16_sum(float v) {
    // xor-1, xor-2 (quad permutes), then mirror within 8 and within 16 lanes: all VALU + DPP
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

// WIDE: a lane owns 8 consecutive elements per step (two adjacent f32x4 of the stream, ONE 16-byte load of a 16-bit
// branch output, ONE 16-byte store of the normalised row) instead of 4: half as many memory instructions on the
// 16-bit operands and 256-byte instead of 128-byte segments per 16-lane row.
template <typename T> __device__ __forceinline__ void load_vec8(const T* p, f32x4& a, f32x4& b) {
    if constexpr (sizeof(T) == 2) {
        typedef T Tx8 __attribute__((ext_vector_type(8)));
        const Tx8 h = *(const Tx8*)p;
        a = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
        b = f32x4{(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
    } else {
        a = *(const f32x4*)p;
        b = *(const f32x4*)(p + 4);
    }
}
template <typename T> __device__ __forceinline__ void store_vec8(T* p, f32x4 a, f32x4 b) {
    if constexpr (sizeof(T) == 2) {
        typedef T Tx8 __attribute__((ext_vector_type(8)));
        Tx8 h = {(T)a[0], (T)a[1], (T)a[2], (T)a[3], (T)b[0], (T)b[1], (T)b[2], (T)b[3]};
        *(Tx8*)p = h;
    } else {
        *(f32x4*)p = a;
        *(f32x4*)(p + 4) = b;
    }
}

// The per-column parameters (gamma, beta and the LayerScale vectors) are staged once per workgroup in LDS (one
// coalesced 16-byte load per thread and vector): read from global memory per row they cost 2-4 extra 16-byte loads
// per lane and vector (uni_v1's two LayerScale vectors made add+LN 26 % slower than vit_l_16's: 50.0 -> 39.7 ms).
template <typename TD, typename TO, int NV, bool WIDE>
__global__ __launch_bounds__(256) void layernorm16_kernel(float* __restrict__ x, long stride,
                                                          LnAdds add, int rows, int dim,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps,
                                                          TO* __restrict__ out, int groups) {
    extern __shared__ __attribute__((aligned(16))) float prm[];      // gamma | beta | ls0 | ls1, dim floats each
    float* sg = prm;
    float* sb = prm + dim;
    const float* sls[2] = {add.ls[0] ? prm + 2 * dim : nullptr, add.ls[1] ? prm + 3 * dim : nullptr};
    for (int i = threadIdx.x * 4; i < dim; i += 1024) {
        *(f32x4*)(sg + i) = *(const f32x4*)(gamma + i);
        *(f32x4*)(sb + i) = *(const f32x4*)(beta + i);
        if (add.ls[0]) *(f32x4*)(prm + 2 * dim + i) = *(const f32x4*)(add.ls[0] + i);
        if (add.ls[1]) *(f32x4*)(prm + 3 * dim + i) = *(const f32x4*)(add.ls[1] + i);
    }
    __syncthreads();
    const int l16 = threadIdx.x & 15;
    // element offset of the lane's i-th f32x4
    auto off = [&](int i) { return WIDE ? (((i >> 1) * 16 + l16) * 8 + (i & 1) * 4) : ((l16 + i * 16) * 4); };
    for (int rg = 0; rg < groups; ++rg) {
        const int row0 = (blockIdx.x * groups + rg) * 16;
        if (row0 >= rows) break;
        int row = row0 + (threadIdx.x >> 4);
        const bool live = row < rows;
        if (!live) row = rows - 1;                       // keep all lanes in the DPP reductions
        float* src = x + (size_t)row * stride;
        f32x4 v[NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = *(const f32x4*)(src + off(i));
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            if (!add.d[a]) continue;
            const TD* dsrc = (const TD*)add.d[a] + (size_t)row * add.ds[a];
            const float* ls = sls[a];
#pragma unroll
            for (int i = 0; i < NV; i += (WIDE ? 2 : 1)) {
                f32x4 d[2];
                if constexpr (WIDE) load_vec8<TD>(dsrc + off(i), d[0], d[1]);
                else d[0] = load_vec4<TD>(dsrc + off(i));
#pragma unroll
                for (int h = 0; h < (WIDE ? 2 : 1); ++h) {
                    if (ls) {
                        const f32x4 sc = *(const f32x4*)(ls + off(i + h));
#pragma unroll
                        for (int e = 0; e < 4; ++e) d[h][e] *= sc[e];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[i + h][e] += d[h][e];
                }
            }
        }
        if (add.store && live && (add.d[0] || add.d[1])) {
#pragma unroll
            for (int i = 0; i < NV; ++i) *(f32x4*)(src + off(i)) = v[i];
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        const float mean = row16_sum(s) / (float)dim;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[i][e] - mean;
                q += d * d;
            }
        const float rstd = 1.0f / sqrtf(row16_sum(q) / (float)dim + eps);
        if (!live) continue;
        TO* dst = out + (size_t)row * dim;
#pragma unroll
        for (int i = 0; i < NV; i += (WIDE ? 2 : 1)) {
            f32x4 y[2];
#pragma unroll
            for (int h = 0; h < (WIDE ? 2 : 1); ++h) {
                const f32x4 ga = *(const f32x4*)(sg + off(i + h));
                const f32x4 be = *(const f32x4*)(sb + off(i + h));
#pragma unroll
                for (int e = 0; e < 4; ++e) y[h][e] = (v[i + h][e] - mean) * rstd * ga[e] + be[e];
            }
            if constexpr (WIDE) store_vec8<TO>(dst + off(i), y[0], y[1]);
            else store_vec4<TO>(dst + off(i), y[0]);
        }
    }
}

// Generic kernel: one wave per row, whole row held in registers (dim <= 64 * 4 * MAXV: kMaxVec = 8 for rows up to 2048 wide,
// kMaxVecWide = 16 up to 4096 -- DINOv3 ViT-7B; the same per-lane order, so a dim served by both would give the same bits).
constexpr int kMaxVec = 8, kMaxVecWide = 16;

template <typename TD, typename TO, int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(float* __restrict__ x, long stride,
                                                        LnAdds add, int rows, int dim,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        TO* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = dim >> 2;
    f32x4* src = (f32x4*)(x + (size_t)row * stride);
    f32x4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + i * 64;
        if (idx < nvec) {
            v[i] = src[idx];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                if (!add.d[a]) continue;
                f32x4 d = load_vec4<TD>((const TD*)add.d[a] + (size_t)row * add.ds[a] + idx * 4);
                if (add.ls[a]) {
                    const f32x4 sc = ((const f32x4*)add.ls[a])[idx];
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[e] *= sc[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[i][e] += d[e];
            }
            if (add.store && (add.d[0] || add.d[1])) src[idx] = v[i];
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    const float mean = wave_sum(s) / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + i * 64;
        if (idx < nvec) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[i][e] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)dim + eps);
    TO* dst = out + (size_t)row * dim;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + i * 64;
        if (idx < nvec) {
            const f32x4 ga = ((const f32x4*)gamma)[idx];
            const f32x4 be = ((const f32x4*)beta)[idx];
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (v[i][e] - mean) * rstd * ga[e] + be[e];
            store_vec4<TO>(dst + idx * 4, y);
        }
    }
}


// ---- fused-LayerNorm path (16-bit residual stream; vit.cpp::run_blocks_fused) -------------------------------
// stream_init: the f32 token matrix the patch-embed GEMM wrote -> the T stream + the row statistics of the ROUNDED
// row (what the first fused GEMM multiplies): two-pass mean / variance like the LayerNorm kernels.  One wave per row.
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void stream_init_kernel(const float* __restrict__ tok, int rows, int dim, float eps,
                                                          T* __restrict__ x, float* __restrict__ rowstats) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = dim >> 2;
    const f32x4* src = (const f32x4*)(tok + (size_t)row * dim);
    T* dst = x + (size_t)row * dim;
    f32x4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + i * 64;
        if (idx < nvec) {
            f32x4 a = src[idx];
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = (float)from_f32<T>(a[e]);
            v[i] = a;
            store_vec4<T>(dst + idx * 4, a);
            s += (a[0] + a[1]) + (a[2] + a[3]);
        }
    }
    const float mean = wave_sum(s) / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + i * 64;
        if (idx < nvec) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[i][e] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)dim + eps);
    if (lane == 0) {
        rowstats[2 * (size_t)row] = rstd;
        rowstats[2 * (size_t)row + 1] = -mean * rstd;
    }
}

// partial (sum, sum of squares) per 64-column group, written by the EPI_RESID_STATS epilogue -> (rstd, -mean * rstd).
// Eight lanes per row, one 16-byte load (two groups) per lane: a wave reads 8 rows x (groups * 8) contiguous bytes
// (a thread per row with a 96-byte stride measured 36 us per call at 403 456 rows).  The lane sums are combined in a
// fixed butterfly order (deterministic), the mean / variance arithmetic is done in double.
// exact class rows (vit.cpp), folded into the same launch: workgroups past the row blocks each take ONE image's class row --
// cls32[img] += branch[img] (f32), the stream row img * tokens becomes its rounding to the compute type, and the row's statistics
// are written directly (fixed order: a thread's columns ascending, wave butterfly, waves 0..3) -- and the row blocks leave the
// rows img * tokens alone.  cls.tokens = 0: plain finalisation.
struct ClsExact { float* cls32; const float* branch; void* x; int tokens; int n; int dtype; };

__global__ __launch_bounds__(256) void rowstats_finalize_kernel(const float* __restrict__ partial, int rows, int groups, int dim,
                                                                float eps, float* __restrict__ rowstats, int row_blocks, ClsExact cls) {
    if ((int)blockIdx.x >= row_blocks) {
        __shared__ float red[8];
        const int img = blockIdx.x - row_blocks;
        const size_t row = (size_t)img * cls.tokens;
        float s = 0.f, q = 0.f;
        for (int col = threadIdx.x; col < dim; col += 256) {
            const size_t at = (size_t)img * dim + col;
            const float c = cls.cls32[at] + cls.branch[at];
            cls.cls32[at] = c;
            float f;
            if (cls.dtype == AP_F16) { const f16 v = (f16)c; ((f16*)cls.x)[row * dim + col] = v; f = (float)v; }
            else { const bf16 v = (bf16)c; ((bf16*)cls.x)[row * dim + col] = v; f = (float)v; }
            s += f;
            q += f * f;
        }
        s = wave_sum(s);
        q = wave_sum(q);
        if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s; red[4 + (threadIdx.x >> 6)] = q; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const double ss = ((double)red[0] + red[1]) + ((double)red[2] + red[3]);
            const double qq = ((double)red[4] + red[5]) + ((double)red[6] + red[7]);
            const double mean = ss / dim;
            double var = qq / dim - mean * mean;
            var = var > 0.0 ? var : 0.0;
            const double rstd = 1.0 / sqrt(var + (double)eps);
            rowstats[2 * row] = (float)rstd;
            rowstats[2 * row + 1] = (float)(-mean * rstd);
        }
        return;
    }
    const int l8 = threadIdx.x & 7;
    int row = blockIdx.x * 32 + (threadIdx.x >> 3);
    bool live = row < rows;
    if (!live) row = rows - 1;                                   // keep all lanes in the DPP reductions
    if (cls.tokens > 0 && row % cls.tokens == 0) live = false;   // a class row: the workgroup of its image writes its statistics
    const f32x4* p = (const f32x4*)(partial + (size_t)row * groups * 2);
    float s = 0.f, q = 0.f;
    for (int g2 = l8; g2 * 2 < groups; g2 += 8) {                // groups is even (dim % 128 == 0)
        const f32x4 v = p[g2];
        s += v[0] + v[2];
        q += v[1] + v[3];
    }
#define AP_DPP8(V, CTRL) __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (V)), (CTRL), 0xF, 0xF, true))
    s += AP_DPP8(s, 0xB1); q += AP_DPP8(q, 0xB1);
    s += AP_DPP8(s, 0x4E); q += AP_DPP8(q, 0x4E);
    s += AP_DPP8(s, 0x141); q += AP_DPP8(q, 0x141);
#undef AP_DPP8
    if (!live || l8 != 0) return;
    const double mean = (double)s / dim;
    double var = (double)q / dim - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    rowstats[2 * (size_t)row] = (float)rstd;
    rowstats[2 * (size_t)row + 1] = (float)(-mean * rstd);
}

// class-token rows of the fused path: x[img * tokens][:] = T(cls + pos[0]) and the row's partial sums per 64-column group
// (one wave per (image, group): lane = column).  Only this kernel ever produces these rows, so its summation order is theirs.
template <typename T>
__global__ __launch_bounds__(64) void cls_stream_kernel(const float* __restrict__ prefix, int prefix_rows, int img_rows, int tokens,
                                                        int dim, T* __restrict__ x, float* __restrict__ partial) {
    const int img = blockIdx.x / prefix_rows, j = blockIdx.x - img * prefix_rows, grp = blockIdx.y, col = grp * 64 + threadIdx.x;
    const size_t row = (size_t)img * tokens + j;
    const T v = from_f32<T>(prefix[((size_t)img * img_rows + j) * dim + col]);
    x[row * dim + col] = v;
    const float f = (float)v;
    const float s = wave_sum(f), q = wave_sum(f * f);
    if (threadIdx.x == 0) {
        partial[(row * (dim >> 6) + grp) * 2] = s;
        partial[(row * (dim >> 6) + grp) * 2 + 1] = q;
    }
}

// exact class rows (vit.cpp): cls32[img] += branch[img] (f32, the unrounded branch a RESID_STATS launch left for row img * tokens),
// then the stream's class row becomes T(cls32[img]) with its partial sums -- cls_stream_kernel's order (one wave per (image, group))
template <typename T>
__global__ __launch_bounds__(256) void cls_exact_update_kernel(float* __restrict__ cls32, const float* __restrict__ branch, int tokens,
                                                               int dim, T* __restrict__ x, float* __restrict__ partial) {
    const int img = blockIdx.x, grp = blockIdx.y * 4 + (threadIdx.x >> 6), col = grp * 64 + (threadIdx.x & 63);
    if (col >= dim) return;                                       // whole waves: dim % 64 == 0
    const size_t row = (size_t)img * tokens, at = (size_t)img * dim + col;
    const float c = cls32[at] + branch[at];
    cls32[at] = c;
    const T v = from_f32<T>(c);
    x[row * dim + col] = v;
    const float f = (float)v;
    const float s = wave_sum(f), q = wave_sum(f * f);
    if ((threadIdx.x & 63) == 0) {
        partial[(row * (dim >> 6) + grp) * 2] = s;
        partial[(row * (dim >> 6) + grp) * 2 + 1] = q;
    }
}

template <typename T>
__global__ void stream_to_f32_kernel(const T* __restrict__ x, long stride, int rows, int dim, float* __restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per 4 elements
    const int per = dim >> 2;
    if (i >= (size_t)rows * per) return;
    const int row = (int)(i / per), c = (int)(i - (size_t)row * per) * 4;
    *(f32x4*)(dst + (size_t)row * dim + c) = load_vec4<T>(x + (size_t)row * stride + c);
}

__device__ __forceinline__ float block_sum256(float v, float* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return r;
}

// one 256-thread workgroup per weight row n
template <typename T>
__global__ __launch_bounds__(256) void fold_ln_kernel(const float* __restrict__ w32, int cols, int ld,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ bias_in, T* __restrict__ wout,
                                                      float* __restrict__ colsum, float* __restrict__ bias_out, int swiglu_h) {
    __shared__ float red[4];
    const int n = blockIdx.x;
    // EPI_NORM_SWIGLU's row order: output row n = 64 q + 32 half + j  <-  fc1 row half * H + 32 q + j
    const int sn = swiglu_h > 0 ? ((n & 63) < 32 ? 0 : swiglu_h) + 32 * (n >> 6) + (n & 31) : n;
    const float* src = w32 + (size_t)sn * ld;
    T* dst = wout + (size_t)n * ld;
    float cs = 0.f, bs = 0.f;
    for (int k = threadIdx.x; k < ld; k += 256) {
        float wf = 0.f;
        if (k < cols) {
            const float w = src[k];
            wf = (float)from_f32<T>(w * gamma[k]);
            bs += w * beta[k];
        }
        dst[k] = from_f32<T>(wf);
        cs += wf;
    }
    cs = block_sum256(cs, red);
    bs = block_sum256(bs, red);
    if (threadIdx.x == 0) {
        colsum[n] = cs;
        bias_out[n] = bias_in[sn] + bs;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void fold_ls_kernel(const float* __restrict__ w32, int cols, int ld,
                                                      const float* __restrict__ ls, const float* __restrict__ bias_in,
                                                      T* __restrict__ wout, float* __restrict__ bias_out) {
    const int n = blockIdx.x;
    const float sc = ls ? ls[n] : 1.0f;
    const float* src = w32 + (size_t)n * ld;
    T* dst = wout + (size_t)n * ld;
    for (int k = threadIdx.x; k < ld; k += 256) dst[k] = from_f32<T>(k < cols ? src[k] * sc : 0.f);
    if (threadIdx.x == 0) bias_out[n] = bias_in[n] * sc;
}

// Narrow rows in float32 (the SAM2 Hiera-T trunk: dims 96 / 192 / 384, neck / decoder 256; no pending branch to add): LPR
// lanes per row, NV float4 per lane (dim = LPR * NV * 4), 64 / LPR rows per wave -- every lane works and a wave's loads cover
// whole rows (the one-wave-per-row kernel below leaves 40 of 64 lanes idle at dim 96).  Two-pass mean / variance on the
// register-resident row like the other LayerNorm kernels.
template <int LPR, int NV>
__global__ __launch_bounds__(256) void layernorm_narrow_kernel(const float* __restrict__ x, long stride, int rows,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float eps, float* __restrict__ out) {
    constexpr int kDim = LPR * NV * 4, kRowsPerWave = 64 / LPR;
    const int lane = threadIdx.x & 63, l = lane % LPR;
    const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * kRowsPerWave + lane / LPR;
    const bool live = row < rows;
    const long r = live ? row : rows - 1;                       // idle lanes re-read the last row: the shuffles need every lane
    const f32x4* src = (const f32x4*)(x + r * stride);
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = src[l + i * LPR];
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
#pragma unroll
    for (int m = 1; m < LPR; m <<= 1) s += __shfl_xor(s, m, 64);
    const float mean = s / (float)kDim;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = v[i][e] - mean;
            q += d * d;
        }
#pragma unroll
    for (int m = 1; m < LPR; m <<= 1) q += __shfl_xor(q, m, 64);
    const float rstd = 1.0f / sqrtf(q / (float)kDim + eps);
    if (!live) return;
    f32x4* dst = (f32x4*)(out + row * kDim);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const f32x4 ga = ((const f32x4*)gamma)[l + i * LPR];
        const f32x4 be = ((const f32x4*)beta)[l + i * LPR];
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = (v[i][e] - mean) * rstd * ga[e] + be[e];
        dst[l + i * LPR] = y;
    }
}

template <typename TD, typename TO>
int launch_ln_typed(float* x, long stride, const LnAdds& add, int rows, int dim,
                    const float* gamma, const float* beta, float eps, void* out, hipStream_t stream) {
    TO* o = (TO*)out;
    if (dim == 768 || dim == 1024) {      // same kernel for any row count: results never depend on the batch size
        // 16 rows per workgroup: more row groups per workgroup were measured slower (2: +4 %, 8: +13 % add+LN time) --
        // the gain is the cooperative, coalesced parameter load, not its amortisation
        const int groups = 1;
        dim3 g16((rows + 16 * groups - 1) / (16 * groups)), b16(256);
        const size_t lds = (size_t)4 * dim * sizeof(float);
        // WIDE measured against the 4-element layout inside bench.py: add+LayerNorm 5.38 -> 5.64 TB/s; nontemporal
        // loads / stores of the stream were also tried: slower (5.25 TB/s)
        if (dim == 768) layernorm16_kernel<TD, TO, 12, true><<<g16, b16, lds, stream>>>(x, stride, add, rows, dim, gamma, beta, eps, o, groups);
        else layernorm16_kernel<TD, TO, 16, true><<<g16, b16, lds, stream>>>(x, stride, add, rows, dim, gamma, beta, eps, o, groups);
    } else {
        if constexpr (std::is_same<TD, float>::value && std::is_same<TO, float>::value) {
            // the narrow kernel moves 16 bytes per lane: any pointer or row stride the C ABI hands over that is not
            // 16-byte aligned takes the general kernel (equal to f32 rounding; the SAM2 path, the only caller at these widths, is always aligned)
            const bool vec_ok = ((((uintptr_t)x | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0) && stride % 4 == 0;
            if (!add.d[0] && !add.d[1] && vec_ok) {
#define AP_LN_NARROW(DIM, LPR, NV)                                                                                       \
                if (dim == DIM) {                                                                                        \
                    const int per_wg = 4 * (64 / LPR);                                                                   \
                    layernorm_narrow_kernel<LPR, NV><<<(rows + per_wg - 1) / per_wg, 256, 0, stream>>>(                  \
                        x, stride, rows, gamma, beta, eps, (float*)out);                                                 \
                    AP_HIP_CHECK(hipGetLastError());                                                                     \
                    return AP_OK;                                                                                        \
                }
                AP_LN_NARROW(96, 8, 3)
                AP_LN_NARROW(192, 16, 3)
                AP_LN_NARROW(384, 32, 3)
                AP_LN_NARROW(256, 16, 4)
                AP_LN_NARROW(128, 8, 4)
#undef AP_LN_NARROW
            }
        }
        dim3 grid((rows + 3) / 4), block(256);
        if (dim <= 64 * 4 * kMaxVec) layernorm_kernel<TD, TO, kMaxVec><<<grid, block, 0, stream>>>(x, stride, add, rows, dim, gamma, beta, eps, o);
        else layernorm_kernel<TD, TO, kMaxVecWide><<<grid, block, 0, stream>>>(x, stride, add, rows, dim, gamma, beta, eps, o);
    }
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

__global__ void cls_init_kernel(float* tok, const float* prefix, int prefix_rows, int n, int tokens, int dim) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = prefix_rows * dim;
    if (i >= n * per) return;
    const int img = i / per, r = i - img * per;             // r = j * dim + d
    tok[(size_t)img * tokens * dim + r] = prefix[r];
}

__global__ void prefix_build_kernel(const float* cls, const float* reg, int reg_rows, const float* pos, int dim, float* prefix) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1 + reg_rows) * dim) return;
    const float t = i < dim ? cls[i] : reg[i - dim];
    prefix[i] = pos ? t + pos[i] : t;
}

// timm SwiGLUPacked (GluMlp, gate_last = False): x1, x2 = fc1(x).chunk(2, -1); silu(x1) * x2.  f32 math on the T values,
// one rounding; 8 elements per lane (16-byte loads / stores)
template <typename T>
__global__ __launch_bounds__(256) void swiglu_kernel(const T* __restrict__ x, int rows, int h, T* __restrict__ out) {
    const size_t per = (size_t)(h >> 3);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)rows * per) return;
    const size_t row = i / per, c = (i - row * per) * 8;
    const T* p = x + row * (size_t)(2 * h) + c;
    const u32x4 a4 = *(const u32x4*)p, b4 = *(const u32x4*)(p + h);
    const T* a = (const T*)&a4; const T* b = (const T*)&b4;
    T y[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x1 = (float)a[e], x2 = (float)b[e];
        y[e] = from_f32<T>((x1 / (1.0f + expf(-x1))) * x2);
    }
    *(u32x4*)(out + row * (size_t)h + c) = *(const u32x4*)y;
}
__global__ __launch_bounds__(256) void swiglu_f32_kernel(const float* __restrict__ x, int rows, int h, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)rows * h) return;
    const size_t row = i / h, c = i - row * h;
    const float x1 = x[row * (size_t)(2 * h) + c], x2 = x[row * (size_t)(2 * h) + h + c];
    out[i] = (x1 / (1.0f + expf(-x1))) * x2;
}

template <typename T>
__global__ void convert_kernel(const float* __restrict__ src, T* __restrict__ dst, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < count; i += step) dst[i] = from_f32<T>(src[i]);
}

// f32 [count] (count % 32 == 0: rows of a K padded to 64) -> per 32 values 32 f16 hi followed by 32 f16 lo, hi = f16(w),
// lo = f16((w - hi) * 2^11): the weight rows of the split-f16 GEMM (gemm.hip), same bytes as the f32 row
__global__ void split_f16_kernel(const float* __restrict__ src, f16* __restrict__ dst, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < count; i += step) {
        const float w = src[i];
        const f16 h = (f16)w;
        const size_t base = (i >> 5) * 64 + (i & 31);
        dst[base] = h;
        dst[base + 32] = (f16)((w - (float)h) * 2048.0f);
    }
}

// [n, 3, S, S] -> rows [(img*g + py)*g + px][c*ps*ps + ky*ps + kx]; one thread per 4 kx
template <typename TI, typename TO>
__global__ void chw_to_patchrows_kernel(const TI* __restrict__ x, int n, int S, int ps,
                                        TO* __restrict__ dst, int ld) {
    const int g = S / ps;
    const int quads = ps >> 2;
    const size_t total = (size_t)n * 3 * S * g * quads;     // (img, c, y, px, quad)
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int quad = i % quads; size_t r = i / quads;
    const int px = r % g; r /= g;
    const int y = r % S; r /= S;
    const int c = r % 3; const int img = r / 3;
    const int py = y / ps, ky = y - py * ps;
    const TI* s = x + (((size_t)img * 3 + c) * S + y) * S + px * ps + quad * 4;
    TO* d = dst + ((size_t)(img * g + py) * g + px) * ld + (c * ps + ky) * ps + quad * 4;
    f32x4 v = {(float)s[0], (float)s[1], (float)s[2], (float)s[3]};
    store_vec4<TO>(d, v);
}
// any patch size (14: rows of 14 elements are not 8- or 16-byte aligned): one thread per element
template <typename TI, typename TO>
__global__ void chw_to_patchrows_any_kernel(const TI* __restrict__ x, int n, int S, int ps, TO* __restrict__ dst, int ld) {
    const int g = S / ps;
    const size_t total = (size_t)n * 3 * S * S;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int xx = i % S; size_t r = i / S;
    const int y = r % S; r /= S;
    const int c = r % 3; const int img = r / 3;
    const int py = y / ps, ky = y - py * ps, px = xx / ps, kx = xx - px * ps;
    dst[((size_t)(img * g + py) * g + px) * ld + (c * ps + ky) * ps + kx] = from_f32<TO>((float)x[i]);
}

// DINOv3 rotary embedding, in place on q / k of the packed qkv (see ap_common.h).  One thread = 8 consecutive channels j .. j + 7 of
// the first half of one head together with their partners j + h .. in the second half (two 16-byte accesses each way for the
// 16-bit types); unit index = ((row * parts + part) * heads + head) * (h / 8) + chunk.
template <typename T> struct Rope8 { };
template <> struct Rope8<float> {
    static __device__ __forceinline__ void load(const float* p, float* v) {
        const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
    }
    static __device__ __forceinline__ void store(float* p, const float* v) {
        *(f32x4*)p = f32x4{v[0], v[1], v[2], v[3]};
        *(f32x4*)(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
};
template <> struct Rope8<f16> {
    static __device__ __forceinline__ void load(const f16* p, float* v) {
        const f16x8 a = *(const f16x8*)p;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)a[e];
    }
    static __device__ __forceinline__ void store(f16* p, const float* v) {
        f16x8 a;
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = (f16)v[e];
        *(f16x8*)p = a;
    }
};
template <> struct Rope8<bf16> {
    static __device__ __forceinline__ void load(const bf16* p, float* v) {
        const bf16x8 a = *(const bf16x8*)p;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)a[e];
    }
    static __device__ __forceinline__ void store(bf16* p, const float* v) {
        bf16x8 a;
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = (bf16)v[e];
        *(bf16x8*)p = a;
    }
};

template <typename T>
__global__ __launch_bounds__(256) void rope_kernel(T* __restrict__ qkv, long units, int tokens, int prefix, int heads, int hd,
                                                   const float* __restrict__ cosv, const float* __restrict__ sinv, int part0, int parts) {
    const long u = (long)blockIdx.x * 256 + threadIdx.x;
    if (u >= units) return;
    const int h = hd >> 1, cpr = h >> 3;                     // chunks of 8 per half head
    const int chunk = (int)(u % cpr);
    long r = u / cpr;
    const int head = (int)(r % heads); r /= heads;
    const int part = part0 + (int)(r % parts); r /= parts;   // 0 = q, 1 = k
    const int patches = tokens - prefix;
    const long img = r / patches;
    const int pidx = (int)(r % patches);
    const long row = img * tokens + prefix + pidx;
    T* x = qkv + row * (long)(3 * heads * hd) + (long)part * heads * hd + head * hd + chunk * 8;
    const float* c = cosv + (long)pidx * hd + chunk * 8;
    const float* s = sinv + (long)pidx * hd + chunk * 8;
    float a[8], b[8], o1[8], o2[8];
    Rope8<T>::load(x, a);
    Rope8<T>::load(x + h, b);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        o1[e] = a[e] * c[e] - b[e] * s[e];
        o2[e] = b[e] * c[h + e] + a[e] * s[h + e];
    }
    Rope8<T>::store(x, o1);
    Rope8<T>::store(x + h, o2);
}

// AP_POOL_CLS_MEAN (midnight.py:58-61, virchow.py:58-61): one thread per (image, channel); the patch rows are summed in row
// order in double (deterministic, coalesced across the channel threads) and divided once
__global__ __launch_bounds__(256) void cls_mean_pool_kernel(const float* __restrict__ y, int tokens, int prefix, int dim,
                                                            float* __restrict__ out) {
    const int d = blockIdx.x * 256 + threadIdx.x, img = blockIdx.y;
    if (d >= dim) return;
    const float* base = y + (size_t)img * tokens * dim + d;
    double acc = 0.0;
    for (int t = prefix; t < tokens; ++t) acc += (double)base[(size_t)t * dim];
    float* o = out + (size_t)img * 2 * dim;
    o[d] = base[0];
    o[dim + d] = (float)(acc / (double)(tokens - prefix));
}

}  // namespace

int launch_add2_layernorm(int delta_dtype, int out_dtype, float* x, long stride, const void* delta0, long dstride0,
                          const float* ls0, const void* delta1, long dstride1, const float* ls1, int store,
                          int rows, int dim, const float* gamma, const float* beta, float eps, void* out,
                          hipStream_t stream) {
    AP_REQUIRE(dim % 4 == 0 && dim <= 64 * 4 * kMaxVecWide, "layernorm: unsupported dim %d", dim);
    AP_REQUIRE(stride % 4 == 0 && (!delta0 || dstride0 % 4 == 0) && (!delta1 || dstride1 % 4 == 0),
               "layernorm: row strides must be multiples of 4");
    if (rows <= 0) return AP_OK;
    if (!delta0 && !delta1) delta_dtype = out_dtype;
    const LnAdds add{{delta0, delta1}, {dstride0, dstride1}, {ls0, ls1}, store};
#define AP_LN(TD, TO) return launch_ln_typed<TD, TO>(x, stride, add, rows, dim, gamma, beta, eps, out, stream)
    if (delta_dtype == AP_F16 && out_dtype == AP_F16) AP_LN(f16, f16);
    if (delta_dtype == AP_BF16 && out_dtype == AP_BF16) AP_LN(bf16, bf16);
    if (delta_dtype == AP_F32 && out_dtype == AP_F32) AP_LN(float, float);
    if (delta_dtype == AP_F16 && out_dtype == AP_F32) AP_LN(f16, float);
    if (delta_dtype == AP_BF16 && out_dtype == AP_F32) AP_LN(bf16, float);
#undef AP_LN
    set_error("layernorm: unsupported dtype pair (delta %d, out %d)", delta_dtype, out_dtype);
    return AP_ERR_INVALID;
}

int launch_add_layernorm(int delta_dtype, int out_dtype, float* x, long stride, const void* delta,
                         long dstride, const float* ls, int rows, int dim, const float* gamma, const float* beta,
                         float eps, void* out, hipStream_t stream) {
    return launch_add2_layernorm(delta_dtype, out_dtype, x, stride, delta, dstride, ls, nullptr, 0, nullptr, 1, rows, dim,
                                 gamma, beta, eps, out, stream);
}

int launch_layernorm(int dtype, const float* x, long stride, int rows, int dim, const float* gamma,
                     const float* beta, float eps, void* out, hipStream_t stream) {
    return launch_add_layernorm(dtype, dtype, const_cast<float*>(x), stride, nullptr, 0, nullptr, rows, dim, gamma, beta,
                                eps, out, stream);
}

int launch_layernorm_f32out(const float* x, long stride, int rows, int dim, const float* gamma,
                            const float* beta, float eps, float* out, hipStream_t stream) {
    return launch_layernorm(AP_F32, x, stride, rows, dim, gamma, beta, eps, out, stream);
}

int launch_stream_init(int dtype, const float* tok, int rows, int dim, float eps, void* x, float* rowstats,
                       hipStream_t stream) {
    AP_REQUIRE(dim % 4 == 0 && dim <= 64 * 4 * kMaxVecWide, "stream_init: unsupported dim %d", dim);
    if (rows <= 0) return AP_OK;
    dim3 grid((rows + 3) / 4), block(256);
    const bool wide = dim > 64 * 4 * kMaxVec;
    if (dtype == AP_F16 && !wide) stream_init_kernel<f16, kMaxVec><<<grid, block, 0, stream>>>(tok, rows, dim, eps, (f16*)x, rowstats);
    else if (dtype == AP_F16) stream_init_kernel<f16, kMaxVecWide><<<grid, block, 0, stream>>>(tok, rows, dim, eps, (f16*)x, rowstats);
    else if (dtype == AP_BF16 && !wide) stream_init_kernel<bf16, kMaxVec><<<grid, block, 0, stream>>>(tok, rows, dim, eps, (bf16*)x, rowstats);
    else if (dtype == AP_BF16) stream_init_kernel<bf16, kMaxVecWide><<<grid, block, 0, stream>>>(tok, rows, dim, eps, (bf16*)x, rowstats);
    else { set_error("stream_init: dtype %d (f16 / bf16 only)", dtype); return AP_ERR_INVALID; }
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int launch_cls_stream(int dtype, const float* prefix, int prefix_rows, int img_rows, int n, int tokens, int dim, void* x, float* partial,
                      hipStream_t stream) {
    AP_REQUIRE(dim % 64 == 0 && prefix_rows > 0, "cls_stream: dim %d must be a multiple of 64", dim);
    if (n <= 0) return AP_OK;
    dim3 grid(n * prefix_rows, dim / 64), block(64);
    if (dtype == AP_F16) cls_stream_kernel<f16><<<grid, block, 0, stream>>>(prefix, prefix_rows, img_rows, tokens, dim, (f16*)x, partial);
    else if (dtype == AP_BF16) cls_stream_kernel<bf16><<<grid, block, 0, stream>>>(prefix, prefix_rows, img_rows, tokens, dim, (bf16*)x, partial);
    else { set_error("cls_stream: dtype %d (f16 / bf16 only)", dtype); return AP_ERR_INVALID; }
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int launch_cls_exact_update(int dtype, float* cls32, const float* branch, int n, int tokens, int dim, void* x, float* partial,
                            hipStream_t stream) {
    AP_REQUIRE(dim % 64 == 0, "cls_exact_update: dim %d must be a multiple of 64", dim);
    if (n <= 0) return AP_OK;
    dim3 grid(n, (dim + 255) / 256), block(256);
    if (dtype == AP_F16) cls_exact_update_kernel<f16><<<grid, block, 0, stream>>>(cls32, branch, tokens, dim, (f16*)x, partial);
    else if (dtype == AP_BF16) cls_exact_update_kernel<bf16><<<grid, block, 0, stream>>>(cls32, branch, tokens, dim, (bf16*)x, partial);
    else { set_error("cls_exact_update: dtype %d (f16 / bf16 only)", dtype); return AP_ERR_INVALID; }
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int launch_rowstats_finalize(const float* partial, int rows, int groups, int dim, float eps, float* rowstats,
                             hipStream_t stream) {
    return launch_rowstats_finalize_cls(partial, rows, groups, dim, eps, rowstats, AP_F16, nullptr, nullptr, nullptr, 0, 0, stream);
}

int launch_rowstats_finalize_cls(const float* partial, int rows, int groups, int dim, float eps, float* rowstats, int dtype,
                                 float* cls32, const float* branch, void* x, int n, int tokens, hipStream_t stream) {
    if (rows <= 0) return AP_OK;
    AP_REQUIRE(groups > 0 && groups % 2 == 0, "rowstats_finalize: groups %d must be even", groups);
    AP_REQUIRE(!cls32 || ((dtype == AP_F16 || dtype == AP_BF16) && branch && x && tokens > 0 && n * (long)tokens == rows),
               "rowstats_finalize: exact class rows need f16 / bf16, the branch buffer, the stream and rows == n * tokens");
    const int row_blocks = (rows + 31) / 32;
    const ClsExact cls{cls32, branch, x, cls32 ? tokens : 0, cls32 ? n : 0, dtype};
    rowstats_finalize_kernel<<<row_blocks + cls.n, 256, 0, stream>>>(partial, rows, groups, dim, eps, rowstats, row_blocks, cls);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int launch_rope(int dtype, void* qkv, int n, int tokens, int prefix, int heads, int head_dim, const float* cos, const float* sin,
                int which, hipStream_t stream) {
    AP_REQUIRE(head_dim % 16 == 0 && tokens > prefix && cos && sin && (which & 3), "rope: head_dim %d, %d tokens, prefix %d", head_dim, tokens, prefix);
    if (n <= 0) return AP_OK;
    const int part0 = (which & 1) ? 0 : 1, parts = (which & 3) == 3 ? 2 : 1;
    const long units = (long)n * (tokens - prefix) * parts * heads * (head_dim / 16);
    dim3 grid((unsigned)((units + 255) / 256)), block(256);
    if (dtype == AP_F16) rope_kernel<f16><<<grid, block, 0, stream>>>((f16*)qkv, units, tokens, prefix, heads, head_dim, cos, sin, part0, parts);
    else if (dtype == AP_BF16) rope_kernel<bf16><<<grid, block, 0, stream>>>((bf16*)qkv, units, tokens, prefix, heads, head_dim, cos, sin, part0, parts);
    else rope_kernel<float><<<grid, block, 0, stream>>>((float*)qkv, units, tokens, prefix, heads, head_dim, cos, sin, part0, parts);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int launch_cls_mean_pool(const float* y, int n, int tokens, int prefix, int dim, float* out, hipStream_t stream) {
    AP_REQUIRE(n >= 0 && tokens > prefix && prefix >= 1 && dim > 0, "cls_mean_pool: %d tokens, prefix %d", tokens, prefix);
    if (n == 0) return AP_OK;
    dim3 grid((unsigned)((dim + 255) / 256), (unsigned)n);
    cls_mean_pool_kernel<<<grid, 256, 0, stream>>>(y, tokens, prefix, dim, out);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int launch_stream_to_f32(int dtype, const void* x, long stride, int rows, int dim, float* dst, hipStream_t stream) {
    AP_REQUIRE(dim % 4 == 0 && stride % 4 == 0, "stream_to_f32: dim / stride must be multiples of 4");
    if (rows <= 0) return AP_OK;
    const size_t total = (size_t)rows * (dim >> 2);
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (dtype == AP_F16) stream_to_f32_kernel<f16><<<blocks, 256, 0, stream>>>((const f16*)x, stride, rows, dim, dst);
    else if (dtype == AP_BF16) stream_to_f32_kernel<bf16><<<blocks, 256, 0, stream>>>((const bf16*)x, stride, rows, dim, dst);
    else { set_error("stream_to_f32: dtype %d (f16 / bf16 only)", dtype); return AP_ERR_INVALID; }
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int launch_fold_ln(int dtype, const float* w32, int rows, int cols, int ld, const float* gamma, const float* beta,
                   const float* bias_in, void* wout, float* colsum, float* bias_out, hipStream_t stream, int swiglu_h) {
    AP_REQUIRE(swiglu_h == 0 || (rows == 2 * swiglu_h && swiglu_h % 32 == 0), "fold_ln: swiglu row order needs rows = 2 h, h %% 32 == 0");
    if (dtype == AP_F16) fold_ln_kernel<f16><<<rows, 256, 0, stream>>>(w32, cols, ld, gamma, beta, bias_in, (f16*)wout, colsum, bias_out, swiglu_h);
    else if (dtype == AP_BF16) fold_ln_kernel<bf16><<<rows, 256, 0, stream>>>(w32, cols, ld, gamma, beta, bias_in, (bf16*)wout, colsum, bias_out, swiglu_h);
    else { set_error("fold_ln: dtype %d (f16 / bf16 only)", dtype); return AP_ERR_INVALID; }
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int launch_fold_ls(int dtype, const float* w32, int rows, int cols, int ld, const float* ls, const float* bias_in,
                   void* wout, float* bias_out, hipStream_t stream) {
    if (dtype == AP_F16) fold_ls_kernel<f16><<<rows, 256, 0, stream>>>(w32, cols, ld, ls, bias_in, (f16*)wout, bias_out);
    else if (dtype == AP_BF16) fold_ls_kernel<bf16><<<rows, 256, 0, stream>>>(w32, cols, ld, ls, bias_in, (bf16*)wout, bias_out);
    else { set_error("fold_ls: dtype %d (f16 / bf16 only)", dtype); return AP_ERR_INVALID; }
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int launch_cls_init(float* tok, const float* prefix, int prefix_rows, int n, int tokens, int dim, hipStream_t stream) {
    if (n <= 0) return AP_OK;
    const int total = n * prefix_rows * dim;
    cls_init_kernel<<<(total + 255) / 256, 256, 0, stream>>>(tok, prefix, prefix_rows, n, tokens, dim);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int launch_prefix_build(const float* cls, const float* reg, int reg_rows, const float* pos, int dim, float* prefix, hipStream_t stream) {
    AP_REQUIRE(cls && prefix && (reg || reg_rows == 0), "prefix_build: null pointer");
    const int total = (1 + reg_rows) * dim;
    prefix_build_kernel<<<(total + 255) / 256, 256, 0, stream>>>(cls, reg, reg_rows, pos, dim, prefix);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int launch_swiglu(int dtype, const void* x, int rows, int h, void* out, hipStream_t stream) {
    AP_REQUIRE(x && out && h > 0 && h % 8 == 0, "swiglu: bad arguments (h %d)", h);
    if (rows <= 0) return AP_OK;
    const size_t total = dtype == AP_F32 ? (size_t)rows * h : (size_t)rows * (h >> 3);
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (dtype == AP_F16) swiglu_kernel<f16><<<blocks, 256, 0, stream>>>((const f16*)x, rows, h, (f16*)out);
    else if (dtype == AP_BF16) swiglu_kernel<bf16><<<blocks, 256, 0, stream>>>((const bf16*)x, rows, h, (bf16*)out);
    else if (dtype == AP_F32) swiglu_f32_kernel<<<blocks, 256, 0, stream>>>((const float*)x, rows, h, (float*)out);
    else { set_error("swiglu: dtype %d", dtype); return AP_ERR_INVALID; }
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int launch_convert(int dtype, const float* src, void* dst, size_t count, hipStream_t stream) {
    if (count == 0) return AP_OK;
    const int blocks = (int)((count + 255) / 256 < 4096 ? (count + 255) / 256 : 4096);
    switch (dtype) {
        case AP_F16: convert_kernel<f16><<<blocks, 256, 0, stream>>>(src, (f16*)dst, count); break;
        case AP_BF16: convert_kernel<bf16><<<blocks, 256, 0, stream>>>(src, (bf16*)dst, count); break;
        case AP_F32: AP_HIP_CHECK(hipMemcpyAsync(dst, src, count * 4, hipMemcpyDeviceToDevice, stream)); return AP_OK;
        default: set_error("convert: unknown dtype %d", dtype); return AP_ERR_INVALID;
    }
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int launch_split_f16_weights(const float* src, void* dst, size_t count, hipStream_t stream) {
    AP_REQUIRE(src && dst && count % 32 == 0 && (const void*)src != dst, "split_f16_weights: count %zu must be a multiple of 32, out of place", count);
    if (count == 0) return AP_OK;
    const int blocks = (int)((count + 255) / 256 < 4096 ? (count + 255) / 256 : 4096);
    split_f16_kernel<<<blocks, 256, 0, stream>>>(src, (f16*)dst, count);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

template <typename TI>
static int chw_rows_typed(int dtype, const TI* x, int n, int S, int ps, void* dst, int ld,
                          hipStream_t stream) {
    const int g = S / ps;
    if (ps % 4 != 0) {
        const size_t total_any = (size_t)n * 3 * S * S;
        const unsigned blocks_any = (unsigned)((total_any + 255) / 256);
        switch (dtype) {
            case AP_F16: chw_to_patchrows_any_kernel<TI, f16><<<blocks_any, 256, 0, stream>>>(x, n, S, ps, (f16*)dst, ld); break;
            case AP_BF16: chw_to_patchrows_any_kernel<TI, bf16><<<blocks_any, 256, 0, stream>>>(x, n, S, ps, (bf16*)dst, ld); break;
            case AP_F32: chw_to_patchrows_any_kernel<TI, float><<<blocks_any, 256, 0, stream>>>(x, n, S, ps, (float*)dst, ld); break;
            default: set_error("chw_to_patchrows: unknown dtype %d", dtype); return AP_ERR_INVALID;
        }
        AP_HIP_CHECK(hipGetLastError());
        return AP_OK;
    }
    const size_t total = (size_t)n * 3 * S * g * (ps >> 2);
    const unsigned blocks = (unsigned)((total + 255) / 256);
    switch (dtype) {
        case AP_F16: chw_to_patchrows_kernel<TI, f16><<<blocks, 256, 0, stream>>>(x, n, S, ps, (f16*)dst, ld); break;
        case AP_BF16: chw_to_patchrows_kernel<TI, bf16><<<blocks, 256, 0, stream>>>(x, n, S, ps, (bf16*)dst, ld); break;
        case AP_F32: chw_to_patchrows_kernel<TI, float><<<blocks, 256, 0, stream>>>(x, n, S, ps, (float*)dst, ld); break;
        default: set_error("chw_to_patchrows: unknown dtype %d", dtype); return AP_ERR_INVALID;
    }
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int launch_chw_to_patchrows(int x_dtype, int dtype, const void* x, int n, int S, int ps, void* dst,
                            int ld, hipStream_t stream) {
    AP_REQUIRE(S % ps == 0, "chw_to_patchrows: image %d / patch %d unsupported", S, ps);
    if (n <= 0) return AP_OK;
    switch (x_dtype) {
        case AP_F32: return chw_rows_typed<float>(dtype, (const float*)x, n, S, ps, dst, ld, stream);
        case AP_F16: return chw_rows_typed<f16>(dtype, (const f16*)x, n, S, ps, dst, ld, stream);
        case AP_BF16: return chw_rows_typed<bf16>(dtype, (const bf16*)x, n, S, ps, dst, ld, stream);
    }
    set_error("chw_to_patchrows: unknown input dtype %d", x_dtype);
    return AP_ERR_INVALID;
}

}  // namespace ap
