// HBM-bound glue kernels of the ViT encoder: LayerNorm, CLS/pos init, dtype convert and
// CHW -> patch-row gather.  All are streaming kernels: 16-byte loads per lane, one wave
// per row where a row reduction is needed (no LDS, no block barrier).
//
// Algorithmic bytes: LayerNorm reads 4*dim and writes sizeof(T)*dim per row.
#include "ap_common.h"

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

// One wave per row, whole row held in registers (dim <= 64 * 4 * kMaxVec).
constexpr int kMaxVec = 8;

template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, long stride,
                                                        int rows, int dim,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        T* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = dim >> 2;
    const f32x4* src = (const f32x4*)(x + (size_t)row * stride);
    f32x4 v[kMaxVec];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxVec; ++i) {
        const int idx = lane + i * 64;
        if (idx < nvec) {
            v[i] = src[idx];
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    const float mean = wave_sum(s) / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxVec; ++i) {
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
    T* dst = out + (size_t)row * dim;
#pragma unroll
    for (int i = 0; i < kMaxVec; ++i) {
        const int idx = lane + i * 64;
        if (idx < nvec) {
            const f32x4 ga = ((const f32x4*)gamma)[idx];
            const f32x4 be = ((const f32x4*)beta)[idx];
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (v[i][e] - mean) * rstd * ga[e] + be[e];
            store_vec4<T>(dst + idx * 4, y);
        }
    }
}

__global__ void cls_init_kernel(float* tok, const float* cls, const float* pos, int n, int tokens,
                                int dim) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * dim) return;
    const int img = i / dim, d = i - img * dim;
    tok[(size_t)img * tokens * dim + d] = cls[d] + pos[d];
}

template <typename T>
__global__ void convert_kernel(const float* __restrict__ src, T* __restrict__ dst, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < count; i += step) dst[i] = from_f32<T>(src[i]);
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

}  // namespace

int launch_layernorm(int dtype, const float* x, long stride, int rows, int dim, const float* gamma,
                     const float* beta, float eps, void* out, hipStream_t stream) {
    AP_REQUIRE(dim % 4 == 0 && dim <= 64 * 4 * kMaxVec, "layernorm: unsupported dim %d", dim);
    AP_REQUIRE(stride % 4 == 0, "layernorm: row stride must be a multiple of 4");
    if (rows <= 0) return AP_OK;
    dim3 grid((rows + 3) / 4), block(256);
    switch (dtype) {
        case AP_F16: layernorm_kernel<f16><<<grid, block, 0, stream>>>(x, stride, rows, dim, gamma, beta, eps, (f16*)out); break;
        case AP_BF16: layernorm_kernel<bf16><<<grid, block, 0, stream>>>(x, stride, rows, dim, gamma, beta, eps, (bf16*)out); break;
        case AP_F32: layernorm_kernel<float><<<grid, block, 0, stream>>>(x, stride, rows, dim, gamma, beta, eps, (float*)out); break;
        default: set_error("layernorm: unknown dtype %d", dtype); return AP_ERR_INVALID;
    }
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int launch_layernorm_f32out(const float* x, long stride, int rows, int dim, const float* gamma,
                            const float* beta, float eps, float* out, hipStream_t stream) {
    return launch_layernorm(AP_F32, x, stride, rows, dim, gamma, beta, eps, out, stream);
}

int launch_cls_init(float* tok, const float* cls, const float* pos, int n, int tokens, int dim,
                    hipStream_t stream) {
    if (n <= 0) return AP_OK;
    const int total = n * dim;
    cls_init_kernel<<<(total + 255) / 256, 256, 0, stream>>>(tok, cls, pos, n, tokens, dim);
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

template <typename TI>
static int chw_rows_typed(int dtype, const TI* x, int n, int S, int ps, void* dst, int ld,
                          hipStream_t stream) {
    const int g = S / ps;
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
    AP_REQUIRE(S % ps == 0 && ps % 4 == 0, "chw_to_patchrows: image %d / patch %d unsupported", S, ps);
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
