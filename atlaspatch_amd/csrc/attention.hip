// Fused multi-head attention for the ViT encoder (short sequences: T <= 288 tokens, head_dim 64).
//
// One workgroup per (image, head): 8 waves for f16 / bf16 (one 32-query block each, so the seven
// blocks of T = 197 run side by side and two resident workgroups give each SIMD four waves to hide
// the K/V staging and Q load latency), 4 waves for f32.  K and V^T of that head live in LDS for
// the whole block (K: XOR-swizzled rows; V: transposed so that 4 consecutive keys of one channel
// are 8/16 contiguous bytes); each wave keeps its whole 32 x T score strip in registers -- no
// online-softmax rescaling is needed at these lengths.
//
//   S^T = K Q^T   (MFMA A-operand = K rows from LDS, B-operand = Q rows from global):
//         lane (q = lane & 31) holds the scores of ITS query against half of the keys, its
//         partner lane ^ 32 holds the other half -> row max / row sum are in-register
//         reductions plus ONE cross-lane exchange each.
//   O^T = V^T P^T (A-operand = V^T rows from LDS, B-operand = the lane's own P registers):
//         the k index of this product is permuted to exactly the key order the S^T
//         accumulators already have, so P never moves between lanes; the result lane again
//         owns query q, so 1/rowsum is lane-local and 4 consecutive channels are stored as
//         one 8-byte (f16/bf16) or 16-byte (f32) write.
//
// f16/bf16: v_mfma_f32_32x32x16; f32: v_mfma_f32_32x32x2_f32 (exact f32).  Softmax is f32 in
// all modes (exp2 with log2(e)/sqrt(d) folded into the scale).  T = 197 is padded to 224 keys
// with zero K/V rows and -inf scores.
//
// Roofline: MFMA for the two contractions (4*T*T*64 flop per head); HBM traffic = read
// q,k,v once + write o once = 4 * T * 64 * sizeof(T) bytes per head.
#include <cstdlib>
#include "ap_common.h"

namespace ap {
namespace {

constexpr int kHD = 64;   // head dim

template <typename T, int NKT>
struct AttnSmem {
    static constexpr int TP = NKT * 32;                         // padded keys
    static constexpr int kRowB = kHD * (int)sizeof(T);          // K row bytes (128 / 256)
    static constexpr int vRowB = (TP + 4) * (int)sizeof(T);     // V^T row bytes (conflict-free pad)
    static constexpr int kBytes = TP * kRowB;
    static constexpr int vBytes = kHD * vRowB;
    static constexpr int total = kBytes + vBytes;
};

template <typename T> __device__ __forceinline__ uint32_t pack2(T a, T b);
template <> __device__ __forceinline__ uint32_t pack2<f16>(f16 a, f16 b) {
    return (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
}
template <> __device__ __forceinline__ uint32_t pack2<bf16>(bf16 a, bf16 b) {
    return (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
}

template <typename T> struct Vec8;
template <> struct Vec8<f16> { using type = f16x8; using half = f16x4; };
template <> struct Vec8<bf16> { using type = bf16x8; using half = bf16x4; };

template <typename T>
__device__ __forceinline__ f32x16 mma16(typename Vec8<T>::type a, typename Vec8<T>::type b, f32x16 c);
template <>
__device__ __forceinline__ f32x16 mma16<f16>(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x16 mma16<bf16>(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

template <typename T, int NKT, int NW>
__global__ __launch_bounds__(NW * 64, (sizeof(T) == 2 ? 2 : 1))
void attention_kernel(const T* __restrict__ qkv, T* __restrict__ out, int tokens, int heads, float scale) {
    constexpr int NT = NW * 64;
    using S = AttnSmem<T, NKT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vt = smem + S::kBytes;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    const int img = blockIdx.x / heads, head = blockIdx.x - img * heads;
    const int dim = heads * kHD;
    const size_t ld = (size_t)3 * dim;
    const T* base = qkv + (size_t)img * tokens * ld + head * kHD;
    const T* Qg = base;
    const T* Kg = base + dim;
    const T* Vg = base + 2 * dim;

    // ---------------- stage K (swizzled rows) and V^T
    if constexpr (sizeof(T) == 2) {
        for (int idx = tid; idx < S::TP * 8; idx += NT) {
            const int t = idx >> 3, c = idx & 7;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (t < tokens) v = *(const u32x4*)(Kg + (size_t)t * ld + c * 8);
            *(u32x4*)(Ks + t * S::kRowB + ((c ^ ((t >> 1) & 7)) << 4)) = v;
        }
        for (int idx = tid; idx < (S::TP / 2) * 8; idx += NT) {
            const int tp = idx >> 3, c = idx & 7;
            const int t0 = tp * 2;
            typename Vec8<T>::type v0, v1;
#pragma unroll
            for (int e = 0; e < 8; ++e) { v0[e] = (T)0.0f; v1[e] = (T)0.0f; }
            if (t0 < tokens) v0 = *(const typename Vec8<T>::type*)(Vg + (size_t)t0 * ld + c * 8);
            if (t0 + 1 < tokens) v1 = *(const typename Vec8<T>::type*)(Vg + (size_t)(t0 + 1) * ld + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                *(uint32_t*)(Vt + (c * 8 + e) * S::vRowB + tp * 4) = pack2<T>(v0[e], v1[e]);
        }
    } else {
        for (int idx = tid; idx < S::TP * 16; idx += NT) {
            const int t = idx >> 4, c = idx & 15;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (t < tokens) v = *(const f32x4*)(Kg + (size_t)t * ld + c * 4);
            *(f32x4*)(Ks + t * S::kRowB + ((c ^ (t & 15)) << 4)) = v;
            f32x4 w = {0.f, 0.f, 0.f, 0.f};
            if (t < tokens) w = *(const f32x4*)(Vg + (size_t)t * ld + c * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) *(float*)(Vt + (c * 4 + e) * S::vRowB + t * 4) = w[e];
        }
    }
    __syncthreads();

    const float scale_log2 = scale * 1.4426950408889634f;       // log2(e) / sqrt(head width)
    const int nqb = (tokens + 31) >> 5;

    for (int qb = wave; qb < nqb; qb += NT / 64) {
        int qrow = qb * 32 + l31;
        const bool qvalid = qrow < tokens;
        if (!qvalid) qrow = tokens - 1;
        const T* qp = Qg + (size_t)qrow * ld;

        f32x16 st[NKT];
        if constexpr (sizeof(T) == 2) {
            using V8 = typename Vec8<T>::type;
            V8 qf[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const V8*)(qp + kk * 16 + hi * 8);
            const int xr = (l31 >> 1) & 7;
            // register double-buffer of the K fragments; the empty asm stops hipcc from hoisting
            // every tile's LDS reads to the top (which spills)
            V8 kcur[4], knext[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                kcur[kk] = *(const V8*)(Ks + l31 * S::kRowB + (((kk * 2 + hi) ^ xr) << 4));
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
                for (int e = 0; e < 16; ++e) st[kt][e] = 0.f;
                if (kt + 1 < NKT) {
                    const char* krow = Ks + ((kt + 1) * 32 + l31) * S::kRowB;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) knext[kk] = *(const V8*)(krow + (((kk * 2 + hi) ^ xr) << 4));
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) st[kt] = mma16<T>(kcur[kk], qf[kk], st[kt]);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) kcur[kk] = knext[kk];
            }
        } else {
            f32x4 qf[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) qf[c] = *(const f32x4*)(qp + hi * 32 + c * 4);
            const int xr = l31 & 15;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
                for (int e = 0; e < 16; ++e) st[kt][e] = 0.f;
                const char* krow = Ks + (kt * 32 + l31) * S::kRowB;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if ((c & 3) == 0) asm volatile("" ::: "memory");
                    f32x4 kf = *(const f32x4*)(krow + (((hi * 8 + c) ^ xr) << 4));
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        st[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[c][e], st[kt], 0, 0, 0);
                }
            }
        }

        // ---------------- softmax over the lane's half of the keys (+ partner lane ^ 32)
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if ((kt + 1) * 32 > tokens) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= tokens) st[kt][r] = -INFINITY;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kt][r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mb = mx * scale_log2;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(st[kt][r] * scale_log2 - mb);
                st[kt][r] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;

        // ---------------- O^T = V^T P^T
        f32x16 ot[2];
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int e = 0; e < 16; ++e) ot[it][e] = 0.f;

        if constexpr (sizeof(T) == 2) {
            using V8 = typename Vec8<T>::type;
            using V4 = typename Vec8<T>::half;
            auto load_v = [&](int s, int it) {
                const int kt = s >> 1, hf = s & 1;
                const char* vrow = Vt + (it * 32 + l31) * S::vRowB + (kt * 32 + hf * 16 + hi * 4) * 2;
                const V4 lo = *(const V4*)(vrow);
                const V4 hi4 = *(const V4*)(vrow + 16);
                V8 vf = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                return vf;
            };
            V8 vcur0 = load_v(0, 0), vcur1 = load_v(0, 1), vnext0 = vcur0, vnext1 = vcur1;
#pragma unroll
            for (int s = 0; s < 2 * NKT; ++s) {
                const int kt = s >> 1, hf = s & 1;
                if (s + 1 < 2 * NKT) { vnext0 = load_v(s + 1, 0); vnext1 = load_v(s + 1, 1); }
                asm volatile("" ::: "memory");
                V8 pf;
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[e] = (T)st[kt][hf * 8 + e];
                ot[0] = mma16<T>(vcur0, pf, ot[0]);
                ot[1] = mma16<T>(vcur1, pf, ot[1]);
                vcur0 = vnext0; vcur1 = vnext1;
            }
        } else {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        if (it == 0) asm volatile("" ::: "memory");
                        const f32x4 vf = *(const f32x4*)(Vt + (it * 32 + l31) * S::vRowB +
                                                         (kt * 32 + g4 * 8 + hi * 4) * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            ot[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[e], st[kt][g4 * 4 + e],
                                                                          ot[it], 0, 0, 0);
                    }
        }

        if (qvalid) {
            T* op = out + ((size_t)img * tokens + qrow) * dim + head * kHD;
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    T* dst = op + it * 32 + g4 * 8 + hi * 4;
                    if constexpr (sizeof(T) == 2) {
                        typename Vec8<T>::half o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (T)(ot[it][g4 * 4 + e] * inv);
                        *(typename Vec8<T>::half*)dst = o;
                    } else {
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = ot[it][g4 * 4 + e] * inv;
                        *(f32x4*)dst = o;
                    }
                }
        }
    }
}

template <typename T, int NKT, int NW>
int launch_nw(const void* qkv, void* out, int n, int tokens, int heads, float scale, hipStream_t stream) {
    using S = AttnSmem<T, NKT>;
    static bool configured = false;
    auto kern = attention_kernel<T, NKT, NW>;
    if (!configured) {
        AP_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         S::total));
        configured = true;
    }
    kern<<<dim3(n * heads), dim3(NW * 64), S::total, stream>>>((const T*)qkv, (T*)out, tokens, heads, scale);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

template <typename T, int NKT>
int launch_one(const void* qkv, void* out, int n, int tokens, int heads, float scale, hipStream_t stream) {
    static const int waves = [] { const char* e = getenv("AP_ATTN_WAVES"); return e ? atoi(e) : 4; }();
    if constexpr (sizeof(T) == 2) {              // (the float kernel exists with 4 waves only: with 8 its 9-tile form spills)
        if (waves == 8) return launch_nw<T, NKT, 8>(qkv, out, n, tokens, heads, scale, stream);
    }
    return launch_nw<T, NKT, 4>(qkv, out, n, tokens, heads, scale, stream);
}

template <typename T>
int launch_by_len(const void* qkv, void* out, int n, int tokens, int heads, float scale, hipStream_t stream) {
    if (tokens <= 224) return launch_one<T, 7>(qkv, out, n, tokens, heads, scale, stream);
    if (tokens <= 288) return launch_one<T, 9>(qkv, out, n, tokens, heads, scale, stream);
    set_error("attention: %d tokens not supported by this build (max 288)", tokens);
    return AP_ERR_UNSUPPORTED;
}

}  // namespace

int launch_attention(int dtype, const void* qkv, void* out, int n, int tokens, int heads,
                     int head_dim, float scale, hipStream_t stream) {
    AP_REQUIRE(head_dim == kHD || ((head_dim == 96 || head_dim == 128) && dtype != AP_F32),
               "attention: head_dim %d unsupported (64; 96 / 128 in f16 / bf16)", head_dim);
    AP_REQUIRE(tokens > 0 && heads > 0 && scale > 0.f, "attention: bad shape");
    if (n <= 0) return AP_OK;
    // f16 / bf16: the tiled online-softmax kernel (attention_flash.hip; any length, 0.37 ms vs 0.49 ms for
    // the strip kernel at n = 1024, T = 197, H = 12).  AP_ATTN_IMPL=strip selects the register-strip kernel
    // below (T <= 288) for A/B timing; f32 always uses it.
    if (dtype != AP_F32) {
        static const bool strip = [] { const char* e = getenv("AP_ATTN_IMPL"); return e && e[0] == 's'; }();
        if (!strip || tokens > 288 || head_dim != kHD) return launch_attention_flash(dtype, qkv, out, n, tokens, heads, head_dim, scale, stream);
    }
    switch (dtype) {
        case AP_F16: return launch_by_len<f16>(qkv, out, n, tokens, heads, scale, stream);
        case AP_BF16: return launch_by_len<bf16>(qkv, out, n, tokens, heads, scale, stream);
        case AP_F32: return launch_by_len<float>(qkv, out, n, tokens, heads, scale, stream);
    }
    set_error("attention: unknown dtype %d", dtype);
    return AP_ERR_INVALID;
}

}  // namespace ap
