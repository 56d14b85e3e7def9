// Attentional pooling with few learned queries (open_clip AttentionalPooler as CONCH's visual tower uses
// it: n_queries = 1): per (image, head) softmax(q_h . k_t / sqrt(64)) over all tokens t, weighted sum of v_t.
// The query is input independent (ln_q(query) projected by q_proj), so it arrives as a constant f32 vector.
//
//   kv : T [n * tokens, 2 * P]   (k | v, each P = heads * 64 wide), q : f32 [P], out : T [n, P]
//
// One 256-thread workgroup per (image, head).  Pass 1: a thread takes whole key rows (64 channels = 128 B,
// eight 16-byte loads) and leaves the scaled score in LDS; block max / sum; pass 2: thread = (channel,
// token group of 4), so a wave reads one 128-byte V row per step; the four groups are reduced through LDS.
// HBM-bound: k and v are read once (2 * tokens * 128 B per head).
#include "ap_common.h"

namespace ap {
namespace {

template <typename T> struct PoolVec;
template <> struct PoolVec<f16> { using v8 = f16x8; };
template <> struct PoolVec<bf16> { using v8 = bf16x8; };

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(v, off, 64);
        v = is_max ? fmaxf(v, o) : v + o;
    }
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    const float a = red[0], b = red[1], c = red[2], d = red[3];
    return is_max ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : (a + b) + (c + d);
}

template <typename T>
__global__ __launch_bounds__(256) void attn_pool_kernel(const T* __restrict__ kv, const float* __restrict__ q,
                                                        T* __restrict__ out, int tokens, int heads) {
    extern __shared__ __attribute__((aligned(16))) float sm[];       // scores[tokens] | red[4] | part[4][64]
    float* scores = sm;
    float* red = sm + ((tokens + 3) & ~3);
    float* part = red + 4;
    using V8 = typename PoolVec<T>::v8;
    const int img = blockIdx.x / heads, head = blockIdx.x - img * heads;
    const int P = heads * 64;
    const size_t ld = (size_t)2 * P;
    const T* kbase = kv + (size_t)img * tokens * ld + head * 64;
    const T* vbase = kbase + P;
    const float* qh = q + head * 64;

    float qr[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) qr[c] = qh[c];
    float mx = -INFINITY;
    for (int t = threadIdx.x; t < tokens; t += 256) {
        const T* kp = kbase + (size_t)t * ld;
        float s = 0.f;
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
            const V8 kk = *(const V8*)(kp + c8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) s = __builtin_fmaf((float)kk[e], qr[c8 * 8 + e], s);
        }
        s *= 0.125f;
        scores[t] = s;
        mx = fmaxf(mx, s);
    }
    mx = block_reduce(mx, red, true);
    float sum = 0.f;
    for (int t = threadIdx.x; t < tokens; t += 256) {
        const float p = __expf(scores[t] - mx);
        scores[t] = p;
        sum += p;
    }
    sum = block_reduce(sum, red, false);          // its barriers also publish scores[]
    const int c = threadIdx.x & 63, grp = threadIdx.x >> 6;
    float acc = 0.f;
    for (int t = grp; t < tokens; t += 4) acc = __builtin_fmaf(scores[t], (float)vbase[(size_t)t * ld + c], acc);
    part[grp * 64 + c] = acc;
    __syncthreads();
    if (grp == 0) {
        const float o = ((part[c] + part[64 + c]) + (part[128 + c] + part[192 + c])) / sum;
        out[(size_t)img * P + head * 64 + c] = (T)o;
    }
}


// One query per (image, head) taken from the data: the CLS row of the LAST encoder block.  Its output is the only
// row of that block anything reads (the CLS readout), so the block's attention is this kernel instead of the full
// [tokens x tokens] one.   q : T [n, heads * 64] (the CLS rows' projected queries, packed),
// kv rows: T [n * tokens, ld] with k at column koff + head * 64 and v at voff + head * 64; out : T [n, heads * 64].
// Same structure as the pooler above (scores in LDS, block softmax in f32, V rows streamed).
template <typename T> struct RowVec { using v8 = typename PoolVec<T>::v8; };
template <> struct RowVec<float> { typedef float v8 __attribute__((ext_vector_type(8))); };

template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_cls_kernel(const T* __restrict__ q, const T* __restrict__ kv, int ld, int koff,
                                                       int voff, T* __restrict__ out, int tokens, int heads, float scale) {
    // thread = (row slot r of kRows, channel octet sub of kSub): every K / V access is a 16-byte load and a wave covers
    // whole rows per step (HD = 64: eight 128-byte rows; HD = 128: four 256-byte rows; HD = 96: four 192-byte rows, the octets
    // 12 .. 15 of a row slot idle -- kSub stays a power of two for the lane exchanges)
    constexpr int kOct = HD / 8, kSub = HD == 64 ? 8 : 16, kRows = 256 / kSub;
    extern __shared__ __attribute__((aligned(16))) float sm[];       // scores[tokens] | red[4] | part[kRows][HD]
    float* scores = sm;
    float* red = sm + ((tokens + 3) & ~3);
    float* part = red + 4;
    using V8 = typename RowVec<T>::v8;
    const int img = blockIdx.x / heads, head = blockIdx.x - img * heads;
    const int P = heads * HD;
    const int sub = threadIdx.x & (kSub - 1), r = threadIdx.x / kSub;
    const bool oct = sub < kOct;
    const int subc = oct ? sub : 0;                                   // idle octets read (and ignore) octet 0
    const T* kbase = kv + (size_t)img * tokens * ld + koff + head * HD + subc * 8;
    const T* vbase = kv + (size_t)img * tokens * ld + voff + head * HD + subc * 8;
    float qr[8];
    {
        const V8 qq = *(const V8*)(q + (size_t)img * P + head * HD + subc * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) qr[e] = oct ? (float)qq[e] : 0.f;
    }
    float mx = -INFINITY;
    for (int t = r; t < tokens; t += kRows) {
        const V8 kk = *(const V8*)(kbase + (size_t)t * ld);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s = __builtin_fmaf((float)kk[e], qr[e], s);
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        if constexpr (kSub == 16) s += __shfl_xor(s, 8, 64);
        s *= scale;
        if (sub == 0) scores[t] = s;
        mx = fmaxf(mx, s);
    }
    mx = block_reduce(mx, red, true);              // its barriers also publish scores[]
    float sum = 0.f;
    for (int t = threadIdx.x; t < tokens; t += 256) {
        const float p = __expf(scores[t] - mx);
        scores[t] = p;
        sum += p;
    }
    sum = block_reduce(sum, red, false);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int t = r; t < tokens; t += kRows) {
        const float p = scores[t];
        const V8 vv = *(const V8*)(vbase + (size_t)t * ld);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = __builtin_fmaf(p, (float)vv[e], acc[e]);
    }
    if (oct) {
#pragma unroll
        for (int e = 0; e < 8; ++e) part[r * HD + sub * 8 + e] = acc[e];
    }
    __syncthreads();
    if (threadIdx.x < HD) {
        const int c = threadIdx.x;
        float o = 0.f;
#pragma unroll
        for (int g = 0; g < kRows; ++g) o += part[g * HD + c];
        out[(size_t)img * P + head * HD + c] = (T)(o / sum);
    }
}

}  // namespace

int launch_attn_pool(int dtype, const void* kv, const float* q, void* out, int n, int tokens, int heads,
                     hipStream_t stream) {
    AP_REQUIRE(dtype == AP_F16 || dtype == AP_BF16, "attn_pool: f16 / bf16 only");
    AP_REQUIRE(tokens > 0 && tokens <= 12000, "attn_pool: %d tokens unsupported", tokens);
    if (n <= 0) return AP_OK;
    const size_t lds = ((size_t)((tokens + 3) & ~3) + 4 + 256) * sizeof(float);
    dim3 grid(n * heads), block(256);
    if (dtype == AP_F16) attn_pool_kernel<f16><<<grid, block, lds, stream>>>((const f16*)kv, q, (f16*)out, tokens, heads);
    else attn_pool_kernel<bf16><<<grid, block, lds, stream>>>((const bf16*)kv, q, (bf16*)out, tokens, heads);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

int launch_attention_cls(int dtype, const void* q, const void* kv, int ld, int koff, int voff, void* out, int n,
                         int tokens, int heads, int head_dim, float scale, hipStream_t stream) {
    AP_REQUIRE(head_dim == 64 || head_dim == 96 || head_dim == 128, "attention_cls: head_dim %d unsupported (64 / 96 / 128)", head_dim);
    AP_REQUIRE(tokens > 0 && tokens <= 12000, "attention_cls: %d tokens unsupported", tokens);
    AP_REQUIRE(ld % 8 == 0 && koff % 8 == 0 && voff % 8 == 0, "attention_cls: misaligned layout");
    if (n <= 0) return AP_OK;
    const size_t lds = ((size_t)((tokens + 3) & ~3) + 4 + 32 * 64) * sizeof(float);       // part: kRows * HD = 2048 floats either way
    dim3 grid(n * heads), block(256);
#define AP_CLS(T, HD) attn_cls_kernel<T, HD><<<grid, block, lds, stream>>>((const T*)q, (const T*)kv, ld, koff, voff, (T*)out, tokens, heads, scale)
#define AP_CLS_HD(T) do { if (head_dim == 64) AP_CLS(T, 64); else if (head_dim == 96) AP_CLS(T, 96); else AP_CLS(T, 128); } while (0)
    if (dtype == AP_F16) AP_CLS_HD(f16);
    else if (dtype == AP_BF16) AP_CLS_HD(bf16);
    else if (dtype == AP_F32) AP_CLS_HD(float);
    else { set_error("attention_cls: unsupported dtype %d", dtype); return AP_ERR_INVALID; }
#undef AP_CLS_HD
#undef AP_CLS
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

}  // namespace ap
