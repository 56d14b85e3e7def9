// Internal helpers shared by the gfx950 kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include "../../include/atlaspatch_hip.h"

namespace ap {

void set_error(const char* fmt, ...);

#define AP_HIP_CHECK(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            ap::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),        \
                          __FILE__, __LINE__);                                          \
            return AP_ERR_HIP;                                                          \
        }                                                                               \
    } while (0)

#define AP_REQUIRE(cond, ...)                                                           \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            ap::set_error(__VA_ARGS__);                                                 \
            return AP_ERR_INVALID;                                                      \
        }                                                                               \
    } while (0)

using f16 = _Float16;
using bf16 = __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;

inline size_t dtype_size(int dt) { return dt == AP_F32 ? 4 : 2; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

template <typename T> struct DType;
template <> struct DType<float> { static constexpr int id = AP_F32; };
template <> struct DType<f16> { static constexpr int id = AP_F16; };
template <> struct DType<bf16> { static constexpr int id = AP_BF16; };

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ f16 from_f32<f16>(float v) { return (f16)v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

// GELU(x) = x * Phi(x) for results that are rounded to f16 / bf16 right after (both GEMM kernels):
//   Phi(x) ~= sigmoid(x * (c0 + c1 t + c2 t^2)),  t = min(x^2, 50)
// (coefficients: minimax fit against 0.5 x (1 + erf(x / sqrt 2)) on [-9, 9], max |error| 2.6e-5,
// below f16's half-ulp for |gelu| >= 0.06; the clamp keeps the odd polynomial monotone so the
// sigmoid saturates correctly for any |x|).  -log2(e) is folded into the coefficients:
// per value 3 packed (mul, 2 fma, fma, add, mul over two values) + v_exp_f32 + v_rcp_f32 (rounds 1-5: + 1 v_min, see below).  The f32 mode uses libm erff.
// Evaluated two values at a time with the full-rate operations packed two per instruction (v_pk_mul_f32 /
// v_pk_fma_f32 / v_pk_add_f32); exp and rcp have no packed form.  The fc1 epilogue is VALU-bound on this.
// Both GEMM kernels call THIS routine (hipcc contracts the scalar form differently: 1-ulp differences), so the
// 128x128 and the 256x256 kernel stay bit-identical.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// Round 6: the routine takes xs = x * kGeluS (kGeluS = 1 / sqrt(50)), so that the clamp of t is the packed multiply's CLAMP bit
//   t' = clamp01(xs * xs) = min(x^2, 50) / 50            (v_pk_mul_f32 ... clamp: no v_min)
// and everything downstream is re-expressed in xs with the constants rescaled (same instruction count otherwise):
//   z' = xs * (C0 + C1 t' + C2 t'^2) + log2(s)  =  x p + log2(s),      gelu = xs / (s + 2^z')  =  x / (1 + 2^(x p)).
// The fused-LayerNorm epilogues produce xs directly (the row's rstd and -mean rstd and the tile's bias registers are
// multiplied by s once per tile: 24 packed multiplies against 128 v_min per tile and wave); the plain-bias epilogues multiply.
constexpr float kGeluS = 0.14142135623730950488f;
__device__ __forceinline__ f32x2_t gelu_sigmoid_poly2_s(f32x2_t xs) {
    f32x2_t t;
    asm("v_pk_mul_f32 %0, %1, %1 clamp" : "=v"(t) : "v"(xs));
    constexpr float c2 = 1.0148166e-3f * 2500.0f / kGeluS, c1 = -1.0677913e-1f * 50.0f / kGeluS, c0 = -2.3011176f / kGeluS;
    f32x2_t p = __builtin_elementwise_fma(t, f32x2_t{c2, c2}, f32x2_t{c1, c1});
    p = __builtin_elementwise_fma(t, p, f32x2_t{c0, c0});
    const f32x2_t z = __builtin_elementwise_fma(xs, p, f32x2_t{-2.8219280948873623f, -2.8219280948873623f});      // + log2(s)
    const f32x2_t d = f32x2_t{kGeluS, kGeluS} + f32x2_t{__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])};
    return xs * f32x2_t{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}
__device__ __forceinline__ f32x2_t gelu_sigmoid_poly2(f32x2_t x) { return gelu_sigmoid_poly2_s(x * f32x2_t{kGeluS, kGeluS}); }

// CLIP's QuickGELU (models/patch/clip.py: the OpenAI weights; transformers "quick_gelu"): x * sigmoid(1.702 x), two values, f32
// (exp2 + rcp; both GEMM kernels call THIS routine)
__device__ __forceinline__ f32x2_t quick_gelu2(f32x2_t x) {
    const f32x2_t z = x * f32x2_t{-1.702f * 1.4426950408889634f, -1.702f * 1.4426950408889634f};
    const f32x2_t d = f32x2_t{1.0f, 1.0f} + f32x2_t{__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])};
    return x * f32x2_t{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}

// silu(a) * b for two values, f32 (exp2 + rcp; both GEMM kernels call THIS routine so that they stay bit-identical)
__device__ __forceinline__ f32x2_t swiglu2(f32x2_t a, f32x2_t b) {
    const f32x2_t z = a * f32x2_t{-1.4426950408889634f, -1.4426950408889634f};
    const f32x2_t d = f32x2_t{1.0f, 1.0f} + f32x2_t{__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])};
    return (a * f32x2_t{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])}) * b;
}

// ---- GEMM ------------------------------------------------------------------------------
// C[m][n] = sum_k A[m][k] * W[n][k]  (+ epilogue); A: activations [M, lda], W: weights [N, ldw],
// both K-contiguous in the compute dtype.
enum GemmEpilogue {
    EPI_BIAS_STORE = 0,    // out[m][n] = T((acc + bias[n]) * (gamma ? gamma[n] : 1))
    EPI_BIAS_GELU = 1,     // out[m][n] = T(gelu_erf(acc + bias[n]))
    EPI_BIAS_RESID = 2,    // resid[m][n] += (acc + bias[n]) * (gamma ? gamma[n] : 1)   (f32, in place)
    EPI_PATCH_EMBED = 3,   // tok[(m / P) * (P + 1) + 1 + m % P][n] = acc + bias[n] + pos[1 + m % P][n]
    // ---- fused-LayerNorm family (16-bit residual stream, gemm256 only; see vit.cpp::run_blocks_fused)
    // The A operand is the raw stream x (T); the weights carry the LayerNorm gain (W' = W * gamma), so
    //   LN(x) W^T + b = rstd[m] * (sum_k x[m][k] W'[n][k] - mean[m] * colsum[n]) + bias'[n]
    // with colsum[n] = sum_k W'[n][k], bias' = b + W beta and rowstats[m] = (rstd, -mean * rstd).
    EPI_NORM_STORE = 4,    // out[m][n] = T(rstd[m] * acc + (-mean[m] rstd[m]) * colsum[n] + bias[n])
    EPI_NORM_GELU = 5,     // ... = T(gelu(that))
    // out[m][n] = T(out[m][n] + T(acc + bias[n]))  (residual add in place, T = f16 / bf16) and, per row and
    // 64-column group, partial[m][n / 64] = (sum, sum of squares) of the NEW row values (f32): what the
    // next fused-LayerNorm GEMM's rowstats are built from (launch_rowstats_finalize)
    EPI_RESID_STATS = 6,
    // patch embedding straight into the T stream: row m = img * P + p of the patch matrix lands in stream row
    // img * (P + 1) + 1 + p as  T(pos16[1 + p][n] + T(acc + bias[n]))  together with that row's partial sums (the RESID_STATS
    // epilogue with the "residual" read from the T copy of the position embedding and remapped output rows)
    EPI_PATCH_STREAM = 7,
    // timm SwiGLUPacked fc1 with the gate in the epilogue (uni_v2): the weight rows are INTERLEAVED in groups of 64 -- rows
    // 64q .. 64q+31 = fc1 rows 32q .. 32q+31 (x1), rows 64q+32 .. 64q+63 = fc1 rows H+32q .. (x2) -- so that a wave's two 32-wide
    // n blocks hold x1 and x2 of the same 32 output columns; out[m][32q + j] = T(silu(norm(x1)) * norm(x2)), out: T [M, N / 2]
    EPI_NORM_SWIGLU = 8,
    EPI_NORM_QGELU = 9,    // EPI_NORM_GELU with CLIP's QuickGELU x * sigmoid(1.702 x)
    EPI_BIAS_QGELU = 10,   // EPI_BIAS_GELU likewise
};

struct GemmArgs {
    const void* A; int lda;
    const void* W; int ldw;
    int M, N, K;
    const float* bias;
    const float* gamma;      // EPI_BIAS_STORE / EPI_BIAS_RESID: LayerScale (may be null)
    const float* pos;        // EPI_PATCH_EMBED: [P + 1, N]
    const float* colsum;     // EPI_NORM_*: f32 [N]
    const float* rowstats;   // EPI_NORM_*: f32 [M, 2] = (rstd, -mean * rstd)
    float* partial;          // EPI_RESID_STATS / EPI_PATCH_STREAM: f32 [rows, N / 64, 2]
    const void* pos16;       // EPI_PATCH_STREAM: T [P + 1, N]
    void* out; int ldo;      // T (STORE / GELU) or f32 (RESID / PATCH_EMBED)
    int P;                   // EPI_PATCH_EMBED / EPI_PATCH_STREAM: patches per image
    int R;                   //   prefix rows per image in the token stream (class + register tokens): patch row
                             //   m = img * P + p goes to stream row img * (P + R) + R + p
    int pos_row0;            //   row of pos / pos16 that belongs to patch 0 (R, or 0 when the position embedding has no
                             //   rows for the prefix tokens: timm no_embed_class)
    int ablate;              // gemm256 A/B twin only: timing ablation flags (results invalid when set)
    int walk_cols;           // gemm256 only: tile-walk column-group width (0 = kernel default)
    int skew_ticks;          // gemm256 only: start-time spread across an XCD's workgroups (100 MHz ticks)
    long long* trace;        // gemm256 diagnostics: per (workgroup, tile) 8 x 100-MHz time stamps, or null
    int trace_tiles;         //   tiles recorded per workgroup
    // EPI_RESID_STATS, exact class rows (vit.cpp): rows m = img * cls_tokens ALSO leave their unrounded branch value
    // (accumulator, bias included) in cls_branch[img][n], f32 [M / cls_tokens, N]; cls_tokens = 0: off.  M < 2^24.
    float* cls_branch;
    int cls_tokens;
    // float32 buffers with split-f16 products (gemm.hip, 128 x 128 kernel): W points at the [hi 32 | lo 32] f16 rows that
    // launch_split_f16_weights made of the f32 matrix (same row stride in bytes, ldw still counts f32 elements)
    int split;
    const float* resid; int ldr;   // split form, EPI_BIAS_STORE / EPI_BIAS_GELU: out = act(acc + bias) + resid[m][n] (f32 [M, ldr], may alias nothing)
    // split form, windowed layers of the SAM2 trunk (hieradet.py window_partition / window_unpartition folded into the GEMM):
    // win_mode 1: A is [B * H * W, lda] in image order and row m of the product is the window-order row (padding rows read
    // zero_row, f32 [>= K] zeros); win_mode 2: out / resid are in image order, window-order row m is scattered, padding rows dropped
    int win_mode, win_ws, win_H, win_W, win_nwy, win_nwx;
    const float* zero_row;
};
void set_gemm_trace(long long* buf, int tiles_per_wg);

int launch_gemm(int dtype, int epilogue, const GemmArgs& a, hipStream_t stream);
// impl: 0 = pick (256x256 persistent kernel when it supports the problem), 128 / 256 = force,
// 257 = the experimental twin of the 256 kernel (see gemm256.hip);
// variant: gemm256 tuning knob: bits 4.. = workgroup start skew in percent of the estimated tile time
int launch_gemm_impl(int dtype, int epilogue, const GemmArgs& a, int impl, int variant, hipStream_t stream);
bool gemm256_supports(int dtype, int epilogue, const GemmArgs& a);
int launch_gemm256(int dtype, int epilogue, const GemmArgs& a, int variant, hipStream_t stream);
int launch_gemm256_alt(int dtype, int epilogue, const GemmArgs& a, int variant, hipStream_t stream);   // A/B twin (impl 257)

// ---- LayerNorm / attention / misc ---------------------------------------------------------
// rows of f32 [rows, dim] (row stride `stride` elements) -> T [rows, dim] (dense)
int launch_layernorm(int dtype, const float* x, long stride, int rows, int dim,
                     const float* gamma, const float* beta, float eps, void* out,
                     hipStream_t stream);
// x[row] += delta[row] * ls (delta: delta_dtype rows, stride dstride elements, may be null; ls:
// f32 [dim] LayerScale, may be null), then out[row] = LayerNorm(x[row]) in out_dtype.
// Supported pairs: (T, T) and (T, f32).
int launch_add_layernorm(int delta_dtype, int out_dtype, float* x, long stride, const void* delta,
                         long dstride, const float* ls, int rows, int dim, const float* gamma, const float* beta,
                         float eps, void* out, hipStream_t stream);
// two pending branch outputs folded in one pass (delta0 first); store = 0: x itself is left untouched
int launch_add2_layernorm(int delta_dtype, int out_dtype, float* x, long stride, const void* delta0, long dstride0,
                          const float* ls0, const void* delta1, long dstride1, const float* ls1, int store,
                          int rows, int dim, const float* gamma, const float* beta, float eps, void* out,
                          hipStream_t stream);
// f32 out variant used for the final norm on CLS rows
int launch_layernorm_f32out(const float* x, long stride, int rows, int dim, const float* gamma,
                            const float* beta, float eps, float* out, hipStream_t stream);
// qkv: T [n*tokens, 3*dim] (q | k | v), out: T [n*tokens, dim]
// head_dim: 64 or 128 (f32: 64); scale: the softmax scale (1 / sqrt(true head width))
int launch_attention(int dtype, const void* qkv, void* out, int n, int tokens, int heads,
                     int head_dim, float scale, hipStream_t stream);
int launch_attention_flash(int dtype, const void* qkv, void* out, int n, int tokens, int heads, int head_dim, float scale,
                           hipStream_t stream);
// attentional pooling (attn_pool.hip): kv T [n*tokens, 2*heads*64] (k | v), q f32 [heads*64] -> out T [n, heads*64]
// one query row per (image, head) (CLS row of the last block); q packed [n, heads*64], k / v inside rows of `kv`
int launch_attention_cls(int dtype, const void* q, const void* kv, int ld, int koff, int voff, void* out, int n,
                         int tokens, int heads, int head_dim, float scale, hipStream_t stream);
int launch_attn_pool(int dtype, const void* kv, const float* q, void* out, int n, int tokens, int heads,
                     hipStream_t stream);
// ---- fused-LayerNorm path (16-bit residual stream) ---------------------------------------
// tok f32 [rows, dim] (dense) -> x T [rows, dim] and rowstats f32 [rows, 2] = (rstd, -mean * rstd) of the ROUNDED row
int launch_stream_init(int dtype, const float* tok, int rows, int dim, float eps, void* x, float* rowstats,
                       hipStream_t stream);
// partial f32 [rows, groups, 2] (sum, sum of squares per 64-column group) -> rowstats f32 [rows, 2]
int launch_rowstats_finalize(const float* partial, int rows, int groups, int dim, float eps, float* rowstats,
                             hipStream_t stream);
// ... with the exact class rows folded in (vit.cpp): n extra workgroups do cls32[img] += branch[img], x[img * tokens] = T(cls32[img])
// and write that row's statistics themselves; cls32 = nullptr: the plain form
int launch_rowstats_finalize_cls(const float* partial, int rows, int groups, int dim, float eps, float* rowstats, int dtype,
                                 float* cls32, const float* branch, void* x, int n, int tokens, hipStream_t stream);
// x T rows (row stride `stride` elements) -> dst f32 [rows, dim] dense
int launch_stream_to_f32(int dtype, const void* x, long stride, int rows, int dim, float* dst, hipStream_t stream);
// Rotary position embedding of DINOv3 (transformers DINOv3ViT apply_rotary_pos_emb): in place on the q (which & 1) and k
// (which & 2) parts of the packed qkv [n * tokens, 3 * heads * head_dim], PATCH tokens only (rows >= prefix of every image):
//   x'[j] = x[j] cos[p][j] - x[j + h] sin[p][j],   x'[j + h] = x[j + h] cos[p][j + h] + x[j] sin[p][j + h],   h = head_dim / 2
// cos / sin: f32 [patches, head_dim]; f32 arithmetic, one rounding to the compute type.
int launch_rope(int dtype, void* qkv, int n, int tokens, int prefix, int heads, int head_dim, const float* cos, const float* sin,
                int which, hipStream_t stream);
// AP_POOL_CLS_MEAN: y f32 [n * tokens, dim] (final LayerNorm of every token) -> out f32 [n, 2 * dim] = [row 0 | mean of rows prefix ..]
int launch_cls_mean_pool(const float* y, int n, int tokens, int prefix, int dim, float* out, hipStream_t stream);
// Weight folding (ap_vit_finalize).  w32: f32 [rows, ld] (zero padded beyond cols).
//   fold_ln: wout T [rows, ld] = T(w32[n][k] * gamma[k]); colsum[n] = sum_k float(wout[n][k]);
//            bias_out[n] = bias_in[n] + sum_k w32[n][k] * beta[k]
//   fold_ls: wout T [rows, ld] = T(w32[n][k] * ls[n]);  bias_out[n] = bias_in[n] * ls[n]   (ls may be null: plain convert)
// swiglu_h > 0: output row r takes source row (r % 64 < 32 ? 0 : swiglu_h) + 32 * (r / 64) + r % 32 (EPI_NORM_SWIGLU's order)
int launch_fold_ln(int dtype, const float* w32, int rows, int cols, int ld, const float* gamma, const float* beta,
                   const float* bias_in, void* wout, float* colsum, float* bias_out, hipStream_t stream, int swiglu_h = 0);
int launch_fold_ls(int dtype, const float* w32, int rows, int cols, int ld, const float* ls, const float* bias_in,
                   void* wout, float* bias_out, hipStream_t stream);
// prefix rows of the token stream (class token, then register tokens): prefix f32 [prefix_rows, dim] = the token values with
// their position-embedding rows already added (ap_vit_finalize).  fused path: x[img * tokens + j] = T(prefix[img * img_rows + j])
// and the rows' partial sums (img_rows = 0: one prefix shared by every image; img_rows = prefix_rows = 1: the exact f32 class
// rows [n, dim] rounded back into the stream); f32 path: tok[img * tokens + j] = prefix[j]
int launch_cls_stream(int dtype, const float* prefix, int prefix_rows, int img_rows, int n, int tokens, int dim, void* x, float* partial,
                      hipStream_t stream);
int launch_cls_init(float* tok, const float* prefix, int prefix_rows, int n, int tokens, int dim, hipStream_t stream);
// exact class rows: cls32[img] += branch[img] (f32 [n, dim] both), x[img * tokens] = T(cls32[img]) and that row's partial sums
int launch_cls_exact_update(int dtype, float* cls32, const float* branch, int n, int tokens, int dim, void* x, float* partial,
                            hipStream_t stream);
// out[m][j] = T(silu(x[m][j]) * x[m][h + j]), x: T [rows, 2h] dense, out: T [rows, h] dense (timm SwiGLUPacked; f32 math)
int launch_swiglu(int dtype, const void* x, int rows, int h, void* out, hipStream_t stream);
// prefix[j][:] = tokens[j][:] (+ pos[j][:] when pos != nullptr), f32
int launch_prefix_build(const float* cls, const float* reg, int reg_rows, const float* pos, int dim, float* prefix, hipStream_t stream);
int launch_convert(int dtype, const float* src, void* dst, size_t count, hipStream_t stream);
// f32 [count] -> per 32 values [hi 32 | lo 32] f16 (count % 32 == 0; the same number of bytes): GemmArgs::split's weight rows
int launch_split_f16_weights(const float* src, void* dst, size_t count, hipStream_t stream);
// [n,3,S,S] (f32 or T) -> patch rows T [n*g*g, ld]
int launch_chw_to_patchrows(int x_dtype, int dtype, const void* x, int n, int S, int ps, void* dst,
                            int ld, hipStream_t stream);

// fused float32 attention of the SAM2 trunk's image-wide blocks (sam2_attention.hip)
bool sattention_supports(int heads, int tq, int tk, int d, long ldq, long ldk, long ldv, long ldo);
int launch_sattention(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, int batch, int heads, int tq,
                      int tk, int d, float scale, float* out, long ldo, hipStream_t stream, bool exact);

// content statistics (content.hip): counts [n, 2] = (#gray < black_thresh, #(S < sat_thresh && V >= value_thresh))
int tile_content_counts(const uint8_t* tiles, int n, int h, int w, int black_thresh, int sat_thresh,
                        int value_thresh, unsigned* counts, hipStream_t stream);

// K1 (preproc.hip)
int get_norm_lut(const float mean[3], const float stdv[3], int dtype, hipStream_t stream,
                 const void** out);
int preproc_chw(const uint8_t* src, int n, int h, int w, int top, int left, int oh, int ow,
                const float mean[3], const float stdv[3], void* dst, int dtype, hipStream_t stream);
int preproc_patchrows(const uint8_t* src, int n, int h, int w, int top, int left, int oh, int ow,
                      int ps, const float mean[3], const float stdv[3], void* dst, int ld, int dtype,
                      hipStream_t stream);

}  // namespace ap
