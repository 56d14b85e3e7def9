// Fused float32 attention of the SAM2 Hiera trunk -- the image-wide ("global") blocks and, batched over windows, the
// windowed ones:
//     out[b][h] = softmax(q[b][h] k[b][h]^T * scale) v[b][h]     q: [batch * tq, ldq], k / v: [batch * tk, ld*] float32 rows,
//                                                              head h at column h * d, window b at rows b * tq / b * tk
// (sam2/modeling/backbones/hieradet.py MultiScaleAttention -> F.scaled_dot_product_attention, as driven by
// services/segmentation.py:120-140).  At 64 x 64 tokens and 4 heads of 96 channels the unfused chain writes, re-reads,
// normalises and reads again a 4 x 4096 x 4096 float32 score matrix (268 MB, 1.07 GB of traffic per block) around two
// batched GEMMs of awkward shape (N = 96); here the scores never leave registers.
//
// Exact-f32 MFMA (v_mfma_f32_32x32x2_f32: 64 cycles each), so the kernel is MFMA-bound by a wide margin and everything
// else is kept simple:
//   * workgroup = 4 waves = ONE block of 32 queries of one head; the KEYS are split four ways (wave w takes key tiles
//     w, w + 4, ...), each wave runs an online softmax over its tiles and the four partial results (running max, sum,
//     32 x d accumulator) are merged through LDS at the end.  512 workgroups for the 4096-token blocks: two per CU.
//   * operands go from global memory straight into MFMA fragment layout, no LDS staging: per 32-key tile a lane loads
//     its K fragment as d / 8 16-byte vectors (row = key, k slots 8 j + 4 hi + e on BOTH operands: any permutation of
//     the contraction index is exact) and its V fragment as 16 * d / 32 dwords (lane = channel: 128 contiguous bytes per
//     half-wave).  A tile is 2 * 16 * d / 32 MFMAs = 6144 cycles at d = 96, against which ~60 load instructions are noise.
//     K of the next tile is requested as soon as this tile's scores are done, V of the next tile after this tile's
//     second product: each operand's latency hides under the other product's MFMAs with a single register set each.
//   * S^T = K Q^T puts a query in a lane (column) and 16 keys in its accumulator registers; the second product
//     O^T = V^T P^T takes exactly those registers as its B operand (k slot of step s = accumulator s), so P never moves.
// Softmax in the log2 domain (scale * log2(e) folded into the exponent), float32 throughout.
//
// Split-f16 form (SPLIT, round 6; the default -- ap_sattention_f32's `exact` argument selects the chain above): every operand
// x = hi + lo with hi = f16(x), lo = f16(x - hi) (22 of 24 mantissa bits; the f16 MFMA takes subnormal operands exactly --
// tools/isa_probes/mfma_f16_denorm.hip -- so a small lo only loses absolute precision below 2^-25) and both products as
//     a b ~= a_hi b_hi + a_hi b_lo + a_lo b_hi        three v_mfma_f32_32x32x16_f16 into the SAME f32 accumulator
// 36 MFMAs of 32 cycles per 32-key tile at d = 96 instead of 96 of 64 cycles; the split costs ~3 VALU per value.  Same
// loads, same fragment assignment (a lane's eight k slots of a 16-deep step are two of its 16-byte vectors), same softmax.
//
// Roofline: f32 MFMA, 4 * tq * tk * d flop per head (split form: three times that on the f16 MFMA).
#include "ap_common.h"

namespace ap {
namespace {

__device__ __forceinline__ float sa_partner_max(float v) {
    float a = v, b = v;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}
__device__ __forceinline__ float sa_partner_sum(float v) {
    float a = v, b = v;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}

// 8 f32 -> f16 hi and f16 lo = f16(x - hi), unscaled
__device__ __forceinline__ void sa_split8(f32x4 x0, f32x4 x1, f16x8& hi8, f16x8& lo8) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f16 h0 = (f16)x0[j], h1 = (f16)x1[j];
        hi8[j] = h0; hi8[4 + j] = h1;
        lo8[j] = (f16)(x0[j] - (float)h0);
        lo8[4 + j] = (f16)(x1[j] - (float)h1);
    }
}
__device__ __forceinline__ f32x16 sa_mma3(f16x8 ah, f16x8 al, f16x8 bh, f16x8 bl, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);          // the small terms first
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
}

struct SAttnArgs {
    const float *q, *k, *v;
    float* out;
    long ldq, ldk, ldv, ldo;
    int tq, tk;
    float scale_log2e;
};

// DB = d / 32 (1..3)
template <int DB, bool SPLIT>
__global__ __launch_bounds__(256, 2) void sattention_kernel(SAttnArgs a) {
    constexpr int D = DB * 32;
    constexpr int KJ = D / 8;                       // 16-byte K / Q vectors per lane
    __shared__ __attribute__((aligned(16))) float red[4][32][D + 4];       // partial O (per wave), + (m, l); rows stay 16-byte aligned

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int head = blockIdx.y;
    const int q0 = blockIdx.x * 32;                 // first query of this block inside its window
    const int win = blockIdx.z;                     // window (batch entry): rows win * tq .. / win * tk ..
    const int qrow = q0 + l31 < a.tq ? q0 + l31 : a.tq - 1;            // rows past the end recompute the last one (never stored)
    const float* qp = a.q + ((size_t)win * a.tq + qrow) * a.ldq + head * D + 4 * hi;
    const float* kb = a.k + (size_t)win * a.tk * a.ldk + (size_t)head * D + 4 * hi;
    const float* vb = a.v + (size_t)win * a.tk * a.ldv + (size_t)head * D + l31;
    const int klast = a.tk - 1;

    // The wave's Q rows live in LDS (rows padded to D + 4 floats: conflict-free 16-byte reads), not in 48 registers: one
    // ds_read_b128 per four MFMAs is nothing next to 64-cycle MFMAs, and the register file holds K, V, S and O without spills.
    // The buffer is the first wave-slice of `red`, which is not written before every wave has finished its key loop.
    float (*qs)[D + 4] = red[0];
    if (wave == 0) {
#pragma unroll
        for (int j = 0; j < KJ; ++j) *(f32x4*)&qs[l31][8 * j + 4 * hi] = *(const f32x4*)(qp + 8 * j);
    }
    __syncthreads();

    const int ntiles = (a.tk + 31) >> 5;            // 32-key tiles (the last one may be ragged); wave w takes w, w + 4, ...
    f32x4 kf[KJ];
    float vf[DB][16];
    auto load_k = [&](int t) {
        int key = t * 32 + l31;
        key = key < klast ? key : klast;                         // keys past the end: last row's data, masked below
        const float* p = kb + (size_t)key * a.ldk;
#pragma unroll
        for (int j = 0; j < KJ; ++j) kf[j] = *(const f32x4*)(p + 8 * j);
    };
    auto load_v = [&](int t) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            int key = t * 32 + 8 * (s >> 2) + 4 * hi + (s & 3);
            key = key < klast ? key : klast;
            const float* p = vb + (size_t)key * a.ldv;
#pragma unroll
            for (int db = 0; db < DB; ++db) vf[db][s] = p[db * 32];
        }
    };

    float m_run = -INFINITY, l_run = 0.f;
    f32x16 ot[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) ot[db][e] = 0.f;

    int t = wave;
    if (t < ntiles) { load_k(t); load_v(t); }
    for (; t < ntiles; t += 4) {
        // ---------------- S^T = K Q^T (32 keys x 32 queries)
        f32x16 st;
#pragma unroll
        for (int e = 0; e < 16; ++e) st[e] = 0.f;
        if constexpr (SPLIT) {
#pragma unroll
            for (int j = 0; j < KJ; j += 2) {                     // one 16-deep step = the lane's vectors j, j + 1 on both operands
                f16x8 kh, kl, qh, ql;
                sa_split8(kf[j], kf[j + 1], kh, kl);
                sa_split8(*(const f32x4*)&qs[l31][8 * j + 4 * hi], *(const f32x4*)&qs[l31][8 * j + 8 + 4 * hi], qh, ql);
                st = sa_mma3(kh, kl, qh, ql, st);
            }
        } else {
#pragma unroll
            for (int j = 0; j < KJ; ++j) {
                const f32x4 qv = *(const f32x4*)&qs[l31][8 * j + 4 * hi];
#pragma unroll
                for (int e = 0; e < 4; ++e) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j][e], qv[e], st, 0, 0, 0);
            }
        }
        if (t + 4 < ntiles) load_k(t + 4);                      // flies under the softmax and the second product
        if (t * 32 + 32 > a.tk) {                               // ragged last tile: keys past the end score -inf
            const int lim = a.tk - t * 32 - 4 * hi;
#pragma unroll
            for (int e = 0; e < 16; ++e) st[e] = (e & 3) + 8 * (e >> 2) < lim ? st[e] : -INFINITY;
        }
        // ---------------- online softmax (log2 domain)
        float mx = st[0];
#pragma unroll
        for (int e = 1; e < 16; ++e) mx = fmaxf(mx, st[e]);
        mx = sa_partner_max(mx) * a.scale_log2e;
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);      // first tile: exp2(-inf) = 0
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[e], a.scale_log2e, -m_new));
            st[e] = p;
            psum += p;
        }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) ot[db][e] *= alpha;
        // ---------------- O^T += V^T P^T
        if constexpr (SPLIT) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {                         // keys 16 u ..: accumulator / V registers 8 u .. 8 u + 7
                f16x8 ph, pl;
                sa_split8(f32x4{st[8 * u], st[8 * u + 1], st[8 * u + 2], st[8 * u + 3]},
                          f32x4{st[8 * u + 4], st[8 * u + 5], st[8 * u + 6], st[8 * u + 7]}, ph, pl);
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    f16x8 vh, vl;
                    sa_split8(f32x4{vf[db][8 * u], vf[db][8 * u + 1], vf[db][8 * u + 2], vf[db][8 * u + 3]},
                              f32x4{vf[db][8 * u + 4], vf[db][8 * u + 5], vf[db][8 * u + 6], vf[db][8 * u + 7]}, vh, vl);
                    ot[db] = sa_mma3(vh, vl, ph, pl, ot[db]);
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int db = 0; db < DB; ++db) ot[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[db][s], st[s], ot[db], 0, 0, 0);
        }
        if (t + 4 < ntiles) load_v(t + 4);                      // flies under the next tile's first product
    }
    l_run = sa_partner_sum(l_run);
    __syncthreads();                                            // every wave is done with the Q rows in red[0]

    // ---------------- merge the four key quarters: O = sum_w O_w 2^(m_w - m) / sum_w l_w 2^(m_w - m)
    // lane (hi, query l31) holds channels db * 32 + 8 (i / 4) + 4 hi + i % 4 of its query
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
            const f32x4 o4 = {ot[db][i], ot[db][i + 1], ot[db][i + 2], ot[db][i + 3]};
            *(f32x4*)&red[wave][l31][db * 32 + 8 * (i >> 2) + 4 * hi] = o4;
        }
    if (hi == 0) { red[wave][l31][D] = m_run; red[wave][l31][D + 1] = l_run; }
    __syncthreads();
    // thread -> (query = tid / 8, 8 threads share a row: channel chunks of 4 floats, strided by 32 floats)
    const int qi = threadIdx.x >> 3, c0 = (threadIdx.x & 7) * 4;
    float mw[4], m_all = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) { mw[w] = red[w][qi][D]; m_all = fmaxf(m_all, mw[w]); }
    float sc[4], l_all = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        sc[w] = mw[w] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mw[w] - m_all);
        l_all += red[w][qi][D + 1] * sc[w];
    }
    const float inv = 1.0f / l_all;
    if (q0 + qi >= a.tq) return;
    float* op = a.out + ((size_t)win * a.tq + q0 + qi) * a.ldo + head * D;
#pragma unroll
    for (int c = c0; c < D; c += 32) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += red[w][qi][c + e] * sc[w];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] *= inv;
        *(f32x4*)(op + c) = acc;
    }
}

}  // namespace

bool sattention_supports(int heads, int tq, int tk, int d, long ldq, long ldk, long ldv, long ldo) {
    return heads > 0 && (d == 32 || d == 64 || d == 96) && tq > 0 && tk > 0 && ldq % 4 == 0 && ldk % 4 == 0 && ldo % 4 == 0 && ldv > 0;
}

int launch_sattention(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, int batch, int heads, int tq,
                      int tk, int d, float scale, float* out, long ldo, hipStream_t stream, bool exact) {
    AP_REQUIRE(q && k && v && out && batch > 0 && batch <= 65535, "sattention: bad arguments");
    AP_REQUIRE(sattention_supports(heads, tq, tk, d, ldq, ldk, ldv, ldo),
               "sattention: unsupported shape (d 32 / 64 / 96, row strides multiples of 4 floats)");
    AP_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)out) & 15) == 0, "sattention: q / k / out must be 16-byte aligned");
    SAttnArgs a{q, k, v, out, ldq, ldk, ldv, ldo, tq, tk, scale * 1.4426950408889634f};
    dim3 grid((tq + 31) / 32, heads, batch), block(256);
    if (exact) {
        switch (d / 32) {
            case 1: sattention_kernel<1, false><<<grid, block, 0, stream>>>(a); break;
            case 2: sattention_kernel<2, false><<<grid, block, 0, stream>>>(a); break;
            default: sattention_kernel<3, false><<<grid, block, 0, stream>>>(a); break;
        }
    } else {
        switch (d / 32) {
            case 1: sattention_kernel<1, true><<<grid, block, 0, stream>>>(a); break;
            case 2: sattention_kernel<2, true><<<grid, block, 0, stream>>>(a); break;
            default: sattention_kernel<3, true><<<grid, block, 0, stream>>>(a); break;
        }
    }
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

}  // namespace ap
