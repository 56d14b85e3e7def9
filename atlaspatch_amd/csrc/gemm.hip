// MFMA GEMM for the ViT encoder on gfx950:  C[m][n] = sum_k A[m][k] * W[n][k]  (+ fused epilogue).
//
// Both operands are K-contiguous ("B^T" form), so every MFMA fragment is one 16-byte
// load.  Block tile 128(m) x 128(n), K-tile = 128 BYTES of K (64 f16/bf16, 32 f32),
// 256 threads = 4 waves in a 2x2 grid, each wave owns 64x64 = 2x2 MFMA tiles of 32x32.
//
//  * staging: global -> LDS by LDS-DMA (global_load_lds, 16 B/lane), double buffered.
//    The DMA writes lane-linear, so the bank swizzle is applied to the per-lane SOURCE
//    address and again on the fragment read:  chunk ^= (row >> 1) & 7  over the eight
//    16-byte chunks of a 128-byte tile row (conflict-free for ds_read_b128's 16-lane
//    groups: rows r, r+1 sit on different bank halves, (row>>1)&7 spreads 8 row pairs).
//  * operand roles are swapped (MFMA A-operand = weight rows, B-operand = activation
//    rows) so that each lane ends up with 4 CONSECUTIVE n for one m: the epilogue reads
//    bias/gamma/pos and the f32 residual as float4 and stores 8/16 bytes per lane.
//  * f32 mode uses v_mfma_f32_32x32x2_f32 (exact f32 fma chain); the k index is
//    permuted (lanes 0-31 take k 0..15 of the tile, lanes 32-63 take k 16..31) so a lane
//    reads its 16 floats as four 16-byte loads.  Any permutation of k is exact as long
//    as both operands use the same one.
//  * split-f16 mode (SPLIT, float32 buffers): the float32-ACCURATE fast path.  x = hi + lo * 2^-11 with hi = f16(x),
//    lo = f16((x - hi) * 2^11) carries 22 of f32's 24 mantissa bits (and keeps lo a NORMAL f16 for every x whose hi is
//    normal: no reliance on subnormal operands); a product keeps its three leading terms
//        w a  ~=  w_hi a_hi  +  2^-11 (w_hi a_lo + w_lo a_hi)
//    as three v_mfma_f32_32x32x16_f16 passes into TWO f32 accumulators (the 2^-11 terms are summed apart and folded
//    in once, after the K loop).  Weights are split once (ap_vit_finalize / ap_split_f16_weights) into rows of
//    [hi 32 | lo 32] f16 per 32 k -- the same 128 bytes per 32 k as the f32 row, so the staging plan is unchanged;
//    activations stay f32 in HBM and LDS and are split in registers after the fragment read.  24 MFMAs of 16 k per
//    32-deep K-tile and wave against 32 x 4 of the exact f32 instruction: 2.7x the arithmetic rate at ~2^-22 per product.
//  * XCD-aware tile order: block b runs on XCD b % 8; logical tile ids are remapped so
//    each XCD owns a contiguous run of tiles (all n-tiles of an m-panel share one L2).
//
// Roofline: MFMA (2*M*N*K flop); HBM traffic ~ A once + output once (weights L2-resident).
#include "ap_common.h"

namespace ap {
namespace {

constexpr int kTile = 128;            // BM = BN
constexpr int kRowBytes = 128;        // bytes of K per tile row
constexpr int kTileBytes = kTile * kRowBytes;   // 16 KiB per operand tile
constexpr int kBufBytes = 2 * kTileBytes;       // W tile + A tile

template <typename T> struct Mma;
template <> struct Mma<f16> {
    using Frag = f16x8;
    static __device__ __forceinline__ f32x16 run(Frag a, Frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<bf16> {
    using Frag = bf16x8;
    static __device__ __forceinline__ f32x16 run(Frag a, Frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};

__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

template <typename T> __device__ __forceinline__ void store4(T* p, f32x4 v);
template <> __device__ __forceinline__ void store4<float>(float* p, f32x4 v) { *(f32x4*)p = v; }
template <> __device__ __forceinline__ void store4<f16>(f16* p, f32x4 v) {
    f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
    *(f16x4*)p = h;
}
template <> __device__ __forceinline__ void store4<bf16>(bf16* p, f32x4 v) {
    bf16x4 h = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    *(bf16x4*)p = h;
}

// ---- fused-LayerNorm epilogues (twins of gemm256.hip's, bit for bit: a problem may be served by either kernel) ----
// A lane holds 4 consecutive n of one row (two packed pairs); its partner lane ^ 32 holds the next 4.  The persistent
// kernel sums a 16-byte chunk (8 columns) as ONE chain  s = dot2(c3, dot2(c2, dot2(c1, dot2(c0, 0))))  and then combines
// the 8 chunks of a 64-column group as ((s0+s1)+(s2+s3)) + ((s4+s5)+(s6+s7)); here the chain starts on the lower lane,
// crosses to the partner through one v_permlane32_swap and finishes there, and the tree is evaluated in that lane.
template <typename T> __device__ __forceinline__ u32x2 resid_add4(u32x2 d, u32x2 r);
template <> __device__ __forceinline__ u32x2 resid_add4<f16>(u32x2 d, u32x2 r) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    u32x2 y;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint32_t dk = d[k], rk = r[k];
        y[k] = __builtin_bit_cast(uint32_t, (h2)(__builtin_bit_cast(h2, dk) + __builtin_bit_cast(h2, rk)));
    }
    return y;
}
template <> __device__ __forceinline__ u32x2 resid_add4<bf16>(u32x2 d, u32x2 r) {
    u32x2 y;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float d0 = __builtin_bit_cast(float, d[k] << 16), d1 = __builtin_bit_cast(float, d[k] & 0xffff0000u);
        const float r0 = __builtin_bit_cast(float, r[k] << 16), r1 = __builtin_bit_cast(float, r[k] & 0xffff0000u);
        const bf16x4 c4 = {(bf16)(d0 + r0), (bf16)(d1 + r1), (bf16)0.0f, (bf16)0.0f};
        y[k] = __builtin_bit_cast(u32x2, c4)[0];
    }
    return y;
}
template <typename T> __device__ __forceinline__ void stats_chain4(u32x2 y, float& s, float& q);
template <> __device__ __forceinline__ void stats_chain4<f16>(u32x2 y, float& s, float& q) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint32_t yk = y[k];
        const h2 c = __builtin_bit_cast(h2, yk);
        s = __builtin_amdgcn_fdot2(c, h2{(_Float16)1.0f, (_Float16)1.0f}, s, false);
        q = __builtin_amdgcn_fdot2(c, c, q, false);
    }
}
template <> __device__ __forceinline__ void stats_chain4<bf16>(u32x2 y, float& s, float& q) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float c0 = __builtin_bit_cast(float, y[k] << 16), c1 = __builtin_bit_cast(float, y[k] & 0xffff0000u);
        s += c0 + c1;
        q = __builtin_fmaf(c1, c1, __builtin_fmaf(c0, c0, q));
    }
}
// value of the partner lane (lane ^ 32)
__device__ __forceinline__ float partner32(float v) {
    float a = v, b = v;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));     // a = {lo, lo}, b = {hi, hi}
    return (threadIdx.x & 32) ? a : b;
}
template <typename T> __device__ __forceinline__ u32x2 pack4t(f32x4 v);
template <> __device__ __forceinline__ u32x2 pack4t<f16>(f32x4 v) {
    f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
    return __builtin_bit_cast(u32x2, h);
}
template <> __device__ __forceinline__ u32x2 pack4t<bf16>(f32x4 v) {
    bf16x4 h = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    return __builtin_bit_cast(u32x2, h);
}
template <> __device__ __forceinline__ u32x2 pack4t<float>(f32x4) { return u32x2{0, 0}; }

// 8 f32 -> hi (f16, round to nearest) and lo = f16((x - hi) * 2^11): x = hi + lo * 2^-11 to ~2^-22 |x|
// (~3.7 VALU per value as hipcc compiles it; measured: a kernel with the split removed altogether is 10 % faster, with
// v_fma_mix_f32 for x - hi the same -- the loop is bound by its staging barrier, not by these, profiles/r06a_split_f16.txt)
__device__ __forceinline__ void split8(f32x4 x0, f32x4 x1, f16x8& hi8, f16x8& lo8) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f16 h0 = (f16)x0[j], h1 = (f16)x1[j];
        hi8[j] = h0; hi8[4 + j] = h1;
        lo8[j] = (f16)((x0[j] - (float)h0) * 2048.0f);
        lo8[4 + j] = (f16)((x1[j] - (float)h1) * 2048.0f);
    }
}

// window-order row -> image-order row of a [B, H, W] token grid cut into ws x ws windows (zero-padded at the right / bottom
// edge: hieradet.py window_partition), or -1 for a padding row.  ap_window_partition's layout.
__device__ __forceinline__ int window_row(int m, int ws, int H, int W, int nwy, int nwx) {
    const int t = ws * ws, w = m / t, r = m - w * t, iy = r / ws, ix = r - iy * ws;
    const int per = nwy * nwx, b = w / per, wi = w - b * per, wy = wi / nwx, wx = wi - wy * nwx;
    const int y = wy * ws + iy, x = wx * ws + ix;
    return (y < H && x < W) ? (b * H + y) * W + x : -1;
}

__device__ __forceinline__ void dma16(const char* gsrc, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <typename T, int EPI, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs g) {
    static_assert(!SPLIT || sizeof(T) == 4, "the split-f16 product runs on float32 buffers");
    __shared__ __attribute__((aligned(16))) char smem[2 * kBufBytes];   // 64 KiB

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave_n = wave >> 1, wave_m = wave & 1;
    const int hi = lane >> 5, l31 = lane & 31;

    // ---- XCD-aware logical tile id (bijective for any grid size)
    const int nblk = gridDim.x, b = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = b & 7, idx = b >> 3;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    // (split-f16 form only: N may be any multiple of 32 -- weight rows past N are staged from the last row and never stored)
    const int tiles_n = SPLIT ? (g.N + kTile - 1) / kTile : g.N / kTile;
    const int m0 = (lid / tiles_n) * kTile, n0 = (lid % tiles_n) * kTile;

    // ---- staging plan: wave-instruction s of this wave fills tile rows [8*(4*wave+s), +8)
    const char* srcW[4];
    const char* srcA[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int row = (wave * 4 + s) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        int am = m0 + row;
        am = am < g.M ? am : g.M - 1;
        int wn = n0 + row;
        if constexpr (SPLIT) wn = wn < g.N ? wn : g.N - 1;
        srcW[s] = (const char*)g.W + ((size_t)wn * g.ldw) * sizeof(T) + chunk * 16;
        srcA[s] = (const char*)g.A + ((size_t)am * g.lda) * sizeof(T) + chunk * 16;
        if constexpr (SPLIT) {
            if (g.win_mode == 1) {             // A is in image order: gather the window-order row, padding rows from the zero row
                const int r = window_row(am, g.win_ws, g.win_H, g.win_W, g.win_nwy, g.win_nwx);
                srcA[s] = (r >= 0 ? (const char*)g.A + ((size_t)r * g.lda) * sizeof(T) : (const char*)g.zero_row) + chunk * 16;
            }
        }
    }
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * kBufBytes + wave * 4096;
#pragma unroll
        for (int s = 0; s < 4; ++s) dma16(srcW[s] + (size_t)kt * kRowBytes, base + s * 1024);
#pragma unroll
        for (int s = 0; s < 4; ++s)
            dma16(srcA[s] + (size_t)kt * kRowBytes, base + kTileBytes + s * 1024);
    };

    // accumulators start from the bias (same convention as the 256x256 kernel: results are
    // bit-identical whichever kernel serves a problem)
    f32x16 acc[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int nb = n0 + wave_n * 64 + nt * 32 + g4 * 8 + hi * 4;
            const f32x4 b4 = !SPLIT || (nb < g.N && g.bias) ? *(const f32x4*)(g.bias + nb) : f32x4{0.f, 0.f, 0.f, 0.f};
            constexpr bool kNormInit = EPI == EPI_NORM_STORE || EPI == EPI_NORM_GELU || EPI == EPI_NORM_SWIGLU || EPI == EPI_NORM_QGELU;   // (these start from zero)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[nt][mt][g4 * 4 + e] = kNormInit ? 0.0f : b4[e];
        }

    f32x16 accl[SPLIT ? 2 : 1][SPLIT ? 2 : 1];                      // split-f16: the 2^-11 terms
    if constexpr (SPLIT) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int e = 0; e < 16; ++e) accl[nt][mt][e] = 0.0f;
    }

    const int xr = (l31 >> 1) & 7;                                  // row-dependent chunk xor
    const int rowW = (wave_n * 64 + l31) * kRowBytes;               // + nt * 32 rows
    const int rowA = kTileBytes + (wave_m * 64 + l31) * kRowBytes;  // + mt * 32 rows

    const int nk = g.K / (int)(kRowBytes / sizeof(T));
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* buf = smem + cur * kBufBytes;
        if constexpr (sizeof(T) == 2) {
            using Frag = typename Mma<T>::Frag;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int co = ((kk * 2 + hi) ^ xr) << 4;
                Frag w0 = *(const Frag*)(buf + rowW + co);
                Frag w1 = *(const Frag*)(buf + rowW + 32 * kRowBytes + co);
                Frag a0 = *(const Frag*)(buf + rowA + co);
                Frag a1 = *(const Frag*)(buf + rowA + 32 * kRowBytes + co);
                acc[0][0] = Mma<T>::run(w0, a0, acc[0][0]);
                acc[0][1] = Mma<T>::run(w0, a1, acc[0][1]);
                acc[1][0] = Mma<T>::run(w1, a0, acc[1][0]);
                acc[1][1] = Mma<T>::run(w1, a1, acc[1][1]);
            }
        } else if constexpr (SPLIT) {
            // split-f16: W row = [hi 32 | lo 32] f16, A row = 32 f32.  MFMA step s covers k = 16 s .. 16 s + 15, the lane's
            // eight k = 16 s + 8 hi ..: W chunks 2 s + hi (hi half) and 4 + 2 s + hi (lo half), A chunks 4 s + 2 hi, + 1
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int cw = ((2 * s + hi) ^ xr) << 4, cl = ((4 + 2 * s + hi) ^ xr) << 4;
                const int ca0 = ((4 * s + 2 * hi) ^ xr) << 4, ca1 = ((4 * s + 2 * hi + 1) ^ xr) << 4;
                f16x8 wh[2], wl[2], ah[2], al[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    wh[t] = *(const f16x8*)(buf + rowW + t * 32 * kRowBytes + cw);
                    wl[t] = *(const f16x8*)(buf + rowW + t * 32 * kRowBytes + cl);
                    split8(*(const f32x4*)(buf + rowA + t * 32 * kRowBytes + ca0),
                           *(const f32x4*)(buf + rowA + t * 32 * kRowBytes + ca1), ah[t], al[t]);
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[nt], ah[mt], acc[nt][mt], 0, 0, 0);
                        accl[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[nt], al[mt], accl[nt][mt], 0, 0, 0);
                        accl[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[nt], ah[mt], accl[nt][mt], 0, 0, 0);
                    }
            }
        } else {
            // f32: lane's 16 k-values = chunks 4*hi .. 4*hi+3 of its row (permuted k, see header)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int co = ((hi * 4 + c) ^ xr) << 4;
                f32x4 w0 = *(const f32x4*)(buf + rowW + co);
                f32x4 w1 = *(const f32x4*)(buf + rowW + 32 * kRowBytes + co);
                f32x4 a0 = *(const f32x4*)(buf + rowA + co);
                f32x4 a1 = *(const f32x4*)(buf + rowA + 32 * kRowBytes + co);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[e], a0[e], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[e], a1[e], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[e], a0[e], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[e], a1[e], acc[1][1], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    if constexpr (SPLIT) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[nt][mt][e] = __builtin_fmaf(accl[nt][mt][e], 1.0f / 2048.0f, acc[nt][mt][e]);
    }

    // ---- epilogue: lane owns m = .. + l31 and, per (nt, g4), n = .. + 8*g4 + 4*hi + {0..3}
    if constexpr (sizeof(T) == 2 && EPI == EPI_NORM_SWIGLU) {
        // the wave's two n blocks are x1 and x2 of the same 32 output columns (interleaved weight rows, see ap_common.h)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = m0 + wave_m * 64 + mt * 32 + l31;
            if (m >= g.M) continue;
            const float rstd = g.rowstats[2 * (size_t)m], nmr = g.rowstats[2 * (size_t)m + 1];
            const f32x2_t rs2 = {rstd, rstd}, nm2 = {nmr, nmr};
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                f32x4 y[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int n = n0 + wave_n * 64 + nt * 32 + g4 * 8 + hi * 4;
                    const f32x4 cs = *(const f32x4*)(g.colsum + n), bb = *(const f32x4*)(g.bias + n);
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[nt][mt][g4 * 4 + e];
                    const f32x2_t lo = __builtin_elementwise_fma(rs2, f32x2_t{v[0], v[1]},
                        __builtin_elementwise_fma(nm2, f32x2_t{cs[0], cs[1]}, f32x2_t{bb[0], bb[1]}));
                    const f32x2_t hi2 = __builtin_elementwise_fma(rs2, f32x2_t{v[2], v[3]},
                        __builtin_elementwise_fma(nm2, f32x2_t{cs[2], cs[3]}, f32x2_t{bb[2], bb[3]}));
                    y[nt] = f32x4{lo[0], lo[1], hi2[0], hi2[1]};
                }
                const f32x2_t a = swiglu2(f32x2_t{y[0][0], y[0][1]}, f32x2_t{y[1][0], y[1][1]});
                const f32x2_t b = swiglu2(f32x2_t{y[0][2], y[0][3]}, f32x2_t{y[1][2], y[1][3]});
                const int nout = ((n0 + wave_n * 64) >> 1) + g4 * 8 + hi * 4;
                store4<T>((T*)g.out + (size_t)m * (size_t)g.ldo + nout, f32x4{a[0], a[1], b[0], b[1]});
            }
        }
        return;
    }
    if constexpr (sizeof(T) == 2 && (EPI == EPI_NORM_STORE || EPI == EPI_NORM_GELU || EPI == EPI_NORM_QGELU)) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = m0 + wave_m * 64 + mt * 32 + l31;
            if (m >= g.M) continue;
            float rstd = g.rowstats[2 * (size_t)m], nmr = g.rowstats[2 * (size_t)m + 1];
            if constexpr (EPI == EPI_NORM_GELU) { rstd *= kGeluS; nmr *= kGeluS; }      // y * kGeluS for the GELU routine (twin of gemm256)
            const f32x2_t rs2 = {rstd, rstd}, nm2 = {nmr, nmr};
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int n = n0 + wave_n * 64 + nt * 32 + g4 * 8 + hi * 4;
                    const f32x4 cs = *(const f32x4*)(g.colsum + n);
                    f32x4 bb = *(const f32x4*)(g.bias + n);
                    if constexpr (EPI == EPI_NORM_GELU) bb *= kGeluS;
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[nt][mt][g4 * 4 + e];
                    const f32x2_t lo = __builtin_elementwise_fma(rs2, f32x2_t{v[0], v[1]},
                        __builtin_elementwise_fma(nm2, f32x2_t{cs[0], cs[1]}, f32x2_t{bb[0], bb[1]}));
                    const f32x2_t hi2 = __builtin_elementwise_fma(rs2, f32x2_t{v[2], v[3]},
                        __builtin_elementwise_fma(nm2, f32x2_t{cs[2], cs[3]}, f32x2_t{bb[2], bb[3]}));
                    v = f32x4{lo[0], lo[1], hi2[0], hi2[1]};
                    if constexpr (EPI == EPI_NORM_GELU) {
                        const f32x2_t a = gelu_sigmoid_poly2_s(f32x2_t{v[0], v[1]}), b = gelu_sigmoid_poly2_s(f32x2_t{v[2], v[3]});
                        v = f32x4{a[0], a[1], b[0], b[1]};
                    }
                    if constexpr (EPI == EPI_NORM_QGELU) {
                        const f32x2_t a = quick_gelu2(f32x2_t{v[0], v[1]}), b = quick_gelu2(f32x2_t{v[2], v[3]});
                        v = f32x4{a[0], a[1], b[0], b[1]};
                    }
                    store4<T>((T*)g.out + (size_t)m * (size_t)g.ldo + n, v);
                }
        }
        return;
    }
    if constexpr (sizeof(T) == 2 && (EPI == EPI_RESID_STATS || EPI == EPI_PATCH_STREAM)) {
        constexpr bool kPatch = EPI == EPI_PATCH_STREAM;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            int m = m0 + wave_m * 64 + mt * 32 + l31;
            const bool live = m < g.M;
            if (!live) m = g.M - 1;                              // every lane takes part in the lane exchanges
            size_t orow = (size_t)m;                             // output (stream) row
            const T* srow = (const T*)g.out + (size_t)m * (size_t)g.ldo;        // where the "residual" comes from
            if constexpr (kPatch) {                              // patch row m = img * P + p -> stream row img * (P + R) + R + p
                const int img = m / g.P, p = m - img * g.P;
                orow = (size_t)m + (size_t)(img + 1) * g.R;
                srow = (const T*)g.pos16 + (size_t)(g.pos_row0 + p) * g.N;
            }
            // exact class rows: the unrounded branch of row img * cls_tokens goes to cls_branch[img] as well
            float* cbr = nullptr;
            if constexpr (!kPatch) {
                if (g.cls_tokens > 0 && live) {
                    const int img = m / g.cls_tokens;
                    if (m == img * g.cls_tokens) cbr = g.cls_branch + (size_t)img * g.N;
                }
            }
            float cs8[8], cq8[8];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int n = n0 + wave_n * 64 + nt * 32 + g4 * 8 + hi * 4;
                    T* px = (T*)g.out + orow * (size_t)g.ldo + n;
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[nt][mt][g4 * 4 + e];
                    if constexpr (!kPatch) { if (cbr) *(f32x4*)(cbr + n) = v; }
                    const u32x2 y = resid_add4<T>(pack4t<T>(v), *(const u32x2*)(srow + n));
                    if (live) *(u32x2*)px = y;
                    // chain: lower lane (columns 0-3 of the chunk) first, then the partner continues with columns 4-7
                    float s = 0.f, q = 0.f;
                    if (hi == 0) stats_chain4<T>(y, s, q);
                    s = partner32(s);                            // upper lane now holds the lower lane's partial chain
                    q = partner32(q);
                    if (hi == 1) stats_chain4<T>(y, s, q);
                    cs8[nt * 4 + g4] = s;
                    cq8[nt * 4 + g4] = q;
                }
            if (live && hi == 1) {
                const float s = ((cs8[0] + cs8[1]) + (cs8[2] + cs8[3])) + ((cs8[4] + cs8[5]) + (cs8[6] + cs8[7]));
                const float q = ((cq8[0] + cq8[1]) + (cq8[2] + cq8[3])) + ((cq8[4] + cq8[5]) + (cq8[6] + cq8[7]));
                typedef float f32x2v __attribute__((ext_vector_type(2)));
                *(f32x2v*)(g.partial + (orow * (g.N >> 6) + ((n0 + wave_n * 64) >> 6)) * 2) = f32x2v{s, q};
            }
        }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = m0 + wave_m * 64 + mt * 32 + l31;
        if (m >= g.M) continue;
        size_t orow;
        const float* posrow = nullptr;
        if constexpr (EPI == EPI_PATCH_EMBED) {
            const int img = m / g.P, p = m - img * g.P;
            orow = ((size_t)img * (g.P + g.R) + g.R + p) * (size_t)g.ldo;
            posrow = g.pos + (size_t)(g.pos_row0 + p) * g.N;
        } else {
            orow = (size_t)m * (size_t)g.ldo;
        }
        int rrow = m;                          // row of the separate residual
        if constexpr (SPLIT) {
            if (g.win_mode == 2) {             // out / resid are in image order: scatter the window-order row, drop padding rows
                rrow = window_row(m, g.win_ws, g.win_H, g.win_W, g.win_nwy, g.win_nwx);
                if (rrow < 0) continue;
                orow = (size_t)rrow * (size_t)g.ldo;
            }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int n = n0 + wave_n * 64 + nt * 32 + g4 * 8 + hi * 4;
                if constexpr (SPLIT) { if (n >= g.N) continue; }
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[nt][mt][g4 * 4 + e];
                if constexpr (SPLIT && (EPI == EPI_BIAS_STORE || EPI == EPI_BIAS_GELU)) {
                    // separate residual (the SAM2 trunk's x = shortcut + f(x)): added AFTER the activation, GemmArgs::resid
                    if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                    }
                    if (g.resid) {
                        const f32x4 r = *(const f32x4*)(g.resid + (size_t)rrow * (size_t)g.ldr + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += r[e];
                    }
                    *(f32x4*)((float*)g.out + orow + n) = v;
                } else if constexpr (EPI == EPI_BIAS_STORE) {
                    if (g.gamma) {
                        const f32x4 ga = *(const f32x4*)(g.gamma + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= ga[e];
                    }
                    store4<T>((T*)g.out + orow + n, v);
                } else if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (sizeof(T) != 2) v[e] = gelu_erf(v[e]);
                    if constexpr (sizeof(T) == 2) {      // the same packed routine as gemm256: the two kernels stay bit-identical
                        const f32x2_t lo = gelu_sigmoid_poly2(f32x2_t{v[0], v[1]}), hi2 = gelu_sigmoid_poly2(f32x2_t{v[2], v[3]});
                        v = f32x4{lo[0], lo[1], hi2[0], hi2[1]};
                    }
                    store4<T>((T*)g.out + orow + n, v);
                } else if constexpr (EPI == EPI_BIAS_QGELU) {
                    if constexpr (sizeof(T) == 2) {      // the packed routine of gemm256 (bit-identical); float32: libm expf
                        const f32x2_t lo = quick_gelu2(f32x2_t{v[0], v[1]}), hi2 = quick_gelu2(f32x2_t{v[2], v[3]});
                        v = f32x4{lo[0], lo[1], hi2[0], hi2[1]};
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] / (1.0f + expf(-1.702f * v[e]));
                    }
                    store4<T>((T*)g.out + orow + n, v);
                } else if constexpr (EPI == EPI_BIAS_RESID) {
                    float* dst = (float*)g.out + orow + n;
                    f32x4 r = *(const f32x4*)dst;
                    if (g.gamma) {
                        const f32x4 ga = *(const f32x4*)(g.gamma + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= ga[e];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[e] += v[e];
                    *(f32x4*)dst = r;
                } else {
                    const f32x4 pe = *(const f32x4*)(posrow + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += pe[e];
                    *(f32x4*)((float*)g.out + orow + n) = v;
                }
            }
        }
    }
}

// float32 buffers, split-f16 products (GemmArgs::split; W = the [hi | lo] rows of launch_split_f16_weights)
int launch_split(int epilogue, const GemmArgs& a, hipStream_t stream) {
    const int tiles = ((a.M + kTile - 1) / kTile) * ((a.N + kTile - 1) / kTile);
    dim3 grid(tiles), block(256);
    switch (epilogue) {
        case EPI_BIAS_STORE: gemm_kernel<float, EPI_BIAS_STORE, true><<<grid, block, 0, stream>>>(a); break;
        case EPI_BIAS_GELU: gemm_kernel<float, EPI_BIAS_GELU, true><<<grid, block, 0, stream>>>(a); break;
        case EPI_BIAS_RESID: gemm_kernel<float, EPI_BIAS_RESID, true><<<grid, block, 0, stream>>>(a); break;
        case EPI_PATCH_EMBED: gemm_kernel<float, EPI_PATCH_EMBED, true><<<grid, block, 0, stream>>>(a); break;
        case EPI_BIAS_QGELU: gemm_kernel<float, EPI_BIAS_QGELU, true><<<grid, block, 0, stream>>>(a); break;
        default: set_error("gemm: epilogue %d has no split-f16 form", epilogue); return AP_ERR_INVALID;
    }
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

template <typename T>
int launch_typed(int epilogue, const GemmArgs& a, hipStream_t stream) {
    const int tiles = ((a.M + kTile - 1) / kTile) * (a.N / kTile);
    dim3 grid(tiles), block(256);
    switch (epilogue) {
        case EPI_BIAS_STORE: gemm_kernel<T, EPI_BIAS_STORE><<<grid, block, 0, stream>>>(a); break;
        case EPI_BIAS_GELU: gemm_kernel<T, EPI_BIAS_GELU><<<grid, block, 0, stream>>>(a); break;
        case EPI_BIAS_RESID: gemm_kernel<T, EPI_BIAS_RESID><<<grid, block, 0, stream>>>(a); break;
        case EPI_PATCH_EMBED: gemm_kernel<T, EPI_PATCH_EMBED><<<grid, block, 0, stream>>>(a); break;
        case EPI_NORM_STORE: gemm_kernel<T, EPI_NORM_STORE><<<grid, block, 0, stream>>>(a); break;
        case EPI_NORM_GELU: gemm_kernel<T, EPI_NORM_GELU><<<grid, block, 0, stream>>>(a); break;
        case EPI_NORM_SWIGLU: gemm_kernel<T, EPI_NORM_SWIGLU><<<grid, block, 0, stream>>>(a); break;
        case EPI_NORM_QGELU: gemm_kernel<T, EPI_NORM_QGELU><<<grid, block, 0, stream>>>(a); break;
        case EPI_BIAS_QGELU: gemm_kernel<T, EPI_BIAS_QGELU><<<grid, block, 0, stream>>>(a); break;
        case EPI_RESID_STATS: gemm_kernel<T, EPI_RESID_STATS><<<grid, block, 0, stream>>>(a); break;
        case EPI_PATCH_STREAM: gemm_kernel<T, EPI_PATCH_STREAM><<<grid, block, 0, stream>>>(a); break;
        default: set_error("gemm: unknown epilogue %d", epilogue); return AP_ERR_INVALID;
    }
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

}  // namespace

int launch_gemm(int dtype, int epilogue, const GemmArgs& a, hipStream_t stream) {
    return launch_gemm_impl(dtype, epilogue, a, 0, 0, stream);
}

int launch_gemm_impl(int dtype, int epilogue, const GemmArgs& a, int impl, int variant, hipStream_t stream) {
    AP_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty problem %d x %d x %d", a.M, a.N, a.K);
    AP_REQUIRE(impl == 0 || impl == 128 || impl == 256 || impl == 257, "gemm: unknown implementation %d", impl);
    AP_REQUIRE(!a.split || (dtype == AP_F32 && (impl == 0 || impl == 128)), "gemm: split-f16 products: float32 buffers, 128 x 128 kernel");
    if (impl == 257) {
#ifdef AP_WITH_TWIN
        return launch_gemm256_alt(dtype, epilogue, a, variant, stream);
#else
        set_error("gemm: impl 257 (the A/B twin with its ablation flags) is not part of the product library; "
                  "build it with `make -C atlaspatch_amd/csrc twin` and load libatlaspatch_hip_twin.so through ATLASPATCH_HIP_LIB");
        return AP_ERR_UNSUPPORTED;
#endif
    }
    const bool fused_epi = epilogue == EPI_NORM_STORE || epilogue == EPI_NORM_GELU || epilogue == EPI_NORM_SWIGLU || epilogue == EPI_NORM_QGELU ||
                           epilogue == EPI_RESID_STATS ||
                           epilogue == EPI_PATCH_STREAM;
    AP_REQUIRE(!fused_epi || dtype != AP_F32, "gemm: the fused-LayerNorm epilogues are f16 / bf16 only");
    AP_REQUIRE(!fused_epi || (epilogue == EPI_RESID_STATS ? a.partial != nullptr :
                              epilogue == EPI_PATCH_STREAM ? (a.partial && a.pos16 && a.P > 0 && a.R > 0) : (a.colsum && a.rowstats)),
               "gemm: missing operand for the fused-LayerNorm epilogue %d", epilogue);
    // Kernel choice (results are bit-identical either way).  The persistent 256 x 256 kernel needs about one tile per CU to
    // pay: with few row tiles and a narrow N (proj / fc2 of a 32-tile extract_batch: 75 tiles for 256 CUs) the 128 x 128
    // kernel's four times as many workgroups finish sooner.
    static const int num_cu = [] {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }();
    const long tiles256 = (long)((a.M + 255) / 256) * (a.N / 256);
    const bool few_tiles = tiles256 * 2 <= num_cu && a.N % kTile == 0;
    if (impl == 256 || (impl == 0 && a.M >= 256 && !few_tiles && gemm256_supports(dtype, epilogue, a)))
        return launch_gemm256(dtype, epilogue, a, variant, stream);
    const int kt = kRowBytes / (int)dtype_size(dtype);
    AP_REQUIRE(a.split ? a.N % 32 == 0 : a.N % kTile == 0, "gemm: N=%d must be a multiple of %d", a.N, a.split ? 32 : kTile);
    AP_REQUIRE(a.resid == nullptr || (a.split && (epilogue == EPI_BIAS_STORE || epilogue == EPI_BIAS_GELU) && a.ldr >= a.N && a.ldr % 4 == 0),
               "gemm: a separate residual is an option of the split-f16 form's bias / GELU epilogues");
    AP_REQUIRE(a.K % kt == 0, "gemm: K=%d must be a multiple of %d for this dtype", a.K, kt);
    AP_REQUIRE(a.lda >= a.K && a.ldw >= a.K, "gemm: leading dimensions smaller than K");
    AP_REQUIRE(((size_t)a.lda * dtype_size(dtype)) % 16 == 0 && ((size_t)a.ldw * dtype_size(dtype)) % 16 == 0,
               "gemm: row strides must be 16-byte multiples");
    if (a.split) {
        AP_REQUIRE(dtype == AP_F32, "gemm: the split-f16 product runs on float32 buffers");
        return launch_split(epilogue, a, stream);
    }
    switch (dtype) {
        case AP_F16: return launch_typed<f16>(epilogue, a, stream);
        case AP_BF16: return launch_typed<bf16>(epilogue, a, stream);
        case AP_F32: return launch_typed<float>(epilogue, a, stream);
    }
    set_error("gemm: unknown dtype %d", dtype);
    return AP_ERR_INVALID;
}

}  // namespace ap
