// MFMA fragment / packing / row-statistics helpers shared by the persistent GEMM kernels (gemm256.hip, gemm_duo.hip).
#pragma once
#include "ap_common.h"

namespace ap {
namespace {

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

template <typename T> __device__ __forceinline__ u32x2 pack4(f32x4 v);
template <> __device__ __forceinline__ u32x2 pack4<f16>(f32x4 v) {
    f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
    return __builtin_bit_cast(u32x2, h);
}
template <> __device__ __forceinline__ u32x2 pack4<bf16>(f32x4 v) {
    bf16x4 h = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    return __builtin_bit_cast(u32x2, h);
}

__device__ __forceinline__ float dpp_add(float v, float w) { return v + w; }
#define AP_DPP_F32(V, CTRL) __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (V)), (CTRL), 0xF, 0xF, true))
// sum over the 8 lanes that share (lane >> 3): xor-1, xor-2 (quad permutes), mirror within 8 -- every lane gets the total
__device__ __forceinline__ float sum8(float v) {
    v += AP_DPP_F32(v, 0xB1);
    v += AP_DPP_F32(v, 0x4E);
    v += AP_DPP_F32(v, 0x141);
    return v;
}

// EPI_RESID_STATS element step on 8 packed values: y = T(d + r) (d = the branch output already rounded to T, r = the
// stream), s += sum(y), q += sum(y^2) in f32.
template <typename T> __device__ __forceinline__ u32x4 resid_add_stats(u32x4 d, u32x4 r, float& s, float& q);
template <> __device__ __forceinline__ u32x4 resid_add_stats<f16>(u32x4 d, u32x4 r, float& s, float& q) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    u32x4 y;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t dk = d[k], rk = r[k];     // (bit_cast of a vector-element lvalue reads element 0: copy first)
        const h2 c = __builtin_bit_cast(h2, dk) + __builtin_bit_cast(h2, rk);           // v_pk_add_f16: correctly rounded
        y[k] = __builtin_bit_cast(uint32_t, c);
        s = __builtin_amdgcn_fdot2(c, h2{(_Float16)1.0f, (_Float16)1.0f}, s, false);
        q = __builtin_amdgcn_fdot2(c, c, q, false);
    }
    return y;
}
template <> __device__ __forceinline__ u32x4 resid_add_stats<bf16>(u32x4 d, u32x4 r, float& s, float& q) {
    u32x4 y;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float d0 = __builtin_bit_cast(float, d[k] << 16), d1 = __builtin_bit_cast(float, d[k] & 0xffff0000u);
        const float r0 = __builtin_bit_cast(float, r[k] << 16), r1 = __builtin_bit_cast(float, r[k] & 0xffff0000u);
        const bf16x4 c4 = {(bf16)(d0 + r0), (bf16)(d1 + r1), (bf16)0.0f, (bf16)0.0f};
        const u32x2 cc = __builtin_bit_cast(u32x2, c4);
        y[k] = cc[0];
        const float c0 = __builtin_bit_cast(float, cc[0] << 16), c1 = __builtin_bit_cast(float, cc[0] & 0xffff0000u);
        s += c0 + c1;
        q = __builtin_fmaf(c1, c1, __builtin_fmaf(c0, c0, q));
    }
    return y;
}


#define AP_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

struct TileWalk {          // this workgroup's tile list: ids first, first + stride, ... (< end)
    int first, stride, count, tiles_n, tiles_m, group;
    // id -> (row tile, column tile): column groups of `group` tiles, row-major inside a group, so the
    // workgroups of an XCD (consecutive ids) form a (workgroups / group) x group block of tiles.
    __device__ __forceinline__ void rc(int id, int& tr, int& tc) const {
        const int per = group * tiles_m;
        const int grp = id / per, rem = id - grp * per;
        const int left = tiles_n - grp * group;
        const int width = left < group ? left : group;
        tr = rem / width;
        tc = grp * group + (rem - tr * width);
    }
};

}  // namespace
}  // namespace ap
