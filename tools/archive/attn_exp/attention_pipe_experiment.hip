// EXPERIMENT (round 4, measured and dropped -- DESIGN.md section 7): attention_flash.hip plus a pipelined head-width-64 kernel
// (fixed row maximum, S^T of the next block beside the softmax of this one, redo vote) with timing-ablation flags.  Not built into
// the product library; tools/attn_exp/build.sh + tools/attn_ablate.py time it.
// Tiled ("flash") multi-head attention for f16 / bf16, heads stored 64 or 128 wide, any sequence length:
//     out = softmax(q k^T * scale) v      per (image, head), f32 softmax, online rescaling.
// (The text below describes the 64-wide instantiation; HD = 128 -- vit_h_14's 80-wide heads, zero-padded -- has 256-byte
// rows: two DMA pieces per thread, operand and tile, K chunks XOR-swizzled by row & 15, V 64-byte windows by row & 3, four
// 32-channel output blocks, 128 KiB of LDS.)
//
// One workgroup = 8 waves = eight 32-query blocks of one (image, head); it walks the keys in tiles
// of 64.  K and V tiles arrive by LDS-DMA (global_load_lds, 16 B per lane, one K and one V piece
// per thread and tile) into a ring of four 16-KiB buffers, three tiles ahead of the one being
// consumed (the kernel is bound by memory-level parallelism, not by bandwidth or MFMA rate: the
// whole K / V of a 197-token head is in flight before the first tile is touched); both stay
// ROW-major in LDS:
//   * S^T = K Q^T: MFMA A-operand = K rows (ds_read_b128, rows XOR-swizzled on the DMA source
//     address), B-operand = the wave's Q rows held in registers.  A lane owns one query and half
//     of the tile's keys, its partner lane ^ 32 the other half: row max / row sum are in-register
//     reductions plus one v_permlane32_swap.
//   * O^T = V^T P^T: the A-operand needs 8 keys of ONE channel per lane, i.e. a column of the
//     row-major V tile: it is read with the gfx950 transpose load ds_read_b64_tr_b16 (a 16-lane
//     group turns a [4 keys][16 channels] block into 4 keys of its own channel per lane), two per
//     fragment.  The k slots of this product are assigned to exactly the keys the S^T accumulators
//     of the lane already hold, so P goes from accumulator to B-operand without leaving the lane.
//     V rows are swizzled (16-B chunk ^= 4 * ((row >> 1) & 1)) so the four rows a half-wave reads
//     fall on four different 64-byte bank groups.
// Online softmax per tile: m' = max(m, rowmax), O *= 2^((m - m') c), l = l * 2^((m - m') c) + rowsum
// with c = log2(e) / 8 folded into the exponent; the rescale is deferred while no row of the wave grew
// by more than 2^8.  Keys past the end are staged from the last valid row (finite data) and masked to -inf.
// One workgroup barrier per key tile: tile j + 3 is staged right BEHIND the barrier that opens iteration j (its
// buffer held tile j - 1, which every wave has left by then).  The output block is transposed through the
// wave's 4 KiB of a free ring buffer and leaves as whole 128-byte rows (16 B per lane).  Workgroups that share a
// head (more than 256 queries) are placed on one XCD so that its L2 serves their common K / V stream.
//
// What paces it (PMC + ablations): at head_dim 64 a wave issues ~170 VALU slots (33 v_exp at
// ~5/3, 32 fma, 32 add, 16 cvt_pk, 24 max; packed f32 forms take two slots, so they save nothing) beside 16 MFMAs
// of 8 slots each, and on one SIMD these slots add up rather than overlap: ~300 slots per 32 x 64 score block.
//
// Roofline: MFMA (4 * T * T * 64 flop per head; padded to 32-query x 64-key tiles);
// HBM traffic = q, k, v read once + o written once.
#include "../../atlaspatch_amd/csrc/ap_common.h"

namespace ap {
namespace {

#ifndef AP_PIPE_ABL
#define AP_PIPE_ABL 0                   // timing ablations of the pipelined kernel (tools/attn_ablate.py builds them; results invalid)
#endif
constexpr int kKV = 64;                 // keys per tile
constexpr int kNW = 8;                  // waves per workgroup
constexpr int kNB = 4;                  // K/V ring buffers (prefetch distance kNB - 1)

template <typename T> struct FMma;
template <> struct FMma<f16> {
    using Frag = f16x8;
    static __device__ __forceinline__ f32x16 run(Frag a, Frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct FMma<bf16> {
    using Frag = bf16x8;
    static __device__ __forceinline__ f32x16 run(Frag a, Frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};

// one LDS-DMA load (16 B per lane): LDS[lds_dst + lane * 16] <- base[off]; only s_mov / s_nop besides
// the load, so SCC is untouched; M0 saved and restored (compiler-reserved)
__device__ __forceinline__ void dma16(const char* base, uint32_t off, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %3\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(off), "s"(lds_dst), "s"(base)
        : "memory");
}

template <int OFF> __device__ __forceinline__ u32x2 tr_read(uint32_t lds_addr) {       // immediate offset: no address VALU
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF) : "memory");
    return v;
}
template <int V> struct IntC { static constexpr int value = V; };

// v_permlane32_swap exchanges the upper half of its first operand with the lower half of its second:
// fed two copies of v it leaves {v.lo, v.lo} and {v.hi, v.hi}.  Inline asm on two distinct registers
// (through the builtin hipcc folded the two results of equal inputs into one); the two v_nop are the
// wait states between a VALU write of an operand and the swap reading it.
__device__ __forceinline__ void half_swap(float v, float& lo, float& hi) {
    float a = v, b = v;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    lo = a;
    hi = b;
}
__device__ __forceinline__ float half_swap_max(float v) {       // max(v[lane & 31], v[(lane & 31) + 32])
    float lo, hi;
    half_swap(v, lo, hi);
    return fmaxf(lo, hi);
}
__device__ __forceinline__ float half_swap_sum(float v) {
    float lo, hi;
    half_swap(v, lo, hi);
    return lo + hi;
}

// The tiled kernel's body for ONE (image, head, part): a device function so that the pipelined kernel below can fall back to it.
template <typename T, int HD>
__device__ __forceinline__ void flash_unit(char* smem, const T* __restrict__ qkv, T* __restrict__ out, int tokens, int heads, int parts,
                                           int units, float scale) {
    constexpr int kHD = HD;
    constexpr int RB = HD * 2;                       // bytes of one K / V row of a head
    constexpr int kTileBytes = kKV * RB;             // one K or V tile (64 rows)
    constexpr int NKK = HD / 16;                     // k-steps of S^T = K Q^T
    constexpr int NIT = HD / 32;                     // 32-channel output blocks
    constexpr int NPC = kTileBytes / (kNW * 64 * 16);   // DMA pieces per thread, operand and tile (1 or 2)
    constexpr int CPR = RB / 16;                     // 16-byte chunks per row (8 or 16)
    using Frag = typename FMma<T>::Frag;

    // opaque to the optimiser: when this body is the redo path of the pipelined kernel its lane constants must not be shared
    // with (and kept alive across) the pipelined loop
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    // XCD-aware walk: workgroup b runs on XCD b % 8 (each XCD has its own L2), so the `parts` workgroups that
    // share one (image, head) -- every one of them streams the head's whole K and V -- take consecutive slots of
    // ONE XCD: they run side by side and K / V come from HBM once instead of `parts` times (785 tokens: 4 parts)
    const int slot = blockIdx.x >> 3;
    const int unit = (slot / parts) * 8 + (blockIdx.x & 7), part = slot % parts;
    if (unit >= units) return;
    const int img = unit / heads, head = unit - img * heads;
    const int dim = heads * kHD;
    const uint32_t ldb = (uint32_t)(3 * dim) * 2;                         // row stride in bytes
    const char* base = (const char*)(qkv + (size_t)img * tokens * 3 * dim + head * kHD);
    const char* kbase = base + (size_t)dim * 2;
    const char* vbase = base + (size_t)dim * 4;
    const int nkv = (tokens + kKV - 1) / kKV;

    // ---- staging plan: piece pc of thread tid is the 16-byte LDS chunk (pc * 512 + tid) of the tile, i.e. tile row
    //      idx / CPR, chunk idx % CPR; the SOURCE chunk is swizzled (K: by row pairs at 128-byte rows, by row & 15 at
    //      256-byte rows; V: 64-byte windows)
    uint32_t kchunk[NPC], vchunk[NPC];
    int srow[NPC];
#pragma unroll
    for (int pc = 0; pc < NPC; ++pc) {
        const int idx = pc * (kNW * 64) + tid;
        const int row = idx / CPR, spos = idx % CPR;
        srow[pc] = row;
        if constexpr (HD == 64) {
            kchunk[pc] = (uint32_t)((spos ^ ((row >> 1) & 7)) << 4);
            vchunk[pc] = (uint32_t)((spos ^ (((row >> 1) & 1) << 2)) << 4);
        } else {
            kchunk[pc] = (uint32_t)((spos ^ (row & 15)) << 4);
            vchunk[pc] = (uint32_t)((spos ^ ((row & 3) << 2)) << 4);
        }
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 1024;
    auto stage = [&](int j) {
#pragma unroll
        for (int pc = 0; pc < NPC; ++pc) {
            int row = j * kKV + srow[pc];
            row = row < tokens ? row : tokens - 1;
            const uint32_t roff = (uint32_t)row * ldb;
            const uint32_t dst = lds0 + (j % kNB) * 2 * kTileBytes + pc * (kNW * 64 * 16);
            dma16(kbase, roff + kchunk[pc], dst);
            dma16(vbase, roff + vchunk[pc], dst + kTileBytes);
        }
    };
#pragma unroll
    for (int j = 0; j < kNB - 1; ++j)
        if (j < nkv) stage(j);

    // ---- this wave's queries
    // the 32-query blocks are dealt evenly to the parts (25 blocks -> 7, 6, 6, 6 instead of 8, 8, 8, 1)
    const int nqb_all = (tokens + 31) / 32;
    const int qcount = nqb_all / parts + (part < nqb_all % parts);
    const int qb = part * (nqb_all / parts) + (part < nqb_all % parts ? part : nqb_all % parts) + wave;
    int qrow = qb * 32 + l31;
    const bool qvalid = qrow < tokens;
    if (!qvalid) qrow = tokens - 1;
    Frag qf[NKK];
    {
        const T* qp = (const T*)(base + (size_t)qrow * ldb);
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) qf[kk] = *(const Frag*)(qp + kk * 16 + hi * 8);
    }
    // The Q registers are "used" HERE, in front of the tile loop: hipcc's wait-count pass then puts its vmcnt(0) for
    // these four loads at this point.  Without it the wait sits in front of their first real use -- the QK^T MFMAs
    // INSIDE the loop (the pass cannot prove that an earlier iteration already waited) -- and, because the LDS-DMA
    // stream is invisible to the pass, that vmcnt(0) drained every staged tile in every iteration: the tile staged
    // a few instructions earlier was waited for at once and the three-tile prefetch never overlapped anything.
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) asm volatile("" :: "v"(qf[kk]));

    // ---- fragment addresses inside a buffer
    const int xr = HD == 64 ? (l31 >> 1) & 7 : l31 & 15;
    uint32_t ka[NKK];                             // K: row l31 (+32 per key block), chunk (2 kk + hi) ^ xr
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) ka[kk] = (uint32_t)(l31 * RB + (((kk * 2 + hi) ^ xr) << 4));
    // V (transpose load): lane = (group g = lane >> 4, s = lane & 15) points at 8 bytes of key row
    // 4 * (g >> 1) + (s >> 2) (+ 16 s' + {0, 8}), channels (g & 1) * 16 + 4 * (s & 3) .. + 3 (+ 32 it)
    const int g = lane >> 4, s16 = lane & 15;
    const int vrow = 4 * (g >> 1) + (s16 >> 2);                          // 0 .. 7
    const int vcol = ((g & 1) * 32 + (s16 & 3) * 8);                     // byte offset inside the 64-B half
    uint32_t va[NIT];                                                    // it = 32-channel block (row swizzle folded in)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int chunk = (it * 4 + (vcol >> 4)) ^ ((HD == 64 ? (vrow >> 1) & 1 : vrow & 3) << 2);
        va[it] = (uint32_t)(kTileBytes + vrow * RB + (chunk << 4) + (vcol & 15));
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    const float c = scale * 1.4426950408889634f;                         // log2(e) * softmax scale
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 ot[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int e = 0; e < 16; ++e) ot[it][e] = 0.f;

    const bool active = qb * 32 < tokens && wave < qcount;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // Deferred rescale (log2 domain): the running max is only raised, and O / l only rescaled, when some
    // row of the wave grew by more than kDefer; otherwise P is taken against the old max and is bounded
    // by 2^kDefer (exact in f32 accumulation, well inside the f16 / bf16 range as an MFMA operand).
    constexpr float kDefer = 8.0f;

    auto tile = [&](const char* buf, uint32_t bufa, int j, bool full) {
        // ---------------- S^T = K Q^T  (32 queries x 64 keys); first MFMA of a block takes C = 0
        f32x16 st[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (kb == 1 && !full) break;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const Frag kf = *(const Frag*)(buf + kb * 32 * RB + ka[kk]);
                st[kb] = FMma<T>::run(kf, qf[kk], kk == 0 ? zero16 : st[kb]);
            }
        }
        if ((j + 1) * kKV > tokens) {                                     // mask keys past the end
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (kb == 1 && !full) break;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = j * kKV + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= tokens) st[kb][r] = -INFINITY;
                }
            }
        }
        // ---------------- online softmax
        float mx = st[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[0][r]);
        if (full) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[1][r]);
        }
        mx = half_swap_max(mx);
        if (__any((mx - m_run) * c > kDefer)) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
            m_run = m_new;
            l_run *= alpha;
            if (j > 0) {
#pragma unroll
                for (int it = 0; it < NIT; ++it)
#pragma unroll
                    for (int e = 0; e < 16; ++e) ot[it][e] *= alpha;
            }
        }
        // exponent arguments and row sums two scores per instruction (v_pk_fma_f32 / v_pk_add_f32): the kernel is
        // bound by the number of instructions its waves issue, not by any one pipe
        const f32x2_t c2 = {c, c}, nmb2 = {-m_run * c, -m_run * c};
        f32x2_t ps2 = {0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (kb == 1 && !full) break;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2_t a = __builtin_elementwise_fma(f32x2_t{st[kb][r], st[kb][r + 1]}, c2, nmb2);
                const f32x2_t p = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
                ps2 += p;
                st[kb][r] = p[0];
                st[kb][r + 1] = p[1];
            }
        }
        l_run += ps2[0] + ps2[1];

        // ---------------- O^T += V^T P^T
        const uint32_t v0 = bufa + va[0], v1 = bufa + va[1];
        auto pv_step = [&](auto SP) {                                     // one 16-key step
            constexpr int sp = decltype(SP)::value;
            Frag pf;
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[e] = (T)st[sp >> 1][(sp & 1) * 8 + e];
            u32x2 v0a = tr_read<sp * 16 * RB>(v0), v0b = tr_read<sp * 16 * RB + 8 * RB>(v0);
            u32x2 v1a = tr_read<sp * 16 * RB>(v1), v1b = tr_read<sp * 16 * RB + 8 * RB>(v1);
            if constexpr (HD == 64) {
                // the loads' destinations count as written only from here on (hipcc does not track asm loads)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0a), "+v"(v0b), "+v"(v1a), "+v"(v1b) :: "memory");
                const u32x4 f0 = {v0a[0], v0a[1], v0b[0], v0b[1]}, f1 = {v1a[0], v1a[1], v1b[0], v1b[1]};
                ot[0] = FMma<T>::run(__builtin_bit_cast(Frag, f0), pf, ot[0]);
                ot[1] = FMma<T>::run(__builtin_bit_cast(Frag, f1), pf, ot[1]);
            } else {
                const uint32_t v2 = bufa + va[NIT - 2], v3 = bufa + va[NIT - 1];
                u32x2 v2a = tr_read<sp * 16 * RB>(v2), v2b = tr_read<sp * 16 * RB + 8 * RB>(v2);
                u32x2 v3a = tr_read<sp * 16 * RB>(v3), v3b = tr_read<sp * 16 * RB + 8 * RB>(v3);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0a), "+v"(v0b), "+v"(v1a), "+v"(v1b), "+v"(v2a), "+v"(v2b), "+v"(v3a), "+v"(v3b) :: "memory");
                const u32x4 f0 = {v0a[0], v0a[1], v0b[0], v0b[1]}, f1 = {v1a[0], v1a[1], v1b[0], v1b[1]};
                const u32x4 f2 = {v2a[0], v2a[1], v2b[0], v2b[1]}, f3 = {v3a[0], v3a[1], v3b[0], v3b[1]};
                ot[0] = FMma<T>::run(__builtin_bit_cast(Frag, f0), pf, ot[0]);
                ot[1] = FMma<T>::run(__builtin_bit_cast(Frag, f1), pf, ot[1]);
                ot[NIT - 2] = FMma<T>::run(__builtin_bit_cast(Frag, f2), pf, ot[NIT - 2]);
                ot[NIT - 1] = FMma<T>::run(__builtin_bit_cast(Frag, f3), pf, ot[NIT - 1]);
            }
        };
        pv_step(IntC<0>{});
        pv_step(IntC<1>{});
        if (full) {
            pv_step(IntC<2>{});
            pv_step(IntC<3>{});
        }
    };

    for (int j = 0; j < nkv; ++j) {
        // tile j has landed when at most the 2 loads of each younger staged tile (j + 1, j + 2) are still in flight
        const int ahead = nkv - 1 - j < kNB - 2 ? nkv - 1 - j : kNB - 2;
        if (ahead == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ONE barrier per tile: tile j + 3 goes into the buffer of tile j - 1, which every wave has left by now
        if (j + kNB - 1 < nkv) stage(j + kNB - 1);
        const char* buf = smem + (j % kNB) * 2 * kTileBytes;
        const uint32_t bufa = lds_base + (j % kNB) * 2 * kTileBytes;

        // A wave without queries (8th wave at T = 197) only stages and keeps the barriers; a tile whose
        // second 32-key block lies wholly past the end (keys 224..255 at T = 197) runs as a half tile.
        if (active) tile(buf, bufa, j, j * kKV + 32 < tokens);
    }

    const float inv = 1.0f / half_swap_sum(l_run);
    {
        if (!active) return;
        // buffers nkv % kNB and (nkv + 1) % kNB held tiles nkv - 4 and nkv - 3: free (every wave has left iteration
        // nkv - 2, nothing is staged any more).  The eight waves' 32 x RB-byte blocks fill exactly these two buffers,
        // WRAPPING around the ring's end (nkv % 4 == 3: before round 4 waves 4-7 then wrote past the allocation, where LDS
        // drops the writes -- 129..192 and 385..448 tokens gave zero rows for their queries; no shape in use hit it).
        // row = query, 16-byte chunk c at (c ^ (row & (CPR - 1))) * 16
        constexpr int kStg = 32 * RB;
        char* stg = smem + ((nkv % kNB) * 2 * kTileBytes + wave * kStg) % (kNB * 2 * kTileBytes);
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const T a = (T)(ot[it][g4 * 4 + 0] * inv), b = (T)(ot[it][g4 * 4 + 1] * inv);
                const T cc = (T)(ot[it][g4 * 4 + 2] * inv), d = (T)(ot[it][g4 * 4 + 3] * inv);
                u32x2 o;
                o[0] = (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
                o[1] = (uint32_t)__builtin_bit_cast(uint16_t, cc) | ((uint32_t)__builtin_bit_cast(uint16_t, d) << 16);
                *(u32x2*)(stg + l31 * RB + (((it * 4 + g4) ^ (l31 & (CPR - 1))) << 4) + hi * 8) = o;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        constexpr int kRowsPer = 64 / CPR;            // rows one wave-wide 16-byte read covers (8 or 4)
#pragma unroll
        for (int i = 0; i < 32 / kRowsPer; ++i) {
            const int row = lane / CPR + kRowsPer * i, ch = lane % CPR;
            const u32x4 v = *(const u32x4*)(stg + row * RB + ((ch ^ (row & (CPR - 1))) << 4));
            const int q = qb * 32 + row;
            if (q < tokens) *(u32x4*)(out + ((size_t)img * tokens + q) * dim + head * kHD + ch * 8) = v;
        }
    }
}

template <typename T, int HD>
__global__ __launch_bounds__(kNW * 64, HD == 64 ? 4 : 2)
void attention_flash_kernel(const T* __restrict__ qkv, T* __restrict__ out, int tokens, int heads, int parts, int units,
                            float scale) {
    __shared__ __attribute__((aligned(16))) char smem[kNB * 2 * kKV * HD * 2];     // [buf][K | V]
    flash_unit<T, HD>(smem, qkv, out, tokens, heads, parts, units, scale);
}



// ---------------------------------------------------------------------------------------------------------------------
// Pipelined form (head width 64): the same workgroup shape, ring and operand layouts as the tiled kernel above, but
//   * the row maximum is fixed ONCE per query, from the scores against key block 0 and against the keys at the queries' own
//     positions (class / register tokens and a token's neighbourhood carry the large scores of a ViT), so the key loop has no
//     running maximum, no rescale and no branch: per 32-key block a lane issues 8 packed fma, 16 v_exp, 16 adds and 8
//     conversions beside 8 MFMAs;
//   * S^T of block b + 1 is accumulated while block b's exponentials run (a block's MFMAs and its VALU work are independent
//     instruction streams inside ONE basic block, so the scheduler can interleave them);
//   * P is taken against that fixed maximum and may exceed 1: f32 row sums, f16 / bf16 P up to the type's range.  A score that
//     exceeds the fixed maximum by more than the type's range (f16: 16 binades = 11 nats) makes P infinite, which reaches the
//     output accumulators as inf / NaN: the workgroup then REDOES its queries with the tiled kernel's running-maximum body
//     (flash_unit).  Results never depend on which body ran beyond the rounding of P (both take P against a maximum that is a
//     power-of-two-free f32 offset: the two bodies differ in the last bits, the kernel is still bit-repeatable).
template <typename T>
__global__ __launch_bounds__(kNW * 64, 4)
void attention_pipe_kernel(const T* __restrict__ qkv, T* __restrict__ out, int tokens, int heads, int parts, int units, float scale) {
    constexpr int RB = 128, kTileBytes = kKV * RB, NKK = 4, CPR = 8;
    __shared__ __attribute__((aligned(16))) char smem[kNB * 2 * kTileBytes];     // [buf][K | V]
    using Frag = typename FMma<T>::Frag;
    typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
    using lds_s16x4 = __attribute__((address_space(3))) s16x4;
    using lds_char = __attribute__((address_space(3))) char;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int slot = blockIdx.x >> 3;
    const int unit = (slot / parts) * 8 + (blockIdx.x & 7), part = slot % parts;
    if (unit >= units) return;
    const int img = unit / heads, head = unit - img * heads;
    const int dim = heads * 64;
    const uint32_t ldb = (uint32_t)(3 * dim) * 2;
    const char* base = (const char*)(qkv + (size_t)img * tokens * 3 * dim + head * 64);
    const char* kbase = base + (size_t)dim * 2;
    const char* vbase = base + (size_t)dim * 4;
    const int nkv = (tokens + kKV - 1) / kKV;
    const int nblk = (tokens + 31) / 32;

    // ---- staging plan (as in the tiled kernel, one piece per thread, operand and tile)
    const int srow = tid / CPR, spos = tid % CPR;
    const uint32_t kchunk = (uint32_t)((spos ^ ((srow >> 1) & 7)) << 4);
    const uint32_t vchunk = (uint32_t)((spos ^ (((srow >> 1) & 1) << 2)) << 4);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_char*)smem;
    const uint32_t lds0 = lds_base + wave * 1024;
    auto stage = [&](int j) {
        int row = j * kKV + srow;
        row = row < tokens ? row : tokens - 1;
        const uint32_t roff = (uint32_t)row * ldb;
        const uint32_t dst = lds0 + (j % kNB) * 2 * kTileBytes;
        if constexpr (!(AP_PIPE_ABL & 64)) {
            dma16(kbase, roff + kchunk, dst);
            dma16(vbase, roff + vchunk, dst + kTileBytes);
        }
    };
#pragma unroll
    for (int j = 0; j < kNB - 1; ++j)
        if (j < nkv) stage(j);

    // ---- this wave's queries, and the keys at the same positions (for the fixed maximum)
    const int nqb_all = nblk;
    const int qcount = nqb_all / parts + (part < nqb_all % parts);
    const int qb = part * (nqb_all / parts) + (part < nqb_all % parts ? part : nqb_all % parts) + wave;
    int qrow = qb * 32 + l31;
    if (qrow >= tokens) qrow = tokens - 1;
    Frag qf[NKK], kd[NKK];
    {
        const T* qp = (const T*)(base + (size_t)qrow * ldb);
        const T* kp = (const T*)(kbase + (size_t)qrow * ldb);
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) qf[kk] = *(const Frag*)(qp + kk * 16 + hi * 8);
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) kd[kk] = *(const Frag*)(kp + kk * 16 + hi * 8);
    }
    // used here so that hipcc's vmcnt(0) for these loads (which also drains the staged tiles: the LDS-DMA stream is invisible
    // to its counter model) sits in front of the loop and not inside it
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) asm volatile("" :: "v"(qf[kk]), "v"(kd[kk]));

    const int xr = (l31 >> 1) & 7;
    uint32_t ka[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) ka[kk] = (uint32_t)(l31 * RB + (((kk * 2 + hi) ^ xr) << 4));
    const int g = lane >> 4, s16 = lane & 15;
    const int vrow = 4 * (g >> 1) + (s16 >> 2);
    const int vcol = ((g & 1) * 32 + (s16 & 3) * 8);
    uint32_t va[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int chunk = (it * 4 + (vcol >> 4)) ^ (((vrow >> 1) & 1) << 2);
        va[it] = (uint32_t)(kTileBytes + vrow * RB + (chunk << 4) + (vcol & 15));
    }

    const bool active = qb * 32 < tokens && wave < qcount;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float c = scale * 1.4426950408889634f;
    f32x16 ot[2] = {zero16, zero16};
    f32x16 st = zero16;
    float l_run = 0.f, nmb = 0.f;

    // every staged tile has landed for THIS thread (the vmcnt(0) above); the barrier makes that true for all of them
    __builtin_amdgcn_s_barrier();
    if (kNB - 1 < nkv) stage(kNB - 1);
    if (active) {
        f32x16 sd = zero16;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) sd = FMma<T>::run(kd[kk], qf[kk], sd);
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) st = FMma<T>::run(*(const Frag*)(smem + ka[kk]), qf[kk], kk == 0 ? zero16 : st);
        float mx = fmaxf(sd[0], st[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(sd[r], st[r]));
        nmb = -half_swap_max(mx) * c;          // rows past the end duplicate valid ones (clamped loads): harmless under a maximum
    }

    typedef short s16x8 __attribute__((__vector_size__(8 * sizeof(short))));
    const f32x2_t c2 = {c, c};
    // the pieces of one 32-key block ------------------------------------------------------------------------------------
    // S^T of a block: 4 MFMAs over the head width, K rows at `krows`
    auto qk = [&](const char* krows, f32x16& s) {
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            if constexpr (AP_PIPE_ABL & 2) { if (kk == 0) s = zero16; asm volatile("" : "+v"(s)); continue; }
            Frag kf;
            if constexpr (AP_PIPE_ABL & 8) kf = qf[(kk + 1) & 3]; else kf = *(const Frag*)(krows + ka[kk]);
            s = FMma<T>::run(kf, qf[kk], kk == 0 ? zero16 : s);
        }
    };
    // exponentials of 16 of the block's keys (step sp), their sum, P in the operand type
    auto softmax_step = [&](auto TAIL, const f32x16& s, int sp, int b, f32x2_t& ps2, Frag& pf) {
        const f32x2_t nmb2 = {nmb, nmb};
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const int r = sp * 8 + e;
            const f32x2_t a = __builtin_elementwise_fma(f32x2_t{s[r], s[r + 1]}, c2, nmb2);
            f32x2_t p;
            if constexpr (AP_PIPE_ABL & 1) p = a; else p = f32x2_t{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
            if constexpr (decltype(TAIL)::value) {
                const int key = b * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (key >= tokens) p[0] = 0.f;
                if (key + 1 >= tokens) p[1] = 0.f;
            }
            ps2 += p;
            pf[e] = (T)p[0];
            pf[e + 1] = (T)p[1];
        }
    };
    // O^T += V^T P^T for 16 keys: V^T fragments = 8 keys of one channel per lane (two transpose loads per 32-channel block)
    auto pv_step = [&](lds_char* vblk, int sp, const Frag& pf) {
        if constexpr (AP_PIPE_ABL & 4) { asm volatile("" :: "v"(pf)); return; }
        if constexpr (AP_PIPE_ABL & 16) {
            ot[0] = FMma<T>::run(qf[sp], pf, ot[0]);
            ot[1] = FMma<T>::run(qf[sp + 2], pf, ot[1]);
            return;
        }
        const s16x4 v0a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vblk + va[0] + sp * 16 * RB));
        const s16x4 v0b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vblk + va[0] + sp * 16 * RB + 8 * RB));
        const s16x4 v1a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vblk + va[1] + sp * 16 * RB));
        const s16x4 v1b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vblk + va[1] + sp * 16 * RB + 8 * RB));
        const s16x8 f0 = __builtin_shufflevector(v0a, v0b, 0, 1, 2, 3, 4, 5, 6, 7);
        const s16x8 f1 = __builtin_shufflevector(v1a, v1b, 0, 1, 2, 3, 4, 5, 6, 7);
        ot[0] = FMma<T>::run(__builtin_bit_cast(Frag, f0), pf, ot[0]);
        ot[1] = FMma<T>::run(__builtin_bit_cast(Frag, f1), pf, ot[1]);
    };
    // a block of the ragged end: no read-ahead unless LOOK, keys past the end masked when TAIL
    auto block = [&](auto LOOK, auto TAIL, const char* knext, lds_char* vblk, int b) {
        f32x16 sn = zero16;
        if constexpr (decltype(LOOK)::value) qk(knext, sn);
        f32x2_t ps2 = {0.f, 0.f};
        Frag pf0, pf1;
        softmax_step(TAIL, st, 0, b, ps2, pf0);
        pv_step(vblk, 0, pf0);
        softmax_step(TAIL, st, 1, b, ps2, pf1);
        pv_step(vblk, 1, pf1);
        l_run += ps2[0] + ps2[1];
        if constexpr (decltype(LOOK)::value) st = sn;
    };
    // the issue order asked of the scheduler for ONE block of an interior tile (everything inside is one basic block):
    // the 4 read-ahead MFMAs each followed by 5 of the first step's 20 VALU / transcendental instructions, the first step's two
    // P V MFMAs each followed by 5 of the second step's, the rest of the second step, its two MFMAs
    auto block_order = [&]() {
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                 // K fragments of the next block
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                 // V fragments, first 16 keys
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002 | 0x400, 5, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                 // V fragments, second 16 keys
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002 | 0x400, 5, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x002 | 0x400, 10, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    };

    const bool ragged = (tokens & 31) != 0;
    // tile j opens with: tile j + 1 (whose first K rows tile j's second block reads ahead) has landed -- at most tile j + 2's two
    // loads still in flight -- for every thread (barrier), and tile j + 3 is staged into the buffer of tile j - 1, which every
    // wave has left
    auto open_tile = [&](int j) {
        if (j > 0) {
            if (j + 2 < nkv) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (!(AP_PIPE_ABL & 128)) __builtin_amdgcn_s_barrier();
            if (j + kNB - 1 < nkv) stage(j + kNB - 1);
        }
    };
    // interior tiles: two blocks, the block after them exists.  S^T alternates between st and sn, no branch inside.
    const int n_int = (nblk - 1) / 2;
    for (int j = 0; j < n_int; ++j) {
        open_tile(j);
        if (active) {
            const char* buf = smem + (j % kNB) * 2 * kTileBytes;
            const char* bufn = smem + ((j + 1) % kNB) * 2 * kTileBytes;
            lds_char* vb = (lds_char*)smem + (j % kNB) * 2 * kTileBytes;
            f32x16 sn = zero16;
            f32x2_t ps2 = {0.f, 0.f};
            Frag pf0, pf1;
            qk(buf + 32 * RB, sn);
            softmax_step(IntC<0>{}, st, 0, 0, ps2, pf0);
            pv_step(vb, 0, pf0);
            softmax_step(IntC<0>{}, st, 1, 0, ps2, pf1);
            pv_step(vb, 1, pf1);
            block_order();
            qk(bufn, st);
            softmax_step(IntC<0>{}, sn, 0, 0, ps2, pf0);
            pv_step(vb + 32 * RB, 0, pf0);
            softmax_step(IntC<0>{}, sn, 1, 0, ps2, pf1);
            pv_step(vb + 32 * RB, 1, pf1);
            block_order();
            l_run += ps2[0] + ps2[1];
        }
    }
    // the last tile: one or two blocks, nothing to read ahead past it, keys past the end masked
    {
        const int j = n_int, b0 = 2 * j;
        open_tile(j);
        if (active) {
            const char* buf = smem + (j % kNB) * 2 * kTileBytes;
            lds_char* vb = (lds_char*)smem + (j % kNB) * 2 * kTileBytes;
            if (b0 + 1 < nblk) {
                block(IntC<1>{}, IntC<0>{}, buf + 32 * RB, vb, b0);
                if (ragged) block(IntC<0>{}, IntC<1>{}, buf, vb + 32 * RB, b0 + 1);
                else block(IntC<0>{}, IntC<0>{}, buf, vb + 32 * RB, b0 + 1);
            } else {
                if (ragged) block(IntC<0>{}, IntC<1>{}, buf, vb, b0);
                else block(IntC<0>{}, IntC<0>{}, buf, vb, b0);
            }
        }
    }

    // ---- did P stay inside the operand type's range?  x * 0 is NaN for an infinite or NaN x
    const float l_tot = half_swap_sum(l_run);
    float chk = l_tot * 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int e = 0; e < 16; ++e) chk = __builtin_fmaf(ot[it][e], 0.f, chk);
    const int bad = active && !(chk == 0.f);
    if ((AP_PIPE_ABL & 32) ? 0 : __syncthreads_or(bad)) {             // every DMA has landed (vmcnt(0) in the last iteration), every wave has left the ring
        flash_unit<T, 64>(smem, qkv, out, tokens, heads, parts, units, scale);
        return;
    }
    if (!active) return;
    const float inv = 1.0f / l_tot;
    {
        constexpr int kStg = 32 * RB;
        char* stg = smem + wave * kStg;
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const T a = (T)(ot[it][g4 * 4 + 0] * inv), b = (T)(ot[it][g4 * 4 + 1] * inv);
                const T cc = (T)(ot[it][g4 * 4 + 2] * inv), d = (T)(ot[it][g4 * 4 + 3] * inv);
                u32x2 o;
                o[0] = (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
                o[1] = (uint32_t)__builtin_bit_cast(uint16_t, cc) | ((uint32_t)__builtin_bit_cast(uint16_t, d) << 16);
                *(u32x2*)(stg + l31 * RB + (((it * 4 + g4) ^ (l31 & (CPR - 1))) << 4) + hi * 8) = o;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = lane / CPR + 8 * i, ch = lane % CPR;
            const u32x4 v = *(const u32x4*)(stg + row * RB + ((ch ^ (row & (CPR - 1))) << 4));
            const int q = qb * 32 + row;
            if (q < tokens) *(u32x4*)(out + ((size_t)img * tokens + q) * dim + head * 64 + ch * 8) = v;
        }
    }
}

}  // namespace

int launch_attention_flash(int dtype, const void* qkv, void* out, int n, int tokens, int heads, int head_dim, float scale,
                           hipStream_t stream) {
    AP_REQUIRE(dtype == AP_F16 || dtype == AP_BF16, "attention_flash: f16 / bf16 only");
    AP_REQUIRE(head_dim == 64 || head_dim == 128, "attention_flash: head_dim %d (64 / 128)", head_dim);
    AP_REQUIRE((size_t)tokens * 3 * heads * head_dim * 2 < 0xffffffffull, "attention_flash: sequence too long");
    if (n <= 0) return AP_OK;
    const int nqb = (tokens + 31) / 32;
    const int parts = (nqb + kNW - 1) / kNW, units = n * heads;
    dim3 grid((unsigned)((units + 7) / 8 * 8 * parts)), block(kNW * 64);
#define AP_FLASH(T, HD) attention_flash_kernel<T, HD><<<grid, block, 0, stream>>>((const T*)qkv, (T*)out, tokens, heads, parts, units, scale)
    // head width 64: the pipelined kernel (fixed row maximum, S^T of the next block beside the softmax of this one);
    // AP_ATTN_IMPL=flash keeps the tiled running-maximum kernel for A/B timing
    static const bool tiled = [] { const char* e = getenv("AP_ATTN_IMPL"); return e && e[0] == 'f'; }();
    if (head_dim == 64 && !tiled) {
        if (dtype == AP_F16) attention_pipe_kernel<f16><<<grid, block, 0, stream>>>((const f16*)qkv, (f16*)out, tokens, heads, parts, units, scale);
        else attention_pipe_kernel<bf16><<<grid, block, 0, stream>>>((const bf16*)qkv, (bf16*)out, tokens, heads, parts, units, scale);
    } else if (head_dim == 64) { if (dtype == AP_F16) AP_FLASH(f16, 64); else AP_FLASH(bf16, 64); }
    else { if (dtype == AP_F16) AP_FLASH(f16, 128); else AP_FLASH(bf16, 128); }
#undef AP_FLASH
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

}  // namespace ap
