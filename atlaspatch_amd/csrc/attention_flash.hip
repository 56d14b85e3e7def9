// Tiled ("flash") multi-head attention for f16 / bf16, heads stored 64, 96 or 128 wide, any sequence length:
//     out = softmax(q k^T * scale) v      per (image, head), f32 softmax, online rescaling.
// (The text below describes the 64-wide instantiation; HD = 128 -- vit_h_14's 80-wide heads, zero-padded -- has 256-byte
// rows: two DMA pieces per thread, operand and tile, K chunks XOR-swizzled by row & 15, V 64-byte windows by row & 3, four
// 32-channel output blocks, 128 KiB of LDS.  HD = 96 -- the 80-wide heads of ViT-H/14 and Virchow, zero-padded: 192-byte rows,
// 12 chunks per row: the 768 chunks of a tile are staged as one piece by every thread and a second by waves 0-3; K chunks
// XOR-swizzled by (row >> 2) & 3 inside their 64-byte window (consecutive rows already step 192 bytes = three quarters of the
// 256-byte bank line, so 16 rows x one chunk land in 16 different 16-byte slots), V rows unswizzled (four rows of a transpose
// load fall in four different 64-byte quarters by the row stride alone), three 32-channel output blocks, 96 KiB of LDS and 150 registers: one workgroup per CU like HD = 128 (a ring of three
// buffers would admit two, but not inside 128 registers: 18 spilled).)
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
#include "ap_common.h"

namespace ap {
namespace {

constexpr int kKV = 64;                 // keys per tile
constexpr int kNW = 8;                  // waves per workgroup (round 6: five-, six- and seven-wave workgroups, with a three-buffer
                                        //   ring so that three are resident, measured slower at 197 / 257 / 261 / 265 / 300 / 320 / 785
                                        //   tokens: profiles/r06c_attention.txt -- the eighth wave's share of the staging is worth more
                                        //   than its slot)
constexpr int kNB = 4;                  // K/V ring buffers (prefetch distance kNB - 1); >= 3 (the epilogue stages through two idle ones)

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

// gfx950 transpose load (ds_read_b64_tr_b16) through the builtin: hipcc folds the constant offset into the instruction, counts
// it in lgkmcnt itself and is free to issue it early (round 6; the inline-asm form of rounds 2-5 had to be waited for on the spot)
template <int OFF> __device__ __forceinline__ u32x2 tr_read(const char* lds_ptr) {
    typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));
    const s4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(lds_ptr + OFF));
    return __builtin_bit_cast(u32x2, v);
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

// row sums of P from the ROUNDED operand values (what the P V product multiplies): sum of 8 packed values into f32
template <typename T> __device__ __forceinline__ float psum8(float acc, typename FMma<T>::Frag pf);
template <> __device__ __forceinline__ float psum8<f16>(float acc, f16x8 pf) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const u32x4 w = __builtin_bit_cast(u32x4, pf);
    const h2 one = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t wk = w[k];
        acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, wk), one, acc, false);      // v_dot2_f32_f16: exact products, f32 sum
    }
    return acc;
}
template <> __device__ __forceinline__ float psum8<bf16>(float acc, bf16x8 pf) {
    const u32x4 w = __builtin_bit_cast(u32x4, pf);
#pragma unroll
    for (int k = 0; k < 4; ++k) acc += __builtin_bit_cast(float, w[k] << 16) + __builtin_bit_cast(float, w[k] & 0xffff0000u);
    return acc;
}

template <typename T, int HD>
__global__ __launch_bounds__(kNW * 64, HD == 64 ? 4 : 2)
void attention_flash_kernel(const T* __restrict__ qkv, T* __restrict__ out, int tokens, int heads, int parts, int units,
                            float scale) {
    constexpr int kHD = HD;
    constexpr int RB = HD * 2;                       // bytes of one K / V row of a head
    constexpr int kTileBytes = kKV * RB;             // one K or V tile (64 rows)
    constexpr int NKK = HD / 16;                     // k-steps of S^T = K Q^T
    constexpr int NIT = HD / 32;                     // 32-channel output blocks
    constexpr int NPC = (kTileBytes + kNW * 64 * 16 - 1) / (kNW * 64 * 16);   // DMA pieces per thread, operand and tile (1 or 2; HD = 96:
                                                     //   the second piece exists for the first half of the threads only)
    constexpr int CPR = RB / 16;                     // 16-byte chunks per row (8, 12 or 16)
    constexpr int kChunks = kTileBytes / 16;         // chunks of one operand tile
    __shared__ __attribute__((aligned(16))) char smem[kNB * 2 * kTileBytes];     // [buf][K | V]
    using Frag = typename FMma<T>::Frag;

    const int tid = threadIdx.x, lane = tid & 63;
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
        } else if constexpr (HD == 96) {
            kchunk[pc] = (uint32_t)((spos ^ ((row >> 2) & 3)) << 4);
            vchunk[pc] = (uint32_t)(spos << 4);
        } else {
            kchunk[pc] = (uint32_t)((spos ^ (row & 15)) << 4);
            vchunk[pc] = (uint32_t)((spos ^ ((row & 3) << 2)) << 4);
        }
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 1024;
    auto stage = [&](int j) {
#pragma unroll
        for (int pc = 0; pc < NPC; ++pc) {
            if (pc * (kNW * 64) + wave * 64 >= kChunks) break;            // HD = 96: no second piece for waves 4-7 (wave-uniform)
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
    const int xr = HD == 64 ? (l31 >> 1) & 7 : (HD == 96 ? (l31 >> 2) & 3 : l31 & 15);
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
        const int chunk = (it * 4 + (vcol >> 4)) ^ ((HD == 64 ? (vrow >> 1) & 1 : (HD == 96 ? 0 : vrow & 3)) << 2);
        va[it] = (uint32_t)(kTileBytes + vrow * RB + (chunk << 4) + (vcol & 15));
    }

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

    // One key tile.  NS = 16-key steps that hold valid keys: 4 = the whole tile, 2 = its first 32-key block only (the second
    // lies wholly past the end: keys 224..255 at 197 tokens), 1 = its first 16 keys only (the last tile of 197 / 257 / 261 / 265
    // / 785 tokens holds 5 / 1 / 5 / 9 / 17 keys): S^T is still one 32-key MFMA block, but max / exp / sum / cvt run on the lane's
    // first 8 scores and one P V step is issued instead of two.
    auto tile = [&](const char* buf, int j, auto NSc, auto LASTc) {
        constexpr int NS = decltype(NSc)::value;
        constexpr bool kLast = decltype(LASTc)::value != 0;              // only the last tile can hold keys past the end
        constexpr int NKB = NS == 4 ? 2 : 1;                              // 32-key blocks of S^T
        constexpr int NR = NS == 1 ? 8 : 16;                              // scores per lane and block that can be valid
        // ---------------- S^T = K Q^T  (32 queries x 64 keys); first MFMA of a block takes C = 0
        f32x16 st[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const Frag kf = *(const Frag*)(buf + kb * 32 * RB + ka[kk]);
                st[kb] = FMma<T>::run(kf, qf[kk], kk == 0 ? zero16 : st[kb]);
            }
        }
        if (kLast && (j + 1) * kKV > tokens) {                            // mask keys past the end
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int key = j * kKV + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= tokens) st[kb][r] = -INFINITY;
                }
            }
        }
        // ---------------- online softmax (row maximum three values per instruction: v_max3_f32)
        float mx = fmaxf(st[0][0], st[0][1]);
#pragma unroll
        for (int r = 2; r < NR; r += 2) mx = fmaxf(fmaxf(mx, st[0][r]), st[0][r + 1]);
        if constexpr (NKB == 2) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, st[1][r]), st[1][r + 1]);
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
        // exponent arguments two scores per instruction (v_pk_fma_f32)
        const f32x2_t c2 = {c, c}, nmb2 = {-m_run * c, -m_run * c};
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int r = 0; r < NR; r += 2) {
                const f32x2_t a = __builtin_elementwise_fma(f32x2_t{st[kb][r], st[kb][r + 1]}, c2, nmb2);
                st[kb][r] = __builtin_amdgcn_exp2f(a[0]);
                st[kb][r + 1] = __builtin_amdgcn_exp2f(a[1]);
            }
        }

        // ---------------- O^T += V^T P^T; the row sums are taken from the ROUNDED P (psum8: the very values the product
        // multiplies -- out = sum p~ v / sum p~ is a convex combination of the v rows whatever the rounding of p~ did)
        auto pv_step = [&](auto SP) {                                     // one 16-key step
            constexpr int sp = decltype(SP)::value;
            Frag pf;
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[e] = (T)st[sp >> 1][(sp & 1) * 8 + e];
            l_run = psum8<T>(l_run, pf);
            u32x2 va_[NIT], vb_[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                va_[it] = tr_read<sp * 16 * RB>(buf + va[it]);
                vb_[it] = tr_read<sp * 16 * RB + 8 * RB>(buf + va[it]);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const u32x4 f = {va_[it][0], va_[it][1], vb_[it][0], vb_[it][1]};
                ot[it] = FMma<T>::run(__builtin_bit_cast(Frag, f), pf, ot[it]);
            }
        };
        pv_step(IntC<0>{});
        if constexpr (NS >= 2) pv_step(IntC<1>{});
        if constexpr (NS == 4) {
            pv_step(IntC<2>{});
            pv_step(IntC<3>{});
        }
    };

    // every tile but the last is whole; the last one runs the variant its valid keys need (peeled: one tile body inside the loop)
    auto step = [&](int j, auto NSc, auto LASTc) {
        // tile j has landed when at most the 2 loads of each younger staged tile (j + 1, j + 2) are still in flight
        const int ahead = nkv - 1 - j < kNB - 2 ? nkv - 1 - j : kNB - 2;
        if (ahead == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ONE barrier per tile: tile j + 3 goes into the buffer of tile j - 1, which every wave has left by now
        if (j + kNB - 1 < nkv) stage(j + kNB - 1);
        const char* buf = smem + (j % kNB) * 2 * kTileBytes;
        // A wave without queries (8th wave at T = 197) only stages and keeps the barriers.
        if (active) tile(buf, j, NSc, LASTc);
    };
    for (int j = 0; j < nkv - 1; ++j) step(j, IntC<4>{}, IntC<0>{});
    {
        const int left = tokens - (nkv - 1) * kKV;                        // valid keys of the last tile (1 .. 64)
        if (left > 32) step(nkv - 1, IntC<4>{}, IntC<1>{});
        else if (left > 16) step(nkv - 1, IntC<2>{}, IntC<1>{});
        else step(nkv - 1, IntC<1>{}, IntC<1>{});
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
                *(u32x2*)(stg + l31 * RB + (((it * 4 + g4) ^ (l31 & (HD == 96 ? 3 : CPR - 1))) << 4) + hi * 8) = o;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        constexpr int kSwz = HD == 96 ? 3 : CPR - 1;  // chunk swizzle mask of the staging rows (inside the 12 chunks at HD = 96)
#pragma unroll
        for (int i = 0; i < 32 * CPR / 64; ++i) {     // 32 rows x CPR chunks, one 16-byte chunk per lane and step
            const int idx = i * 64 + lane;
            const int row = idx / CPR, ch = idx % CPR;
            const u32x4 v = *(const u32x4*)(stg + row * RB + ((ch ^ (row & kSwz)) << 4));
            const int q = qb * 32 + row;
            if (q < tokens) *(u32x4*)(out + ((size_t)img * tokens + q) * dim + head * kHD + ch * 8) = v;
        }
    }
}


}  // namespace

int launch_attention_flash(int dtype, const void* qkv, void* out, int n, int tokens, int heads, int head_dim, float scale,
                           hipStream_t stream) {
    AP_REQUIRE(dtype == AP_F16 || dtype == AP_BF16, "attention_flash: f16 / bf16 only");
    AP_REQUIRE(head_dim == 64 || head_dim == 96 || head_dim == 128, "attention_flash: head_dim %d (64 / 96 / 128)", head_dim);
    AP_REQUIRE((size_t)tokens * 3 * heads * head_dim * 2 < 0xffffffffull, "attention_flash: sequence too long");
    if (n <= 0) return AP_OK;
    const int nqb = (tokens + 31) / 32;
    const int parts = (nqb + kNW - 1) / kNW, units = n * heads;
    dim3 grid((unsigned)((units + 7) / 8 * 8 * parts)), block(kNW * 64);
#define AP_FLASH(T, HD) attention_flash_kernel<T, HD><<<grid, block, 0, stream>>>((const T*)qkv, (T*)out, tokens, heads, parts, units, scale)
    if (head_dim == 64) { if (dtype == AP_F16) AP_FLASH(f16, 64); else AP_FLASH(bf16, 64); }
    else if (head_dim == 96) { if (dtype == AP_F16) AP_FLASH(f16, 96); else AP_FLASH(bf16, 96); }
    else { if (dtype == AP_F16) AP_FLASH(f16, 128); else AP_FLASH(bf16, 128); }
#undef AP_FLASH
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

}  // namespace ap
