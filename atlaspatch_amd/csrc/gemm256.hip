// Persistent 256x256-tile MFMA GEMM for the ViT encoder's large linears (f16 / bf16 operands):
//     C[m][n] = sum_k Act[m][k] * W[n][k]   (+ fused epilogue),  both operands K-contiguous.
//
// One 512-thread workgroup per CU (grid = min(#CU, #tiles)) walks its list of 256(m) x 256(n)
// output tiles; the K loop of all its tiles is ONE continuous stream of 64-element K-tiles that
// never drains: while a tile's epilogue runs, the LDS-DMA for the next tile's first K-tiles is
// already in flight.
//
//  * waves: 8 = 2 (wr, along m) x 4 (wc, along n); a wave owns 128 m x 64 n = 4 x 2 MFMA tiles of
//    32x32 (v_mfma_f32_32x32x16_{f16,bf16}; 128 accumulator registers).  The MFMA "A" operand is the
//    weight fragment so a lane ends up with 4 consecutive n of one m.
//  * LDS (160 KiB): two K-tile buffers of 64 KiB + 8 x 4 KiB per-wave epilogue scratch.  A buffer
//    holds four 16-KiB staging UNITS of 128 rows x 128 B, cut along the waves' output QUADRANTS
//    rather than along the tile:  X0 / X1 = the first / second 64 m-rows of both wr groups,
//    Y0 / Y1 = the first / second 32 n-rows of all four wc groups.  A K-tile is consumed in four
//    phases  (X0,Y0) (X0,Y1) (X1,Y1) (X1,Y0)  of 8 MFMAs per wave, so every unit has ONE reading
//    phase (Y0's fragment stays in registers for phase 3).  That makes a unit free two phases
//    after it was read and lets a plain double buffer run ~4 phases (a whole K-tile) of prefetch:
//        phase 0: read X0,Y0   stage Y1 of K-tile v+1      phase 2: read X1   stage X0 of v+2
//        phase 1: read Y1      stage X1 of K-tile v+1      phase 3: -         stage Y0 of v+2
//    A phase is  s_waitcnt vmcnt(6); s_barrier; [8 MFMAs with the phase's unit staged between them: 2 x
//    global_load_lds_dwordx4 per lane -- and, since round 5, the NEXT phase's fragment reads].  vmcnt(6) = "everything staged four
//    phases ago has landed"; data is read no earlier than behind the barrier that follows the wait that retires it
//    (round 2-4's loop: one phase later) and a unit is restaged no earlier than one barrier after its reading phase (raw s_barrier,
//    never vmcnt(0) in the loop).  Loads retire in issue order among themselves, which is all the counted
//    wait relies on; vmcnt also counts the epilogue's global stores, and a store may retire before OR after a
//    load issued around it: older stores still in the queue only make vmcnt(6) stricter (safe), but nothing may
//    be concluded about a load from a count once YOUNGER stores are outstanding -- there the wait is a full
//    drain.  The epilogue opens with such a drain, so the first K-tile after it skips the counted wait and the
//    epilogue's stores drain under the next tile's MFMAs.
//  * the LDS-DMA is issued from inline asm (SGPR base + 32-bit lane offset, M0 = LDS address): the
//    compiler treats the builtin as a FLAT access and then turns every later LDS / VMEM wait into a
//    full drain; hidden from it, the fragment reads get counted lgkmcnt(N) waits, so the first
//    MFMAs of a phase start as soon as their own operands have arrived.
//  * the LDS-DMA writes lane-linear, so the bank swizzle (16-B chunk ^= (row >> 1) & 7, conflict
//    free for ds_read_b128's lane groups) is applied to the per-lane SOURCE address and again on
//    the fragment read.
//  * accumulators start from the bias; epilogue: (GELU | *LayerScale) in registers, transposed through the wave's private LDS
//    scratch (XOR-swizzled) so that global stores / the f32 residual read-modify-write are whole
//    128-byte rows, 16 B per lane.
//  * fused-LayerNorm epilogues (EPI_NORM_STORE / EPI_NORM_GELU / EPI_RESID_STATS / EPI_PATCH_STREAM, see ap_common.h): the A operand
//    is the raw 16-bit residual stream; NORM applies the row statistics and the rank-one mean correction per element
//    (accumulators start from zero there), RESID_STATS / PATCH_STREAM add the accumulator to the stream window (or to the
//    position-embedding row) after the transposition, 16 bytes per lane, and emit per-row partial sums for the next
//    statistics.  RESID_STATS / PATCH_STREAM fetch their operands (bias, the 128 x 64 stream window) with plain loads +
//    __builtin_amdgcn_s_waitcnt before the epilogue body; the NORM epilogues' operands arrive in the wave's idle scratch by
//    LDS-DMA during the tile's K loop and the drain sits in front of the tile's first store (round 5, see kLdsOps below).
//  * round 5: the fragment reads of a phase are issued between the MFMAs of the phase BEFORE it (ktile_p below); 16-bit
//    plain-store epilogues write their rows with the non-temporal hint (out_store16).
//  * XCD-aware tile order: block b runs on XCD b % 8; each XCD owns a contiguous range of tile ids
//    (n fastest), its 32 workgroups take consecutive ids, so concurrently running tiles share
//    activation panels and weight panels in that XCD's L2.
//
// Roofline: MFMA (2*M*N*K flop per launch).
#include "ap_common.h"
#include "gemm_mma.h"

// The file is compiled twice: the product build, and (-DAP_G256_ALT) a twin that ap_gemm reaches as impl 257, so a
// schedule change can be A/B-timed inside one process (tools/gemm_check.py, tools/gemm_twin_ab.py).  The twin normally
// also carries the diagnostics (-DAP_G256_DIAG: per-tile time stamps, start skew, ablation flags); an experiment is
// timed with `make ALT_FLAGS="-DAP_G256_ALT -DAP_EXP_..."`, i.e. the product code plus the experiment, without them.
// Round 3 timed three such experiments on the four ViT-B shapes (2048 images), all inside +-1 % of the product kernel
// and none kept: the tile's first K-tile writing the accumulators with C = 0 instead of a zeroed register block; the
// last K-tile's phase-2 / 3 stagings issued behind the drain that opens the epilogue; the two LDS-DMA loads of a phase
// issued two MFMA pairs apart.  A fourth replaced the epilogue's LDS transposition by v_permlane32_swap pairs and 16-byte
// row-per-lane stores (32 rows x 32 bytes per instruction): bit-identical, qkv 5 % slower, fc1 unchanged -- the whole-row
// stores are worth their LDS round trip.  s_setprio 1 for waves 4-7 before the main loop, and s_setprio 1 / 0 around every phase's
// MFMA cluster: both inside +-0.5 %.  The non-temporal hint on the activation panel's LDS-DMA (so that the six weight panels an XCD
// re-reads wave after wave stay in its L2): 1.3-2.6 % slower.
#ifdef AP_G256_ALT
#define AP_G256_FN(name) name##_alt
#else
#define AP_G256_FN(name) name
#endif
#ifdef AP_G256_DIAG
#define AP_G256_DIAG_ON true
#else
#define AP_G256_DIAG_ON false
#endif

namespace ap {
namespace {

constexpr bool kDiag = AP_G256_DIAG_ON;           // the twin's diagnostics (start skew, time stamps, ablation flags): `if constexpr`, no code in the product
constexpr int kBM = 256, kBN = 256;
constexpr int kRowBytes = 128;                    // bytes of K per tile row (64 f16 / bf16)
constexpr int kUnitBytes = 128 * kRowBytes;       // 16 KiB
constexpr int kBufBytes = 4 * kUnitBytes;         // X0 X1 Y0 Y1
constexpr int kScratchOff = 2 * kBufBytes;        // 128 KiB
constexpr int kLdsBytes = kScratchOff + 8 * 4096; // 160 KiB
enum { U_X0 = 0, U_X1 = 1, U_Y0 = 2, U_Y1 = 3 };

// Two LDS-DMA loads of one staging unit (16 B per lane each): LDS rows [0, 64) and [64, 128) of the
// unit slice owned by this wave.  base: wave-uniform global address, off0 / off1: per-lane byte
// offsets, lds_dst: wave-uniform LDS byte address.  M0 is saved and restored (compiler-reserved).
__device__ __forceinline__ void dma_unit(const char* base, uint32_t off0, uint32_t off1, uint32_t lds_dst) {
    uint32_t keep;
    const uint32_t lds_dst1 = lds_dst + 64 * kRowBytes;
    asm volatile(              // only s_mov / s_nop besides the loads: SCC stays untouched
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %5\n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %5\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(off0), "v"(off1), "s"(lds_dst), "s"(lds_dst1), "s"(base)
        : "memory");
}

// Position of one staging stream (which tile / K-tile its next units come from).
struct Cursor {
    int ti, kt;
    const char* abase;     // Act + m0 * lda * 2   (wave-uniform)
    const char* wbase;     // W + n0 * ldw * 2
    uint32_t xo[2], yo[2]; // per-lane byte offsets of the two 64-row rounds of this stream's X / Y unit
};

// The 16-byte output-row stores.  NT = non-temporal hint: measured (round 5, twin A/B on the ViT-B shapes, 2048 tiles) -2 ... -3 %
// on the plain-store epilogues (qkv: its 3.6 MB of output per 32-tile wave of an XCD is read next by another kernel, from
// HBM either way, and without the hint it evicts the weight and activation panels the XCD is re-reading), +-0.5 % on fc1 /
// proj / fc2, which keep the default policy.  sc1 (write through, drop the line) measured the same as nt on qkv and slightly
// worse elsewhere.  The s_nop: a store of more than 8 bytes followed by a write of its data registers needs wait states
// that the compiler inserts for its own stores only (without it the 140-case bit-equality check fails).
template <bool NT> __device__ __forceinline__ void out_store16(void* p, u32x4 v) {
    if constexpr (NT) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else *(u32x4*)p = v;
}
#define AP_OUT_STORE(PTR, V) out_store16<kStoreNT>((PTR), (V))

template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) char smem[kLdsBytes];

    constexpr bool kNorm = EPI == EPI_NORM_STORE || EPI == EPI_NORM_GELU || EPI == EPI_NORM_SWIGLU || EPI == EPI_NORM_QGELU;
    constexpr bool kSwiglu = EPI == EPI_NORM_SWIGLU;       // x1 | x2 in the wave's two n blocks -> 32 gated output columns
    constexpr bool kPatch = EPI == EPI_PATCH_STREAM;
    constexpr bool kRes = EPI == EPI_RESID_STATS || kPatch;
    constexpr bool kStoreNT = !kRes && EPI != EPI_BIAS_RESID;       // every epilogue whose 16-bit output another kernel reads next
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int hi = lane >> 5, l31 = lane & 31;
    using Frag = typename Mma<T>::Frag;

    // ---- tile list (XCD-aware, see header)
    TileWalk tw;
    {
        const int tiles_m = (g.M + kBM - 1) / kBM;
        tw.tiles_n = g.N / kBN;
        tw.tiles_m = tiles_m;
        // default: whole rows up to 9 column tiles (qkv); wider problems (fc1: 12) in groups of 6 -- measured
        // +1.8 % on fc1, and fewer L2 misses (each XCD-wave touches 6 weight panels instead of 12)
        tw.group = g.walk_cols > 0 && g.walk_cols < tw.tiles_n ? g.walk_cols : (tw.tiles_n > 9 ? 6 : tw.tiles_n);
        const int total = tiles_m * tw.tiles_n;
        const int nblk = gridDim.x, b = blockIdx.x;
        const int nx = nblk < 8 ? nblk : 8;                    // XCDs that received workgroups
        const int xcd = b % nx, j = b / nx;
        const int nwx = (nblk - xcd + nx - 1) / nx;            // workgroups on this XCD
        const int t0 = (int)(((long)total * xcd) / nx), t1 = (int)(((long)total * (xcd + 1)) / nx);
        tw.first = t0 + j;
        tw.stride = nwx;
        tw.count = tw.first < t1 ? (t1 - tw.first + nwx - 1) / nwx : 0;
    }
    if (tw.count == 0) return;
    const int nk = g.K / 64;
    const float inv_p = kPatch ? 1.0f / (float)g.P : 0.0f;
    // Start-time skew: equal tiles keep every CU in step, so all epilogues (the HBM write bursts)
    // would coincide.  Workgroup j of an XCD starts j / nwx of a tile period late; the early ones are
    // the ones that own one tile more.
    // (diagnostics -- start skew, per-tile time stamps, ablation flags -- exist only under -DAP_G256_DIAG: the twin, impl 257)
    if constexpr (kDiag) {
        if (g.skew_ticks > 0) {
            const int nx = gridDim.x < 8 ? gridDim.x : 8;
            const long long until = (long long)__builtin_amdgcn_s_memrealtime() +
                                    (long long)g.skew_ticks * (int)(blockIdx.x / nx) / tw.stride;
            while ((long long)__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(16);
        }
    }

    // ---- staging plan.  Round j of a unit covers unit rows j*64 + wave*8 + (lane >> 3); the lane's
    //      16-byte source chunk is swizzled by the LDS row it lands on.
    const int srow = wave * 8 + (lane >> 3);                                 // unit row within the round
    const uint32_t schunk = (uint32_t)(((lane & 7) ^ ((srow >> 1) & 7)) << 4);
    auto set_tile = [&](Cursor& c, int q, int ti) {
        c.ti = ti;
        const int id = tw.first + ti * tw.stride;
        int tr, tc;
        tw.rc(id, tr, tc);
        const int m0 = tr * kBM, n0 = tc * kBN;
        c.abase = (const char*)g.A + (size_t)m0 * g.lda * 2;
        c.wbase = (const char*)g.W + (size_t)n0 * g.ldw * 2;
        const int mlast = g.M - 1 - m0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int xr_ = j * 128 + q * 64 + srow;                                // tile m-row
            xr_ = xr_ < mlast ? xr_ : mlast;
            const int yr_ = (j * 2 + (wave >> 2)) * 64 + q * 32 + (wave & 3) * 8 + (lane >> 3);   // tile n-row
            c.xo[j] = (uint32_t)xr_ * (uint32_t)(g.lda * 2) + schunk;
            c.yo[j] = (uint32_t)yr_ * (uint32_t)(g.ldw * 2) + schunk;
        }
    };
    auto advance = [&](Cursor& c, int q) {
        if (++c.kt == nk) {
            if (c.ti + 1 < tw.count) { c.kt = 0; set_tile(c, q, c.ti + 1); }
            else c.kt = nk - 1;                       // past the end: re-stage the last K-tile (harmless)
        }
    };
    const uint32_t lds_wave = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 1024;
    [[maybe_unused]] bool dma_off = false;
    auto stage = [&](const Cursor& c, bool is_x, int buf, int unit) {
        if constexpr (kDiag) { if ((g.ablate & 4) && dma_off) return; }
        const char* base = (is_x ? c.abase : c.wbase) + (size_t)c.kt * kRowBytes;
        dma_unit(base, is_x ? c.xo[0] : c.yo[0], is_x ? c.xo[1] : c.yo[1],
                 lds_wave + buf * kBufBytes + unit * kUnitBytes);
    };

    Cursor ca, cb;            // ca: X0 / Y0 stream (two K-tiles ahead), cb: Y1 / X1 stream (one ahead)
    ca.kt = cb.kt = 0;
    set_tile(ca, 0, 0);
    set_tile(cb, 1, 0);

    // ---- fragment read addresses (byte offsets into a buffer)
    const int xr = (l31 >> 1) & 7;
    int pa[4], pb[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int co = ((kk * 2 + hi) ^ xr) << 4;
        pa[kk] = (wr * 64 + l31) * kRowBytes + co;          // + unit * 16K + rb * 32 rows
        pb[kk] = (wc * 32 + l31) * kRowBytes + co;
    }

    // Accumulators start from the bias (lane's 32 n values, the same for its four m blocks), so the
    // epilogue has no bias pass; the next tile's bias is fetched while the current epilogue runs.
    f32x16 acc[2][4];         // [n block of 32][m block of 32]
    f32x4 nbias[2][4];
    // fused-LayerNorm epilogues (EPI_NORM_*): the accumulators start from zero, and the CURRENT tile's bias (in nbias),
    // column sums and row statistics are requested before the drain that opens its epilogue
    f32x4 ncs[2][4];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 rst[4];
    u32x4 res[16];            // EPI_RESID_STATS: the wave's 128 x 64 window of the stream, row-major 16 B per lane
    u32x2 pk[2][4][4];        //   and the accumulators rounded to T ([n block][m block][group of 4 n])
    // The bias loads are inline asm with hand-placed waits.  A load hipcc tracks that is still pending at the
    // tile-loop header makes its waitcnt pass put static `s_waitcnt vmcnt(0..4)` in front of the first MFMAs of
    // EVERY tile (the first-entry state is merged into the back edge).  The waits are FULL drains placed where only
    // loads are outstanding: vmcnt retires loads in order among themselves, but a store may retire before an older
    // load, so "all but the N youngest" proves nothing once stores are in the queue (a counted wait behind the
    // epilogue's stores let a late bias load slip through about once in 10^5 tiles: one or two wrong images per
    // few hundred forwards).  The next tile's bias is therefore requested just before the drain that opens the
    // epilogue (the main loop's 64 fragment registers are free by then) and is complete when the drain returns.
    auto bias_ptr = [&](int ti, int hi_) {
        int tr, tc;
        tw.rc(tw.first + ti * tw.stride, tr, tc);
        return g.bias + tc * kBN + wc * 64 + hi_ * 4;
    };
#define AP_BIAS_LD(P, NB, G4) \
    asm volatile("global_load_dwordx4 %0, %1, off offset:" #NB "*128+" #G4 "*32" : "=&v"(nbias[NB][G4]) : "v"(P) : "memory")
#define AP_BIAS_LD8(P)                                                                  \
    AP_BIAS_LD(P, 0, 0); AP_BIAS_LD(P, 0, 1); AP_BIAS_LD(P, 0, 2); AP_BIAS_LD(P, 0, 3); \
    AP_BIAS_LD(P, 1, 0); AP_BIAS_LD(P, 1, 1); AP_BIAS_LD(P, 1, 2); AP_BIAS_LD(P, 1, 3)
#define AP_BIAS_WAIT(N)                                                                                           \
    asm volatile("s_waitcnt vmcnt(" #N ")"                                                                        \
                 : "+v"(nbias[0][0]), "+v"(nbias[0][1]), "+v"(nbias[0][2]), "+v"(nbias[0][3]), "+v"(nbias[1][0]), \
                   "+v"(nbias[1][1]), "+v"(nbias[1][2]), "+v"(nbias[1][3])::"memory")
    auto init_acc = [&]() {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[nb][mb][e] = kNorm ? 0.0f : nbias[nb][e >> 2][e & 3];
    };
    if constexpr (!kNorm) {
        const float* bp0 = bias_ptr(0, hi);
        AP_BIAS_LD8(bp0);
        AP_BIAS_WAIT(0);
    }

    Frag fa[2][4], fb0[4], fb1[4];

    // ---- prologue: K-tile 0 complete, X0 / Y0 of K-tile 1
    stage(ca, true, 0, U_X0); stage(ca, false, 0, U_Y0); advance(ca, 0);
    stage(cb, false, 0, U_Y1); stage(cb, true, 0, U_X1); advance(cb, 1);
    stage(ca, true, 1, U_X0); stage(ca, false, 1, U_Y0); advance(ca, 0);
    AP_VMCNT(8);              // X0 / Y0 of K-tile 0 have landed (phase 0 reads them)
    __builtin_amdgcn_s_barrier();
    init_acc();
    if constexpr (kDiag) dma_off = true;

    // Phase boundary.  The counted wait retires what was staged four phases ago (allowed in flight:
    // the 3 x 2 loads of the last three phases); the barrier publishes it and orders this phase's
    // DMA (issued after it) behind every wave's fragment reads of the unit it overwrites.  The
    // fragment reads written before it may be hoisted by the compiler into the previous phase's
    // MFMA block (everything they read was published by the previous barrier); the wait may not.
    // ablation twin (-DAP_G256_DIAG; timing only, results are wrong when a flag is set; ap_gemm impl 257, variant bits 0-2):
    // 1 = no counted wait, 2 = no barrier, 4 = no LDS-DMA after the prologue.  Measured (fc2, K = 3072):
    // none 1.005 ms, no wait 1.013, no barrier 0.947, no DMA 0.940, all three 0.812 -> the fragment-read ->
    // MFMA dependence inside a wave, not the synchronisation, bounds this structure at ~1.17 PF/s.
    // A second barrier per phase with the two wave rows half a phase apart (+ s_setprio) was also measured: -1 %.
    [[maybe_unused]] const bool abl_nowait = kDiag && (g.ablate & 1) != 0, abl_nobar = kDiag && (g.ablate & 2) != 0;
#define AP_PHASE_SYNC()                                       \
    __builtin_amdgcn_sched_barrier(0);                        \
    if (wait && !(kDiag && abl_nowait)) AP_VMCNT(6);          \
    __builtin_amdgcn_sched_barrier(0);                        \
    if (!(kDiag && abl_nobar)) __builtin_amdgcn_s_barrier();  \
    __builtin_amdgcn_sched_barrier(0)
#define AP_MMA(ACC, B, A) ACC = Mma<T>::run(B, A, ACC)

    // (the loop of rounds 2-4: still what EPI_PATCH_STREAM runs -- its epilogue keeps more registers live across the K loop and
    //  the pipelined form below would spill 5-11 of them)
    // wait = false only for the first K-tile after an epilogue: everything staged before the epilogue
    // was drained there (vmcnt(0)), so its four phases need no counted wait and the epilogue's own
    // stores keep draining under them.
    [[maybe_unused]] auto ktile = [&](auto bufc, const bool wait) {
        constexpr int BUF = decltype(bufc)::value;
        const char* buf = smem + BUF * kBufBytes;
        // ---------------- phase 0: (X0, Y0), stage Y1 of the next K-tile
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            fb0[kk] = *(const Frag*)(buf + U_Y0 * kUnitBytes + pb[kk]);
            fa[0][kk] = *(const Frag*)(buf + U_X0 * kUnitBytes + pa[kk]);
            fa[1][kk] = *(const Frag*)(buf + U_X0 * kUnitBytes + 32 * kRowBytes + pa[kk]);
        }
        AP_PHASE_SYNC();
        AP_MMA(acc[0][0], fb0[0], fa[0][0]); AP_MMA(acc[0][1], fb0[0], fa[1][0]);
        stage(cb, false, BUF ^ 1, U_Y1);
        AP_MMA(acc[0][0], fb0[1], fa[0][1]); AP_MMA(acc[0][1], fb0[1], fa[1][1]);
        AP_MMA(acc[0][0], fb0[2], fa[0][2]); AP_MMA(acc[0][1], fb0[2], fa[1][2]);
        AP_MMA(acc[0][0], fb0[3], fa[0][3]); AP_MMA(acc[0][1], fb0[3], fa[1][3]);
        // ---------------- phase 1: (X0, Y1), stage X1 of the next K-tile
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fb1[kk] = *(const Frag*)(buf + U_Y1 * kUnitBytes + pb[kk]);
        AP_PHASE_SYNC();
        AP_MMA(acc[1][0], fb1[0], fa[0][0]); AP_MMA(acc[1][1], fb1[0], fa[1][0]);
        stage(cb, true, BUF ^ 1, U_X1);
        advance(cb, 1);
        AP_MMA(acc[1][0], fb1[1], fa[0][1]); AP_MMA(acc[1][1], fb1[1], fa[1][1]);
        AP_MMA(acc[1][0], fb1[2], fa[0][2]); AP_MMA(acc[1][1], fb1[2], fa[1][2]);
        AP_MMA(acc[1][0], fb1[3], fa[0][3]); AP_MMA(acc[1][1], fb1[3], fa[1][3]);
        // ---------------- phase 2: (X1, Y1), stage X0 of the K-tile after next
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            fa[0][kk] = *(const Frag*)(buf + U_X1 * kUnitBytes + pa[kk]);
            fa[1][kk] = *(const Frag*)(buf + U_X1 * kUnitBytes + 32 * kRowBytes + pa[kk]);
        }
        AP_PHASE_SYNC();
        AP_MMA(acc[1][2], fb1[0], fa[0][0]); AP_MMA(acc[1][3], fb1[0], fa[1][0]);
        stage(ca, true, BUF, U_X0);
        AP_MMA(acc[1][2], fb1[1], fa[0][1]); AP_MMA(acc[1][3], fb1[1], fa[1][1]);
        AP_MMA(acc[1][2], fb1[2], fa[0][2]); AP_MMA(acc[1][3], fb1[2], fa[1][2]);
        AP_MMA(acc[1][2], fb1[3], fa[0][3]); AP_MMA(acc[1][3], fb1[3], fa[1][3]);
        // ---------------- phase 3: (X1, Y0)   (Y0 fragment still in registers), stage Y0 likewise
        AP_PHASE_SYNC();
        AP_MMA(acc[0][2], fb0[0], fa[0][0]); AP_MMA(acc[0][3], fb0[0], fa[1][0]);
        stage(ca, false, BUF, U_Y0);
        advance(ca, 0);
        AP_MMA(acc[0][2], fb0[1], fa[0][1]); AP_MMA(acc[0][3], fb0[1], fa[1][1]);
        AP_MMA(acc[0][2], fb0[2], fa[0][2]); AP_MMA(acc[0][3], fb0[2], fa[1][2]);
        AP_MMA(acc[0][2], fb0[3], fa[0][3]); AP_MMA(acc[0][3], fb0[3], fa[1][3]);
    };

    constexpr bool kOldLoop = kPatch;
    // Software-pipelined fragment reads (round 5): the reads of a phase are issued BETWEEN the MFMAs of the phase before it, each as soon
    // as the last MFMA that uses its destination register has been issued (program order pinned with sched_barrier), so every
    // phase opens with MFMAs whose operands arrived long ago instead of with all eight waves' ds_read burst and the
    // latency of its first reply.  Nothing is read earlier than one barrier after the wait that retired its staging (same
    // argument as above: Y1 of this K-tile behind this K-tile's phase-0 barrier, X1 behind its phase-1 barrier, X0 / Y0 of the
    // NEXT K-tile behind its phase-3 barrier), and units are re-staged no earlier than before.  FIRST: the K-tile that opens a
    // tile reads its phase-0 fragments up front (the fragment registers are the epilogue's scratch between tiles);
    // PREFETCH = false: the K-tile that closes a tile reads nothing ahead.
#define AP_SB() __builtin_amdgcn_sched_barrier(0)
    // Measured on top of this form and dropped (round 5, profiles/r05_gemm_experiments.txt): barriers only in front of phases 1 and 3
    // (-1 ... +0.5 %: with the operands already in registers the barriers cost nothing -- the twin's no-barrier ablation is
    // SLOWER than the product), and the two waves of a SIMD issuing a phase's LDS-DMA two MFMA pairs apart (+1 ... +3 %).
    // fragment addresses of both K-tile buffers in registers (the second buffer starts past ds_read's 16-bit offset field:
    // without these every read of it is preceded by a v_add / v_or in the MFMA stream)
    int pa1[4], pb1[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { pa1[kk] = pa[kk] + kBufBytes; pb1[kk] = pb[kk] + kBufBytes; }
#define AP_PA(B, kk) ((B) ? pa1[kk] : pa[kk])
#define AP_PB(B, kk) ((B) ? pb1[kk] : pb[kk])
#define AP_FRAG(UNIT_OFF, P) (*(const Frag*)(smem + (UNIT_OFF) + (P)))
    [[maybe_unused]] auto ktile_p = [&](auto bufc, auto firstc, auto prefc, const bool wait) {
        constexpr int BUF = decltype(bufc)::value, NBUF = BUF ^ 1;
        constexpr bool FIRST = decltype(firstc)::value, PREFETCH = decltype(prefc)::value;
        // ---------------- phase 0: (X0, Y0), stage Y1 of the next K-tile; read Y1 of this one
        if constexpr (FIRST) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                fb0[kk] = AP_FRAG(U_Y0 * kUnitBytes, AP_PB(BUF, kk));
                fa[0][kk] = AP_FRAG(U_X0 * kUnitBytes, AP_PA(BUF, kk));
                fa[1][kk] = AP_FRAG(U_X0 * kUnitBytes + 32 * kRowBytes, AP_PA(BUF, kk));
            }
        }
        AP_PHASE_SYNC();
        AP_MMA(acc[0][0], fb0[0], fa[0][0]); AP_MMA(acc[0][1], fb0[0], fa[1][0]);
        stage(cb, false, NBUF, U_Y1);
        AP_SB();
        fb1[0] = AP_FRAG(U_Y1 * kUnitBytes, AP_PB(BUF, 0));
        fb1[1] = AP_FRAG(U_Y1 * kUnitBytes, AP_PB(BUF, 1));
        AP_SB();
        AP_MMA(acc[0][0], fb0[1], fa[0][1]); AP_MMA(acc[0][1], fb0[1], fa[1][1]);
        AP_SB();
        fb1[2] = AP_FRAG(U_Y1 * kUnitBytes, AP_PB(BUF, 2));
        fb1[3] = AP_FRAG(U_Y1 * kUnitBytes, AP_PB(BUF, 3));
        AP_SB();
        AP_MMA(acc[0][0], fb0[2], fa[0][2]); AP_MMA(acc[0][1], fb0[2], fa[1][2]);
        AP_MMA(acc[0][0], fb0[3], fa[0][3]); AP_MMA(acc[0][1], fb0[3], fa[1][3]);
        // ---------------- phase 1: (X0, Y1), stage X1 of the next K-tile; read X1 of this one into fa as its registers free up
        AP_PHASE_SYNC();
        AP_MMA(acc[1][0], fb1[0], fa[0][0]); AP_MMA(acc[1][1], fb1[0], fa[1][0]);
        stage(cb, true, NBUF, U_X1); advance(cb, 1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            AP_SB();
            fa[0][kk] = AP_FRAG(U_X1 * kUnitBytes, AP_PA(BUF, kk));
            fa[1][kk] = AP_FRAG(U_X1 * kUnitBytes + 32 * kRowBytes, AP_PA(BUF, kk));
            AP_SB();
            if (kk < 3) { AP_MMA(acc[1][0], fb1[kk + 1], fa[0][kk + 1]); AP_MMA(acc[1][1], fb1[kk + 1], fa[1][kk + 1]); }
        }
        // ---------------- phase 2: (X1, Y1), stage X0 of the K-tile after next
        AP_PHASE_SYNC();
        AP_MMA(acc[1][2], fb1[0], fa[0][0]); AP_MMA(acc[1][3], fb1[0], fa[1][0]);
        stage(ca, true, BUF, U_X0);
        AP_SB();
        AP_MMA(acc[1][2], fb1[1], fa[0][1]); AP_MMA(acc[1][3], fb1[1], fa[1][1]);
        AP_MMA(acc[1][2], fb1[2], fa[0][2]); AP_MMA(acc[1][3], fb1[2], fa[1][2]);
        AP_MMA(acc[1][2], fb1[3], fa[0][3]); AP_MMA(acc[1][3], fb1[3], fa[1][3]);
        // ---------------- phase 3: (X1, Y0), stage Y0 likewise; read X0 / Y0 of the NEXT K-tile (other buffer)
        AP_PHASE_SYNC();
        AP_MMA(acc[0][2], fb0[0], fa[0][0]); AP_MMA(acc[0][3], fb0[0], fa[1][0]);
        stage(ca, false, BUF, U_Y0); advance(ca, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if constexpr (PREFETCH) {
                AP_SB();
                fb0[kk] = AP_FRAG(U_Y0 * kUnitBytes, AP_PB(NBUF, kk));
                fa[0][kk] = AP_FRAG(U_X0 * kUnitBytes, AP_PA(NBUF, kk));
                fa[1][kk] = AP_FRAG(U_X0 * kUnitBytes + 32 * kRowBytes, AP_PA(NBUF, kk));
                AP_SB();
            }
            if (kk < 3) { AP_MMA(acc[0][2], fb0[kk + 1], fa[0][kk + 1]); AP_MMA(acc[0][3], fb0[kk + 1], fa[1][kk + 1]); }
        }
    };

    char* scr = smem + kScratchOff + wave * 4096;
    // Measured and dropped (round 5, profiles/r05_gemm_experiments.txt): EPI_RESID_STATS reads its 128 x 64 window of the stream
    // (128 KiB per workgroup) when the tile's K loop is over -- the "drain" of proj / fc2 is 4.2-4.7 us per tile against
    // 1.3-1.5 us for the NORM epilogues.  Touching every line of the window four K-tiles early (two LDS-DMA loads per wave into
    // the idle epilogue scratch, so that the epilogue's loads hit the L2) made proj 11 % and fc2 1.6 % SLOWER: the window is
    // already served from the XCD's L2 / the Infinity Cache (the previous kernel wrote it), and the extra loads sit in
    // the same queue as the operand stream.
    auto stamp = [&](int ti, int k) {
        if constexpr (kDiag) {
            if (g.trace && ti < g.trace_tiles && threadIdx.x == 0)
                g.trace[((size_t)blockIdx.x * g.trace_tiles + ti) * 8 + k] = (long long)__builtin_amdgcn_s_memrealtime();
        }
    };
    auto stamp_clk = [&](int ti, int k) {       // shader-clock counter: (slot 6 - slot 5) / (slot 1 - slot 0) = core clock in the main loop
        if constexpr (kDiag) {
            if (g.trace && ti < g.trace_tiles && threadIdx.x == 0)
                g.trace[((size_t)blockIdx.x * g.trace_tiles + ti) * 8 + k] = (long long)__builtin_amdgcn_s_memtime();
        }
    };
    // NORM epilogues (round 5; -DAP_G256_NO_LDSOPS restores the loads + drain of rounds 2-4: qkv 4.5 %, fc1 3.2 % slower): this tile's row statistics (128 rows x 8 B), bias and column sums (64 floats each) are brought into the wave's
    // idle epilogue scratch by three LDS-DMA loads at the START of the tile's K loop (older than every staging load that the
    // loop's counted waits leave in flight, so they have landed long before the epilogue), and the epilogue reads them with
    // ds_read instead of opening with 20 global loads and a full drain.  The drain (the first K-tile after an epilogue runs
    // without counted waits, so everything staged must be back before the epilogue's stores go out) moves in front of the
    // tile's FIRST store, behind the first 32-row block's arithmetic.  A tile that holds the last row of an odd M keeps the
    // loads (the 16-byte row-statistics pieces cover row pairs).
    constexpr bool kLdsOps = kNorm;
    const uint32_t scr_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + kScratchOff + wave * 4096;
    auto tile_uses_lds_ops = [&](int m0_tile) { return !((g.M & 1) && m0_tile + kBM > g.M); };
    [[maybe_unused]] auto stage_ops = [&](int ti) {
        int tr, tc;
        tw.rc(tw.first + ti * tw.stride, tr, tc);
        if (!tile_uses_lds_ops(tr * kBM)) return;
        const int m0 = tr * kBM + wr * 128, n0 = tc * kBN + wc * 64;
        int p = m0 + 2 * lane;
        p = p + 1 < g.M ? p : g.M - 2;
        p = p < 0 ? 0 : p;
        const uint32_t off_rs = (uint32_t)p * 8u, off_b = (uint32_t)(n0 + lane) * 4u;
        uint32_t keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %6\n\t"
            "s_mov_b32 m0, %4\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dword %2, %7\n\t"
            "s_mov_b32 m0, %5\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dword %2, %8\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(off_rs), "v"(off_b), "s"(scr_lds), "s"(scr_lds + 1024), "s"(scr_lds + 1280), "s"(g.rowstats), "s"(g.bias), "s"(g.colsum)
            : "memory");
    };
    if constexpr (kLdsOps) stage_ops(0);
    for (int ti = 0; ti < tw.count; ++ti) {
        stamp(ti, 0);
        stamp_clk(ti, 5);
        if constexpr (kLdsOps) { if (ti > 0) stage_ops(ti); }
        if constexpr (!kOldLoop) {
            using B0 = std::integral_constant<int, 0>; using B1 = std::integral_constant<int, 1>;
            using Y = std::true_type; using N = std::false_type;
            ktile_p(B0{}, Y{}, Y{}, ti == 0);
            for (int kt = 2; kt < nk; kt += 2) {
                ktile_p(B1{}, N{}, Y{}, true);
                ktile_p(B0{}, N{}, Y{}, true);
            }
            ktile_p(B1{}, N{}, N{}, true);
        } else {
            for (int kt = 0; kt < nk; kt += 2) {
                if constexpr (kDiag) {
                    // fine timeline (twin only): start of every K-tile pair, in a second [workgroups, tiles, 8] block of the buffer
                    if (g.trace && ti < g.trace_tiles && threadIdx.x == 0 && (kt >> 1) < 8)
                        g.trace[((size_t)(gridDim.x + blockIdx.x) * g.trace_tiles + ti) * 8 + (kt >> 1)] =
                            (long long)__builtin_amdgcn_s_memrealtime();
                }
                ktile(std::integral_constant<int, 0>{}, ti == 0 || kt != 0);
                ktile(std::integral_constant<int, 1>{}, true);
            }
        }
        // ---------------- epilogue
        stamp_clk(ti, 6);
        stamp(ti, 1);
        // lane-derived epilogue constants are recomputed per tile from a FRESH lane id (v_mbcnt: no register has to stay
        // live -- or be spilled, as an opaque copy of `lane` was in the BIAS_GELU instantiation -- across the K loop)
        // (the v_mbcnt pair itself is volatile asm: through the builtins hipcc hoisted it in front of the tile loop in the
        // BIAS_QGELU instantiation and spilled THAT register)
        int lane_e;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
        const int hi = lane_e >> 5, l31 = lane_e & 31;
        const int id = tw.first + ti * tw.stride;
        int tr, tc;
        tw.rc(id, tr, tc);
        const int m0 = tr * kBM + wr * 128, n0 = tc * kBN + wc * 64;
        const int rrow = lane_e >> 3, rc = lane_e & 7;
        // The fused-LayerNorm epilogues fetch their operands with PLAIN loads followed by an explicit s_waitcnt built with
        // the compiler's own builtin: the waitcnt pass sees that instruction and knows nothing is pending afterwards, and
        // no register of a load can be copied before the wait (an inline-asm load with a separate inline-asm wait lets the
        // register allocator place such copies in between: seen as a run-to-run race).  vmcnt(0) = 0x0F70 on gfx9
        // (expcnt / lgkmcnt fields left at their maxima).  Only loads are outstanding here.
        bool drain_late = false;
        if constexpr (kLdsOps) drain_late = tile_uses_lds_ops(tr * kBM);
        if (kLdsOps && drain_late) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    nbias[nb][g4] = *(const f32x4*)(scr + 1024 + (nb * 32 + g4 * 8 + hi * 4) * 4);
                    ncs[nb][g4] = *(const f32x4*)(scr + 1280 + (nb * 32 + g4 * 8 + hi * 4) * 4);
                }
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) rst[mb] = *(const f32x2*)(scr + (mb * 32 + l31) * 8);
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (kNorm) {
            // THIS tile's bias, column sums and row statistics
            __builtin_amdgcn_sched_barrier(0);
            const float* bp = bias_ptr(ti, hi);
            const float* cp = g.colsum + (bp - g.bias);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    nbias[nb][g4] = *(const f32x4*)(bp + nb * 32 + g4 * 8);
                    ncs[nb][g4] = *(const f32x4*)(cp + nb * 32 + g4 * 8);
                }
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                int m = m0 + mb * 32 + l31;
                m = m < g.M ? m : g.M - 1;
                rst[mb] = *(const f32x2*)(g.rowstats + 2 * (size_t)m);
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (kRes) {
            // next tile's bias and this tile's window of the stream (16 B per lane and row: the layout the transposed
            // stores use)
            __builtin_amdgcn_sched_barrier(0);
            const float* nbp = bias_ptr(ti + 1 < tw.count ? ti + 1 : ti, hi);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) nbias[nb][g4] = *(const f32x4*)(nbp + nb * 32 + g4 * 8);
            auto res_ld = [&](int j) {
                int m = m0 + (j >> 2) * 32 + (j & 3) * 8 + rrow;
                m = m < g.M ? m : g.M - 1;
                if constexpr (kPatch) {                              // "residual" = the patch's position embedding row
                    int q = (int)((float)m * inv_p);                 // m / P for m < 2^24 (float estimate, corrected)
                    int p = m - q * g.P;
                    p = p < 0 ? p + g.P : (p >= g.P ? p - g.P : p);
                    res[j] = *(const u32x4*)((const T*)g.pos16 + (size_t)(g.pos_row0 + p) * g.N + n0 + rc * 8);
                } else {
                    res[j] = *(const u32x4*)((const T*)g.out + (size_t)m * g.ldo + n0 + rc * 8);
                }
            };
#pragma unroll
            for (int j = 0; j < 8; ++j) res_ld(j);
            __builtin_amdgcn_sched_barrier(0);
            // exact class rows (vit.cpp): a row m = img * cls_tokens also leaves its unrounded accumulators (bias included) in
            // cls_branch[img].  At most one lane in 197 (ViT-B) takes the branch; m / cls_tokens by a float estimate, corrected.
            if constexpr (!kPatch) {
                if (g.cls_tokens > 0) {
                    const float inv_t = 1.0f / (float)g.cls_tokens;
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) {
                        const int m = m0 + mb * 32 + l31;
                        int q = (int)((float)m * inv_t);
                        const int r = m - q * g.cls_tokens;
                        q = r < 0 ? q - 1 : (r >= g.cls_tokens ? q + 1 : q);
                        if (m == q * g.cls_tokens && m < g.M) {
                            float* cbr = g.cls_branch + (size_t)q * g.N + n0 + hi * 4;
#pragma unroll
                            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                                for (int g4 = 0; g4 < 4; ++g4) {
                                    f32x4 v;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] = acc[nb][mb][g4 * 4 + e];
                                    *(f32x4*)(cbr + nb * 32 + g4 * 8) = v;
                                }
                        }
                    }
                }
            }
            // The accumulators (bias included) are rounded to T between the two groups of loads: the first group and the
            // bias land in the main loop's 64 fragment registers, the packing frees 64 more for the second group.
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[nb][mb][g4 * 4 + e];
                        pk[nb][mb][g4] = pack4<T>(v);
                    }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 8; j < 16; ++j) res_ld(j);
            __builtin_amdgcn_sched_barrier(0);
            // (measured, round 5: leaving the waits for these loads to the compiler -- counted vmcnt(15 .. 12) in front of the first
            //  32-row block, full drains once its stores are in flight -- changes nothing: proj / fc2 +0.1 %.  The 4-5 us this
            //  drain takes per tile are the window's 128 KiB per workgroup at a CU's share of the memory system, not one latency.)
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            // next tile's bias (its accumulators start from it), then ONE drain: the stream staged so far (next
            // tile's first K-tiles) and the bias have landed; only loads are outstanding here
            const float* nbp = bias_ptr(ti + 1 < tw.count ? ti + 1 : ti, hi);
            AP_BIAS_LD8(nbp);
            AP_BIAS_WAIT(0);
        }
        stamp(ti, 2);
        if constexpr (EPI == EPI_NORM_GELU) {
            // the GELU routine takes y * kGeluS (its clamp is the packed multiply's CLAMP bit, ap_common.h): scale the tile's bias
            // registers and the rows' (rstd, -mean rstd) once -- 24 packed multiplies per tile and wave instead of 128 v_min
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) nbias[nb][g4] *= kGeluS;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) rst[mb] *= kGeluS;
        }
        const bool has_gamma = (EPI == EPI_BIAS_STORE || EPI == EPI_BIAS_RESID) && g.gamma != nullptr;
        const float* gp = g.gamma + n0 + hi * 4;
        if constexpr (kDiag) { if (g.trace) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(ti, 3); } }
        // ablation bit 8 (twin, timing only): no epilogue body at all (drain and accumulator restart stay) = the bound on what
        // hiding the epilogue behind another tile's MFMAs could return
        if (kDiag && (g.ablate & 8)) {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) asm volatile("" ::"v"(acc[nb][mb]));
        } else
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            if constexpr (EPI == EPI_BIAS_RESID) {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[nb][mb][g4 * 4 + e];
                        if (has_gamma) {
                            const f32x4 ga = *(const f32x4*)(gp + nb * 32 + g4 * 8);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] *= ga[e];
                        }
                        *(f32x4*)(scr + l31 * 128 + (((g4 * 2 + hi) ^ (l31 & 7)) << 4)) = v;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = i * 8 + rrow;
                        const f32x4 v = *(const f32x4*)(scr + row * 128 + ((rc ^ (row & 7)) << 4));
                        const int m = m0 + mb * 32 + row;
                        if (m < g.M) {
                            float* dst = (float*)g.out + (size_t)m * g.ldo + n0 + nb * 32 + rc * 4;
                            f32x4 r = *(const f32x4*)dst;
#pragma unroll
                            for (int e = 0; e < 4; ++e) r[e] += v[e];
                            *(f32x4*)dst = r;
                        }
                    }
                }
            } else if constexpr (kSwiglu) {
                // out[m][32 q + j] = silu(norm(x1)) * norm(x2): the lane holds both in acc[0] / acc[1] (interleaved weight rows).
                // The wave's block is 32 rows x 32 columns = 64 bytes per row: four 16-byte chunks, XOR-swizzled by row & 3
                const f32x2_t rs2 = {rst[mb][0], rst[mb][0]}, nm2 = {rst[mb][1], rst[mb][1]};
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    f32x4 y[2];
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        const f32x2_t lo = __builtin_elementwise_fma(rs2, f32x2_t{acc[nb][mb][g4 * 4 + 0], acc[nb][mb][g4 * 4 + 1]},
                            __builtin_elementwise_fma(nm2, f32x2_t{ncs[nb][g4][0], ncs[nb][g4][1]}, f32x2_t{nbias[nb][g4][0], nbias[nb][g4][1]}));
                        const f32x2_t hi2 = __builtin_elementwise_fma(rs2, f32x2_t{acc[nb][mb][g4 * 4 + 2], acc[nb][mb][g4 * 4 + 3]},
                            __builtin_elementwise_fma(nm2, f32x2_t{ncs[nb][g4][2], ncs[nb][g4][3]}, f32x2_t{nbias[nb][g4][2], nbias[nb][g4][3]}));
                        y[nb] = f32x4{lo[0], lo[1], hi2[0], hi2[1]};
                    }
                    const f32x2_t a = swiglu2(f32x2_t{y[0][0], y[0][1]}, f32x2_t{y[1][0], y[1][1]});
                    const f32x2_t b = swiglu2(f32x2_t{y[0][2], y[0][3]}, f32x2_t{y[1][2], y[1][3]});
                    *(u32x2*)(scr + l31 * 64 + ((g4 ^ (l31 & 3)) << 4) + hi * 8) = pack4<T>(f32x4{a[0], a[1], b[0], b[1]});
                }
                if (mb == 0 && drain_late) AP_VMCNT(0);        // (the drain that used to open the epilogue)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = i * 16 + (lane_e >> 2), ch = lane_e & 3;
                    const u32x4 v = *(const u32x4*)(scr + row * 64 + ((ch ^ (row & 3)) << 4));
                    const int m = m0 + mb * 32 + row;
                    if (m < g.M) *(u32x4*)((T*)g.out + (size_t)m * g.ldo + (n0 >> 1) + ch * 8) = v;
                }
            } else {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[nb][mb][g4 * 4 + e];
                        if constexpr (kNorm) {
                            // two values per instruction (v_pk_fma_f32): y = rstd * acc + (nmr * colsum + bias)
                            // (EPI_NORM_GELU: rst and nbias were multiplied by kGeluS above -> y * kGeluS, what the GELU routine takes)
                            const f32x2_t rs2 = {rst[mb][0], rst[mb][0]}, nm2 = {rst[mb][1], rst[mb][1]};
                            const f32x2_t lo = __builtin_elementwise_fma(rs2, f32x2_t{v[0], v[1]},
                                __builtin_elementwise_fma(nm2, f32x2_t{ncs[nb][g4][0], ncs[nb][g4][1]}, f32x2_t{nbias[nb][g4][0], nbias[nb][g4][1]}));
                            const f32x2_t hi2 = __builtin_elementwise_fma(rs2, f32x2_t{v[2], v[3]},
                                __builtin_elementwise_fma(nm2, f32x2_t{ncs[nb][g4][2], ncs[nb][g4][3]}, f32x2_t{nbias[nb][g4][2], nbias[nb][g4][3]}));
                            v = f32x4{lo[0], lo[1], hi2[0], hi2[1]};
                        }
                        if constexpr (EPI == EPI_BIAS_GELU) {
                            const f32x2_t lo = gelu_sigmoid_poly2(f32x2_t{v[0], v[1]}), hi2 = gelu_sigmoid_poly2(f32x2_t{v[2], v[3]});
                            v = f32x4{lo[0], lo[1], hi2[0], hi2[1]};
                        }
                        if constexpr (EPI == EPI_NORM_GELU) {
                            const f32x2_t lo = gelu_sigmoid_poly2_s(f32x2_t{v[0], v[1]}), hi2 = gelu_sigmoid_poly2_s(f32x2_t{v[2], v[3]});
                            v = f32x4{lo[0], lo[1], hi2[0], hi2[1]};
                        }
                        if constexpr (EPI == EPI_BIAS_QGELU || EPI == EPI_NORM_QGELU) {
                            const f32x2_t lo = quick_gelu2(f32x2_t{v[0], v[1]}), hi2 = quick_gelu2(f32x2_t{v[2], v[3]});
                            v = f32x4{lo[0], lo[1], hi2[0], hi2[1]};
                        }
                        if (has_gamma) {
                            const f32x4 ga = *(const f32x4*)(gp + nb * 32 + g4 * 8);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] *= ga[e];
                        }
                        if constexpr (kRes) *(u32x2*)(scr + l31 * 128 + (((nb * 4 + g4) ^ (l31 & 7)) << 4) + hi * 8) = pk[nb][mb][g4];
                        else *(u32x2*)(scr + l31 * 128 + (((nb * 4 + g4) ^ (l31 & 7)) << 4) + hi * 8) = pack4<T>(v);
                    }
                if (mb == 0 && drain_late) AP_VMCNT(0);        // (the drain that used to open the epilogue)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = i * 8 + rrow;
                    u32x4 v = *(const u32x4*)(scr + row * 128 + ((rc ^ (row & 7)) << 4));
                    const int m = m0 + mb * 32 + row;
                    size_t orow = (size_t)m;
                    if constexpr (kPatch) {                          // patch row -> stream row img * (P + R) + R + p
                        int q = (int)((float)m * inv_p);
                        const int p = m - q * g.P;
                        q = p < 0 ? q - 1 : (p >= g.P ? q + 1 : q);
                        orow = (size_t)m + (size_t)(q + 1) * g.R;
                    }
                    if constexpr (kRes) {
                        float s = 0.f, q = 0.f;
                        v = resid_add_stats<T>(v, res[mb * 4 + i], s, q);
                        s = sum8(s);
                        q = sum8(q);
                        if (m < g.M && rc == 0) {
                            f32x2 sq = {s, q};
                            *(f32x2*)(g.partial + (orow * (g.N >> 6) + (n0 >> 6)) * 2) = sq;
                        }
                    }
                    if (m < g.M) AP_OUT_STORE((T*)g.out + orow * g.ldo + n0 + rc * 8, v);
                }
            }
        }
        stamp(ti, 4);
        init_acc();
    }
    AP_VMCNT(0);
#undef AP_PHASE_SYNC
#undef AP_MMA
#undef AP_BIAS_LD
#undef AP_BIAS_LD8
#undef AP_BIAS_WAIT
}

template <typename T, int EPI>
int launch_epi(const GemmArgs& a, int num_cu, int variant, hipStream_t stream) {
    const int tiles = ((a.M + kBM - 1) / kBM) * (a.N / kBN);
    dim3 grid(tiles < num_cu ? tiles : num_cu), block(512);
    (void)variant;
    gemm256_kernel<T, EPI><<<grid, block, 0, stream>>>(a);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

template <typename T>
int launch_typed(int epilogue, const GemmArgs& a, int num_cu, int variant, hipStream_t stream) {
    switch (epilogue) {
        case EPI_BIAS_STORE: return launch_epi<T, EPI_BIAS_STORE>(a, num_cu, variant, stream);
        case EPI_BIAS_GELU: return launch_epi<T, EPI_BIAS_GELU>(a, num_cu, variant, stream);
        case EPI_BIAS_RESID: return launch_epi<T, EPI_BIAS_RESID>(a, num_cu, variant, stream);
        case EPI_NORM_STORE: return launch_epi<T, EPI_NORM_STORE>(a, num_cu, variant, stream);
        case EPI_NORM_GELU: return launch_epi<T, EPI_NORM_GELU>(a, num_cu, variant, stream);
        case EPI_NORM_SWIGLU: return launch_epi<T, EPI_NORM_SWIGLU>(a, num_cu, variant, stream);
        case EPI_NORM_QGELU: return launch_epi<T, EPI_NORM_QGELU>(a, num_cu, variant, stream);
        case EPI_BIAS_QGELU: return launch_epi<T, EPI_BIAS_QGELU>(a, num_cu, variant, stream);
        case EPI_RESID_STATS: return launch_epi<T, EPI_RESID_STATS>(a, num_cu, variant, stream);
        case EPI_PATCH_STREAM: return launch_epi<T, EPI_PATCH_STREAM>(a, num_cu, variant, stream);
    }
    set_error("gemm256: unsupported epilogue %d", epilogue);
    return AP_ERR_INVALID;
}

}  // namespace

#ifndef AP_G256_ALT
long long* g_gemm_trace = nullptr;
int g_gemm_trace_tiles = 0;
void set_gemm_trace(long long* buf, int tiles_per_wg) { g_gemm_trace = buf; g_gemm_trace_tiles = tiles_per_wg; }
#else
extern long long* g_gemm_trace;
extern int g_gemm_trace_tiles;
#endif

bool AP_G256_FN(gemm256_supports)(int dtype, int epilogue, const GemmArgs& a) {
    if (dtype != AP_F16 && dtype != AP_BF16) return false;
    if (epilogue != EPI_BIAS_STORE && epilogue != EPI_BIAS_GELU && epilogue != EPI_BIAS_RESID &&
        epilogue != EPI_NORM_STORE && epilogue != EPI_NORM_GELU && epilogue != EPI_NORM_SWIGLU && epilogue != EPI_RESID_STATS &&
        epilogue != EPI_NORM_QGELU && epilogue != EPI_BIAS_QGELU &&
        epilogue != EPI_PATCH_STREAM)
        return false;
    if (epilogue == EPI_PATCH_STREAM && (!a.partial || !a.pos16 || a.P <= 0 || a.R <= 0 || a.M >= (1 << 24))) return false;
    if ((epilogue == EPI_NORM_STORE || epilogue == EPI_NORM_GELU || epilogue == EPI_NORM_SWIGLU || epilogue == EPI_NORM_QGELU) &&
        (!a.colsum || !a.rowstats)) return false;
    if (epilogue == EPI_RESID_STATS && !a.partial) return false;
    if (a.N % kBN != 0 || a.K % 128 != 0 || a.K < 128) return false;
    if (((size_t)a.lda * 2) % 16 != 0 || ((size_t)a.ldw * 2) % 16 != 0) return false;
    if ((size_t)255 * a.lda * 2 + 128 > 0xffffffffull || (size_t)255 * a.ldw * 2 + 128 > 0xffffffffull) return false;
    // 16-byte row stores: output row stride and base must keep 16-byte alignment
    const size_t oes = epilogue == EPI_BIAS_RESID ? 4 : 2;
    if (((size_t)a.ldo * oes) % 16 != 0 || ((uintptr_t)a.out & 15) != 0) return false;
    return true;
}

int AP_G256_FN(launch_gemm256)(int dtype, int epilogue, const GemmArgs& a, int variant, hipStream_t stream) {
    static int num_cu = 0;
    if (num_cu == 0) {
        int dev = 0;
        AP_HIP_CHECK(hipGetDevice(&dev));
        hipDeviceProp_t prop;
        AP_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    AP_REQUIRE(AP_G256_FN(gemm256_supports)(dtype, epilogue, a), "gemm256: unsupported problem");
    GemmArgs b = a;
    b.trace = nullptr; b.trace_tiles = 0; b.skew_ticks = 0; b.ablate = 0;
    if constexpr (kDiag) {
        b.trace = g_gemm_trace; b.trace_tiles = g_gemm_trace_tiles;
        const int skew_pct = (variant >> 4) & 0xfff;
        const int tiles = ((a.M + kBM - 1) / kBM) * (a.N / kBN);
        // estimated tile period in 10-ns ticks: ~1.65 us per 64-deep K-tile + epilogue
        b.skew_ticks = tiles > num_cu ? (int)((long long)((a.K / 64) * 165 + 400) * skew_pct / 100) : 0;
        b.ablate = variant & 15;
    }
    if ((variant >> 16) & 15) b.walk_cols = (variant >> 16) & 15;
    variant &= 15;
    return dtype == AP_F16 ? launch_typed<f16>(epilogue, b, num_cu, variant, stream)
                           : launch_typed<bf16>(epilogue, b, num_cu, variant, stream);
}

}  // namespace ap
