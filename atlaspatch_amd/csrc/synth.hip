// Synthetic slide tiles rendered straight into HBM: the device twin of
// atlaspatch_amd/core/wsi/synth_pixels.py::render_region, bit-identical (32-bit mixing hash,
// int64 ellipse tests on a 16-pixel lattice).  Stands in for tile decode when benchmarking the
// device pipeline: HBM-bound, 3 bytes written per pixel.
#include "ap_common.h"

namespace ap {
namespace {

__device__ __forceinline__ uint32_t mix32(uint32_t a) {
    a ^= a >> 16; a *= 0x7FEB352Du; a ^= a >> 15; a *= 0x846CA68Bu; a ^= a >> 16;
    return a;
}

// xy == nullptr: ONE region of pw x ph level pixels whose level-0 corner is (rx, ry) (the whole-level read of the thumbnail
// path); otherwise n tiles of pw x ph at the corners in xy.
__global__ __launch_bounds__(256) void synth_tiles_kernel(const int32_t* __restrict__ xy, long long rx, long long ry, int n,
                                                          int pw, int ph, int ds, int level, long long width,
                                                          long long height, uint32_t seed,
                                                          const long long* __restrict__ ell, int k,
                                                          uint8_t* __restrict__ dst) {
    extern __shared__ long long sell[];
    for (int i = threadIdx.x; i < k * 4; i += 256) sell[i] = ell[i];
    __syncthreads();
    const size_t per_tile = (size_t)pw * ph;
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)n * per_tile;
    if (idx >= total) return;
    const int tile = (int)(idx / per_tile);
    const int rem = (int)(idx - (size_t)tile * per_tile);
    const int py = rem / pw, px = rem - py * pw;
    const long long gx = (xy ? (long long)xy[2 * tile] : rx) + (long long)px * ds;
    const long long gy = (xy ? (long long)xy[2 * tile + 1] : ry) + (long long)py * ds;
    const long long ux = gx >> 4, uy = gy >> 4;
    bool tissue = false;
    for (int e = 0; e < k; ++e) {
        const long long cx = sell[4 * e], cy = sell[4 * e + 1], a = sell[4 * e + 2], b = sell[4 * e + 3];
        const long long dx = (ux - cx) * b, dy = (uy - cy) * a, ab = a * b;
        tissue |= (dx * dx + dy * dy) <= ab * ab;
    }
    const uint32_t key = (uint32_t)gx * 0x9E3779B1u + (uint32_t)gy * 0x85EBCA77u + seed +
                         (uint32_t)level * 0xC2B2AE3Du;
    const uint32_t h = mix32(key);
    const int n0 = h & 0xFF, n1 = (h >> 8) & 0xFF, n2 = (h >> 16) & 0xFF;
    const int bg = 236 + (n0 & 7);
    int r = tissue ? 168 + (n0 >> 2) : bg;
    int g = tissue ? 72 + (n1 >> 1) : bg;
    int b = tissue ? 136 + (n2 >> 2) : bg;
    if (gx < 0 || gy < 0 || gx >= width || gy >= height) { r = g = b = 0; }
    uint8_t* o = dst + idx * 3;
    o[0] = (uint8_t)r; o[1] = (uint8_t)g; o[2] = (uint8_t)b;
}

}  // namespace
}  // namespace ap

extern "C" int ap_synth_tiles(const int32_t* xy, int n, int ps, int level_ds, int level, int64_t width,
                              int64_t height, uint32_t seed, const int64_t* ellipses, int k, uint8_t* dst,
                              ap_stream_t stream) {
    AP_REQUIRE(xy && dst && (ellipses || k == 0), "synth_tiles: null pointer");
    AP_REQUIRE(n >= 0 && ps > 0 && level_ds > 0 && k >= 0 && k <= 256, "synth_tiles: bad arguments");
    if (n == 0) return AP_OK;
    const size_t total = (size_t)n * ps * ps;
    const size_t blocks = (total + 255) / 256;
    AP_REQUIRE(blocks < (1ull << 31), "synth_tiles: too many pixels in one call");
    ap::synth_tiles_kernel<<<(unsigned)blocks, 256, (size_t)k * 4 * sizeof(long long), (hipStream_t)stream>>>(
        xy, 0, 0, n, ps, ps, level_ds, level, (long long)width, (long long)height, seed, (const long long*)ellipses, k, dst);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

extern "C" int ap_synth_region(int64_t x, int64_t y, int w, int h, int level_ds, int level, int64_t width, int64_t height,
                               uint32_t seed, const int64_t* ellipses, int k, uint8_t* dst, ap_stream_t stream) {
    AP_REQUIRE(dst && (ellipses || k == 0), "synth_region: null pointer");
    AP_REQUIRE(w > 0 && h > 0 && level_ds > 0 && k >= 0 && k <= 256, "synth_region: bad arguments");
    const size_t total = (size_t)w * h;
    AP_REQUIRE(total < (1ull << 31), "synth_region: region of %d x %d pixels is too large for one call", w, h);
    const size_t blocks = (total + 255) / 256;
    ap::synth_tiles_kernel<<<(unsigned)blocks, 256, (size_t)k * 4 * sizeof(long long), (hipStream_t)stream>>>(
        nullptr, (long long)x, (long long)y, 1, w, h, level_ds, level, (long long)width, (long long)height, seed,
        (const long long*)ellipses, k, dst);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

// ---- measurement aid: shader-clock stamps -------------------------------------------------------------------------------
// 32 one-wave workgroups (the dispatcher deals consecutive workgroups round-robin over the 8 XCDs); lane 0 of each writes
// {s_memtime, s_memrealtime} into the slot of the XCD it runs on (HW_REG_XCC_ID).  s_memtime ticks with the shader clock,
// s_memrealtime with the fixed 100 MHz reference: two probes on one stream around a timed region give the AVERAGE shader
// clock the governor granted there, per XCD, without a profiler:  GHz = 0.1 * d(memtime) / d(memrealtime).
namespace ap {
namespace {
__global__ __launch_bounds__(64) void clock_probe_kernel(long long* __restrict__ out) {
    if (threadIdx.x != 0) return;
    const unsigned xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 7u;       // HW_REG_XCC_ID[3:0]
    const unsigned hw = __builtin_amdgcn_s_getreg(((16 - 1) << 11) | (0 << 6) | 4);             // HW_REG_HW_ID[15:0]
    const unsigned cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
    const unsigned slot = (xcc << 8) | (se << 5) | (sh << 4) | cu;
    typedef long long ll2 __attribute__((ext_vector_type(2)));
    ll2 v;
    v[0] = (long long)__builtin_amdgcn_s_memtime();
    v[1] = (long long)__builtin_amdgcn_s_memrealtime();
    *(ll2*)(out + 2 * slot) = v;                                                                 // one 16-byte store per stamp
}
}  // namespace
}  // namespace ap

extern "C" int ap_clock_probe(long long* out, ap_stream_t stream) {
    AP_REQUIRE(out, "ap_clock_probe: null pointer");
    ap::clock_probe_kernel<<<1024, 64, 0, (hipStream_t)stream>>>(out);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}
