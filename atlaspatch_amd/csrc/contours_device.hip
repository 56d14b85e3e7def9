// Border following on the device: cv2.findContours(mask_u8, RETR_CCOMP, CHAIN_APPROX_NONE) of
// /root/reference/atlas_patch/utils/contours.py:59 as HIP kernels (the host form is contours.cpp; both give the same
// ContourSet, tests).
//
// Suzuki & Abe's raster scan is sequential only through its MARKS (a followed border is marked so that it is not followed
// twice, and LNBD carries the parent).  Both uses can be restated on connected components:
//   * every 8-connected foreground component has exactly one outer border, and the scan starts it at the component's first
//     pixel in raster order (its west neighbour is background of the region the component lies in);
//   * every 4-connected background component that does not reach the image frame is a hole of exactly one foreground
//     component; the scan starts its border at scan position (r, c) = the hole's first pixel in raster order, from the
//     foreground pixel (r, c - 1).  (An east edge foreground -> background is examined as a zero pixel only by the trace of the
//     border between those two regions: the zero neighbours a trace examines between two consecutive border pixels form a
//     4-connected run, so they all belong to one background region.  Every other east edge of that border is therefore marked
//     negative by the time the scan reaches it, and the first one never is.)
//   * RETR_CCOMP: a hole's parent is the outer border of the component of (r, c - 1).
// The followed path itself only reads zero / non-zero, never the marks.  So:
//   K1  runs:     per row, label of a pixel = node id of the first pixel of its horizontal run (block max-scan), node 0 = frame
//   K2  merge:    one union per pair of touching runs in adjacent rows (foreground: 8-connected, background: 4-connected;
//                 background runs at the image border join node 0) -- lock-free union-find, links point to the smaller id, so
//                 a component's root is its first pixel in raster order
//   K3  flatten + event flags (component roots = border starts) ; K4 exclusive scan -> discovery index per start
//   K5  pack:     the mask as a zero-framed BIT image (1024 x 1024 -> 139 KB)
//   K6  trace:    one thread per border on the bit image held in LDS: length, exact shoelace sum, parent.  A mask has
//                 thousands of short borders (speckle, pinholes) and a handful of long ones; a walk is a pointer chase of one
//                 lane (~0.4 us per step on the device against ~5 ns on a host core), so a walk is cut at kLongCap steps
//   host          the few LONG borders are walked on the host, on the same bit image (139 KB come back instead of the 1 MB byte
//                 mask); RETR_CCOMP order + area / hole filters on the summaries (select_contours, shared with the host
//                 form); the kept borders' points (tens of borders) are written by the same host walk
// Measured per 1024 x 1024 mask (tools/contours_ab.py): the raster scan, labelling and the ~10^4 short walks -- what the host
// form spends its time on -- run on the device; the device never waits for a single long walk.
// Integer work, bit-exact by construction (contourArea's float64 sum is an exact integer here).
#include <algorithm>
#include <vector>
#include "ap_common.h"
#include "coords_internal.h"
#include "coords_arena.h"

namespace ap {
namespace {

constexpr int kFrame = 0;                 // node id of the zero frame around the image; pixel i is node i + 1

__device__ __forceinline__ int uf_find(int* __restrict__ L, int a) {
    int p = L[a];
    while (p != a) {
        const int g = L[p];
        if (g != p) L[a] = g;             // path halving (benign race: only ever shortens a path towards the same root)
        a = p; p = g;
    }
    return a;
}
__device__ __forceinline__ void uf_union(int* __restrict__ L, int a, int b) {
    for (;;) {
        a = uf_find(L, a); b = uf_find(L, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }          // a < b: link the larger root to the smaller
        const int old = atomicMin(&L[b], a);
        if (old == b) return;
        b = old;                                                  // somebody linked b first: continue from where it points
    }
}

// K1: one workgroup per row; L[node] = node of the first pixel of the horizontal run (same value) the pixel belongs to
__global__ __launch_bounds__(256) void run_label_kernel(const uint8_t* __restrict__ bin, int h, int w, int* __restrict__ L) {
    __shared__ int carry, part[256];
    const int y = blockIdx.x, tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0) L[kFrame] = kFrame;
    if (tid == 0) carry = 0;
    __syncthreads();
    const uint8_t* row = bin + (size_t)y * w;
    for (int x0 = 0; x0 < w; x0 += 256 * 4) {
        int st[4], best = -1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int x = x0 + tid * 4 + e;
            st[e] = -1;
            if (x < w && (x == 0 || (row[x] != 0) != (row[x - 1] != 0))) st[e] = x;
            best = st[e] > best ? st[e] : best;
        }
        part[tid] = best;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {                       // inclusive max-scan of the per-thread maxima
            const int v = tid >= d ? part[tid - d] : -1;
            __syncthreads();
            part[tid] = v > part[tid] ? v : part[tid];
            __syncthreads();
        }
        int run = tid ? part[tid - 1] : -1;
        run = run > carry ? run : carry;                          // carry = last run start of the previous chunk
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int x = x0 + tid * 4 + e;
            run = st[e] > run ? st[e] : run;
            if (x < w) L[(size_t)y * w + x + 1] = y * w + run + 1;
        }
        __syncthreads();
        if (tid == 255) carry = part[255] > carry ? part[255] : carry;
        __syncthreads();
    }
}

// K2: unions between runs of adjacent rows (once per touching pair) and of border background runs with the frame
__global__ __launch_bounds__(256) void run_merge_kernel(const uint8_t* __restrict__ bin, int h, int w, int* __restrict__ L) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)h * w) return;
    const int y = (int)(i / w), x = (int)(i - (size_t)y * w);
    const bool fg = bin[i] != 0;
    const int me = (int)i + 1;
    const bool start = x == 0 || (bin[i - 1] != 0) != fg;
    const bool end = x == w - 1 || (bin[i + 1] != 0) != fg;
    if (!fg && (((y == 0 || y == h - 1) && start) || x == 0 || x == w - 1)) uf_union(L, me, kFrame);
    if (y == 0) return;
    const size_t u = i - w;
    const bool ufg = bin[u] != 0;
    if (ufg == fg) {
        // vertical contact: one union per pair of runs, at the leftmost column they share
        const bool ustart = x == 0 || (bin[u - 1] != 0) != ufg;
        if (start || ustart) uf_union(L, me, (int)u + 1);
    } else if (fg) {
        // diagonal-only contacts of the 8-connected foreground (the pixel above is background): with the run that ends at
        // (x - 1, y - 1) when this run starts here, with the run that starts at (x + 1, y - 1) when this run ends here --
        // otherwise this run continues under that one and the vertical rule has joined them
        if (start && x > 0 && bin[u - 1] != 0) uf_union(L, me, (int)u);
        if (end && x < w - 1 && bin[u + 1] != 0) uf_union(L, me, (int)u + 2);
    }
}

// K3: flatten; flags[i] = 1 when scan position i starts a border (root of a foreground component, or of a background
// component other than the frame's)
__global__ __launch_bounds__(256) void flatten_flags_kernel(int h, int w, int* __restrict__ L, int* __restrict__ flags) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)h * w) return;
    // READ-ONLY walk to the root: with path halving, another thread's late "L[n] = grandparent" could overwrite the root this
    // kernel has just stored in L[n], and the trace reads L[] as final (seen once in 10^4 borders: a hole with the wrong parent)
    int r = (int)i + 1;
    for (int p = L[r]; p != r; p = L[r]) r = p;
    L[i + 1] = r;
    flags[i] = r == (int)i + 1 ? 1 : 0;
}

// K4: exclusive scan of flags (int32) in three launches: 4096 elements per workgroup
constexpr int kScanPer = 16;
__global__ __launch_bounds__(256) void scan_block_kernel(const int* __restrict__ in, size_t n, int* __restrict__ out,
                                                         int* __restrict__ sums) {
    __shared__ int part[256];
    const int tid = threadIdx.x;
    const size_t base = ((size_t)blockIdx.x * 256 + tid) * kScanPer;
    int v[kScanPer], s = 0;
#pragma unroll
    for (int e = 0; e < kScanPer; ++e) { v[e] = base + e < n ? in[base + e] : 0; s += v[e]; }
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const int t = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += t;
        __syncthreads();
    }
    int run = tid ? part[tid - 1] : 0;
#pragma unroll
    for (int e = 0; e < kScanPer; ++e) { if (base + e < n) out[base + e] = run; run += v[e]; }
    if (tid == 255) sums[blockIdx.x] = part[255];
}
__global__ __launch_bounds__(1024) void scan_sums_kernel(int* __restrict__ sums, int nb, int* __restrict__ total) {
    __shared__ int part[1024];
    int carry = 0;
    for (int b0 = 0; b0 < nb; b0 += 1024) {
        const int tid = threadIdx.x, i = b0 + tid;
        const int mine = i < nb ? sums[i] : 0;
        part[tid] = mine;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const int t = tid >= d ? part[tid - d] : 0;
            __syncthreads();
            part[tid] += t;
            __syncthreads();
        }
        if (i < nb) sums[i] = carry + part[tid] - mine;           // exclusive
        __syncthreads();
        carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ __launch_bounds__(256) void scan_add_kernel(int* __restrict__ out, size_t n, const int* __restrict__ sums) {
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * kScanPer;
    const int add = sums[blockIdx.x];
#pragma unroll
    for (int e = 0; e < kScanPer; ++e) if (base + e < n) out[base + e] += add;
}

// starts[d] = scan position of the border with discovery index d
__global__ __launch_bounds__(256) void collect_starts_kernel(const int* __restrict__ flags, const int* __restrict__ disc, size_t n,
                                                             int* __restrict__ starts) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && flags[i]) starts[disc[i]] = (int)i;
}

// ---- the trace.  Bit image in LDS, one zero ROW above and below and one zero WORD left and right of every row: row stride
//      `ws` = ceil(w / 32) + 2 words, pixel (x, y) is bit x & 31 of word (y + 1) * ws + 1 + (x >> 5).  Padded coordinates
//      px = x + 32, py = y + 1, so that bit_at needs no bounds checks one pixel beyond the image.
struct DevBorder { int origin, is_hole, n, parent; long long area2; };   // = BorderSummary

// The 8 neighbours of padded pixel (px, py) as a ring mask: bit s = the neighbour in direction s (0 = E, 1 = NE, 2 = N,
// 3 = NW, 4 = W, 5 = SW, 6 = S, 7 = SE; y grows downwards).  Three independent two-word LDS reads (one round trip), then
// register work only: the walk is one lane following pointers, so the LDS latency per step is what it costs.
__host__ __device__ __forceinline__ uint32_t ring_at(const uint32_t* __restrict__ img, int ws, int px, int py) {
    const int col = (px - 1) >> 5, sh = (px - 1) & 31;
    const uint32_t* r = img + (py - 1) * ws + col;
    const unsigned long long up = ((unsigned long long)r[1] << 32) | r[0];
    const unsigned long long mid = ((unsigned long long)r[ws + 1] << 32) | r[ws];
    const unsigned long long dn = ((unsigned long long)r[2 * ws + 1] << 32) | r[2 * ws];
    const uint32_t u = (uint32_t)(up >> sh) & 7u, m = (uint32_t)(mid >> sh) & 7u, d = (uint32_t)(dn >> sh) & 7u;   // bit 0 = x - 1
    return ((m >> 2) & 1u) | (((u >> 2) & 1u) << 1) | (((u >> 1) & 1u) << 2) | ((u & 1u) << 3) | ((m & 1u) << 4) |
           ((d & 1u) << 5) | (((d >> 1) & 1u) << 6) | (((d >> 2) & 1u) << 7);
}

// EMIT = false: count + shoelace;  EMIT = true: write the points (unpadded x, y) to out.  max_steps bounds the walk: a walk
// that does not close within it reports n = -1 (the device cuts long borders this way; the host's bound is four visits per
// pixel, which a well-formed walk cannot exceed).
// Same walk as contours.cpp::follow (which reads the marked int32 image): the path only depends on zero / non-zero.
__host__ __device__ __forceinline__ int first_set(uint32_t v) {          // 1-based index of the lowest set bit (v != 0)
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffs((int)v);
#else
    return __builtin_ffs((int)v);
#endif
}
template <bool EMIT>
__host__ __device__ __forceinline__ void follow_border(const uint32_t* __restrict__ img, int ws, int ox, int oy, bool is_hole, int max_steps,
                                                       int& n_out, long long& area2, int32_t* __restrict__ out) {
    // offsets + 1 of direction s packed two bits each (dx + 1: 2 2 1 0 0 0 1 2 -> 0x901A, dy + 1: 1 0 0 0 1 2 2 2 -> 0xA901)
    auto dxs = [](int s) { return (int)((0x901Au >> (2 * s)) & 3u) - 1; };
    auto dys = [](int s) { return (int)((0xA901u >> (2 * s)) & 3u) - 1; };
    const int x0 = ox + 32, y0 = oy + 1;
    uint32_t ring = ring_at(img, ws, x0, y0);
    // first neighbour clockwise from the side the scan came from: s_end - 1, s_end - 2, ... (7 candidates)
    const int s_end = is_hole ? 0 : 4;
    int s = s_end, found = 0;
    for (int k = 1; k < 8; ++k) {
        s = (s_end - k) & 7;
        if ((ring >> s) & 1u) { found = 1; break; }
    }
    if (!found) {                                                 // isolated pixel: one point, zero area
        if (EMIT) { out[0] = ox; out[1] = oy; }
        n_out = 1; area2 = 0;
        return;
    }
    const int x1 = x0 + dxs(s), y1 = y0 + dys(s);
    int n = 0;
    long long a2 = 0;
    int x3 = x0, y3 = y0, fx = 0, fy = 0, lx = 0, ly = 0;         // first / last emitted point (unpadded)
    for (;;) {
        // next neighbour counter-clockwise after direction s: s + 1, s + 2, ... s + 8
        const uint32_t rot = ((ring | (ring << 8)) >> ((s + 1) & 7)) & 0xFFu;
        if (rot == 0) { n_out = -1; area2 = 0; return; }          // cannot happen: the pixel we came from is set
        s = (s + first_set(rot)) & 7;
        const int x4 = x3 + dxs(s), y4 = y3 + dys(s);
        const int ex = x3 - 32, ey = y3 - 1;
        if (EMIT) { out[2 * n] = ex; out[2 * n + 1] = ey; }
        if (n == 0) { fx = ex; fy = ey; } else a2 += (long long)lx * ey - (long long)ly * ex;
        lx = ex; ly = ey;
        ++n;
        if (x4 == x0 && y4 == y0 && x3 == x1 && y3 == y1) break;
        if (n >= max_steps) { n_out = -1; area2 = 0; return; }
        x3 = x4; y3 = y4;
        ring = ring_at(img, ws, x3, y3);
        s = (s + 4) & 7;
    }
    a2 += (long long)lx * fy - (long long)ly * fx;               // closing term (the shoelace sum starts from the last point)
    n_out = n; area2 = a2;
}

constexpr int kLongCap = 768;             // steps after which the device hands a border to the host

// K5: zero-framed bit image in global memory (the frame words are zeroed by a memset before the launch)
__global__ __launch_bounds__(256) void pack_bits_kernel(const uint8_t* __restrict__ bin, int h, int w, int ws, uint32_t* __restrict__ img) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    if ((w & 15) == 0 && (((uintptr_t)bin) & 15) == 0) {
        // 16 pixels per lane and load (the threshold kernel writes 0 / 255: bit 0 of a byte = the pixel): 1024 pixels per wave
        // step, two lanes make one word
        const int chunks = (w + 1023) / 1024;
        for (int t = wave; t < h * chunks; t += nwaves) {
            const int y = t / chunks, x = (t - y * chunks) * 1024 + lane * 16;
            uint32_t bits = 0;
            if (x < w) {
                const u32x4 v = *(const u32x4*)(bin + (size_t)y * w + x);
#pragma unroll
                for (int q = 0; q < 4; ++q) bits |= ((((v[q] & 0x01010101u) * 0x00204081u) >> 21) & 0xFu) << (4 * q);
            }
            const uint32_t hi = (uint32_t)__shfl_down((int)bits, 1, 64);
            if ((lane & 1) == 0 && x < w) img[(y + 1) * ws + 1 + (x >> 5)] = bits | (hi << 16);
        }
    } else {
        const int groups = (w + 63) / 64;
        for (int g = wave; g < h * groups; g += nwaves) {        // any width: 64 pixels per wave step, two words by ballot
            const int y = g / groups, x = (g - y * groups) * 64 + lane;
            const bool on = x < w && bin[(size_t)y * w + x] != 0;
            const unsigned long long m = __ballot(on);
            if (lane == 0) {
                uint32_t* dst = img + (y + 1) * ws + 1 + ((g - y * groups) << 1);
                dst[0] = (uint32_t)m;
                if ((g - y * groups) * 64 + 32 < w) dst[1] = (uint32_t)(m >> 32);
            }
        }
    }
}

// K6: every workgroup copies the bit image into LDS, then its threads walk one border each: summaries (n = -1: longer than
// kLongCap, left to the host)
__global__ __launch_bounds__(1024) void trace_kernel(const uint8_t* __restrict__ bin, const uint32_t* __restrict__ packed, int h, int w,
                                                     int ws, const int* __restrict__ L, const int* __restrict__ disc,
                                                     const int* __restrict__ starts, int nb, DevBorder* __restrict__ summary) {
    extern __shared__ uint32_t img[];
    const int words = (h + 2) * ws;
    for (int i = threadIdx.x; i < words; i += blockDim.x) img[i] = packed[i];
    __syncthreads();
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nb; k += gridDim.x * blockDim.x) {
        const int pos = starts[k];                                // scan position of the start event
        const bool is_hole = bin[pos] == 0;
        const int origin = is_hole ? pos - 1 : pos;
        const int oy = origin / w, ox = origin - oy * w;
        int n; long long a2;
        follow_border<false>(img, ws, ox, oy, is_hole, kLongCap, n, a2, nullptr);
        DevBorder b;
        b.origin = origin; b.is_hole = is_hole ? 1 : 0; b.n = n; b.area2 = a2;
        b.parent = is_hole ? disc[L[origin + 1] - 1] : -1;        // outer border of the foreground component of (r, c - 1)
        summary[k] = b;
    }
}

}  // namespace

bool contours_device_supported(int h, int w) {
    const int ws = (w + 31) / 32 + 2;
    return (size_t)(h + 2) * ws * 4 <= 150 * 1024 && (size_t)h * w < (1u << 30);
}

// dbin: device uint8 [h * w] (non-zero = tissue).  Same result as contours_from_binary on the same image.
int contours_from_binary_device(const uint8_t* dbin, int h, int w, double tissue_area_thresh, int min_hole_area, int max_n_holes,
                                double sx, double sy, ContourSet& out, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    AP_REQUIRE(contours_device_supported(h, w), "contours (device): %d x %d does not fit the LDS bit image", h, w);
    static_assert(sizeof(DevBorder) == sizeof(BorderSummary), "summary layouts differ");
    const size_t npx = (size_t)h * w;
    const int ws = (w + 31) / 32 + 2;
    const size_t words = (size_t)(h + 2) * ws, lds = words * 4;
    static bool configured = false;
    if (!configured) {
        AP_HIP_CHECK(hipFuncSetAttribute((const void*)trace_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        configured = true;
    }
    DevBuf<int> L, flags, disc, sums, total_d, starts;
    DevBuf<uint32_t> packed;
    int rc;
    const unsigned pxb = (unsigned)((npx + 255) / 256);
    const unsigned scb = (unsigned)((npx + 256 * kScanPer - 1) / (256 * kScanPer));
    if ((rc = L.alloc(npx + 1)) || (rc = flags.alloc(npx)) || (rc = disc.alloc(npx)) || (rc = sums.alloc(scb)) || (rc = total_d.alloc(1)) ||
        (rc = packed.alloc(words)))
        return rc;
    AP_HIP_CHECK(hipMemsetAsync(packed.p, 0, lds, s));
    pack_bits_kernel<<<64, 256, 0, s>>>(dbin, h, w, ws, packed.p);
    run_label_kernel<<<h, 256, 0, s>>>(dbin, h, w, L.p);
    run_merge_kernel<<<pxb, 256, 0, s>>>(dbin, h, w, L.p);
    flatten_flags_kernel<<<pxb, 256, 0, s>>>(h, w, L.p, flags.p);
    scan_block_kernel<<<scb, 256, 0, s>>>(flags.p, npx, disc.p, sums.p);
    scan_sums_kernel<<<1, 1024, 0, s>>>(sums.p, (int)scb, total_d.p);
    scan_add_kernel<<<scb, 256, 0, s>>>(disc.p, npx, sums.p);
    AP_HIP_CHECK(hipGetLastError());
    int nb = 0;
    std::vector<uint32_t> img(words);                      // the host walks the long borders on the same bit image
    AP_HIP_CHECK(hipMemcpyAsync(&nb, total_d.p, sizeof(int), hipMemcpyDeviceToHost, s));
    AP_HIP_CHECK(hipMemcpyAsync(img.data(), packed.p, lds, hipMemcpyDeviceToHost, s));
    AP_HIP_CHECK(hipStreamSynchronize(s));
    out.polys.clear();
    out.tissue.clear();
    if (nb == 0) return AP_OK;
    DevBuf<DevBorder> dsum;
    if ((rc = starts.alloc(nb)) || (rc = dsum.alloc(nb))) return rc;
    collect_starts_kernel<<<pxb, 256, 0, s>>>(flags.p, disc.p, npx, starts.p);
    const int wgs = nb >= 64 * 1024 ? 64 : (nb + 1023) / 1024;
    trace_kernel<<<wgs, 1024, lds, s>>>(dbin, packed.p, h, w, ws, L.p, disc.p, starts.p, nb, dsum.p);
    AP_HIP_CHECK(hipGetLastError());
    std::vector<BorderSummary> found(nb);
    AP_HIP_CHECK(hipMemcpyAsync(found.data(), dsum.p, (size_t)nb * sizeof(BorderSummary), hipMemcpyDeviceToHost, s));
    AP_HIP_CHECK(hipStreamSynchronize(s));
    const int host_cap = 4 * h * w + 16;
    for (BorderSummary& b : found) {
        if (b.n > 0) continue;                             // closed on the device
        follow_border<false>(img.data(), ws, b.origin % w, b.origin / w, b.is_hole != 0, host_cap, b.n, b.area2, nullptr);
        AP_REQUIRE(b.n > 0, "contours (device): a border walk did not close (internal error)");
    }
    Selection sel;
    select_contours(found, h, w, tissue_area_thresh, min_hole_area, max_n_holes, sel);
    build_contour_set(sel, sx, sy, [&](int d) {
        const BorderSummary& b = found[d];
        std::vector<int32_t> xy((size_t)b.n * 2);
        int n; long long a2;
        follow_border<true>(img.data(), ws, b.origin % w, b.origin / w, b.is_hole != 0, b.n, n, a2, xy.data());
        return xy;
    }, out);
    return AP_OK;
}

}  // namespace ap
