// Tissue mask -> patch coordinates on the device.
//
//   threshold_kernel   C1: (mask > 0.5) -> u8, one 16-byte load per lane (HBM-bound, 5 B/px)
//   grid_rows_kernel   C5: one lane per grid cell, one workgroup per (tissue contour, grid row, 256 cells).  Each lane runs
//                      OpenCV's integer point-in-polygon test for the cell centre against every hole (strictly inside ->
//                      reject) and for the four diagonal probes against the tissue polygon (inside or on the edge -> keep),
//                      all four probes in one pass.  Only the edges that can matter for the row are visited: an edge
//                      changes the state of a probe (a crossing, or the on-vertex / on-horizontal-edge case) only if
//                      min(y0, y1) <= probe y <= max(y0, y1), so the host buckets every polygon's edges by the grid rows
//                      whose probe band [cy - shift, cy + shift] their y-range meets (CSR: row pointers + explicit
//                      endpoint pairs), and the workgroup streams its row's bucket through LDS (every lane reads the same
//                      edge -> LDS broadcast).  Crossing parity and "on the boundary" are order-independent, so the result
//                      equals the full scan's (grid_flags_kernel, kept for A/B: AP_GRID_FLAGS_LEGACY=1) bit for bit; the
//                      work drops from cells x vertices to cells x (edges meeting the row): 1.5 ms -> ~0.1 ms per
//                      100 000^2 slide with a SAM2 mask.
//   block counts + scan + compact: kept cells are ranked with wave ballots / popcounts so the
//                      rows come out in the reference's order (contour-major, row-major grid).
//
// Integer work, bit-exact by construction: int32 coordinates, int64 cross products
// (extraction.py:67-103, contours.py:22-38; cv::pointPolygonTest integer branch [3P]).
// Algorithmic bytes: 8 B per polygon vertex per 256-cell block (LDS-served after the first
// touch) + 20 B per emitted row; latency/ALU-bound, reported as cells/s.
#include <cstdlib>
#include <vector>
#include "ap_common.h"
#include "coords_internal.h"
#include "coords_arena.h"

struct ap_contours {
    ap::ContourSet set;
};

namespace ap {
namespace {

__global__ void threshold_kernel(const float* __restrict__ mask, uint8_t* __restrict__ out, size_t count) {
    const size_t i4 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 + 3 < count) {
        const f32x4 v = *(const f32x4*)(mask + i4);
        uint32_t packed = (v[0] > 0.5f ? 255u : 0u) | (v[1] > 0.5f ? 255u << 8 : 0u) |
                          (v[2] > 0.5f ? 255u << 16 : 0u) | (v[3] > 0.5f ? 255u << 24 : 0u);
        *(uint32_t*)(out + i4) = packed;
    } else {
        for (size_t i = i4; i < count; ++i) out[i] = mask[i] > 0.5f ? 255 : 0;
    }
}

struct TissueDesc {          // one per tissue contour
    int poly_off, poly_cnt;  // vertex range in the packed vertex array
    int hole_first, hole_cnt;   // range in the hole table
    int x0, y0, nx, ny;      // grid origin and size
    long cell_off;           // first cell index (global)
};
struct HoleDesc { int poly_off, poly_cnt; };
struct BlockDesc { int tissue; int cell0; };   // 256 cells of one tissue contour

constexpr int kChunk = 1024;   // vertices per LDS chunk

struct PipState {
    int cnt; int on;
};

__device__ __forceinline__ void pip_edge(int px, int py, int v0x, int v0y, int vx, int vy, PipState& s) {
    const bool skip = (v0y <= py && vy <= py) || (v0y > py && vy > py) || (v0x < px && vx < px);
    if (skip) {
        if (py == vy && (px == vx || (py == v0y && ((v0x <= px && px <= vx) || (vx <= px && px <= v0x)))))
            s.on = 1;
        return;
    }
    long long dist = (long long)(py - v0y) * (vx - v0x) - (long long)(px - v0x) * (vy - v0y);
    if (dist == 0) { s.on = 1; return; }
    if (vy < v0y) dist = -dist;
    s.cnt += dist > 0;
}

// result of cv::pointPolygonTest(measureDist=false): +1 inside, 0 on edge, -1 outside
__device__ __forceinline__ int pip_result(const PipState& s) { return s.on ? 0 : ((s.cnt & 1) ? 1 : -1); }

__global__ __launch_bounds__(256) void grid_flags_kernel(const int2* __restrict__ verts,
                                                         const TissueDesc* __restrict__ tissues,
                                                         const HoleDesc* __restrict__ holes,
                                                         const BlockDesc* __restrict__ blocks,
                                                         int patch, int step, uint8_t* __restrict__ flags) {
    __shared__ int2 sv[kChunk + 1];
    const BlockDesc bd = blocks[blockIdx.x];
    const TissueDesc td = tissues[bd.tissue];
    const int ncell = td.nx * td.ny;
    const int cell = bd.cell0 + threadIdx.x;
    const bool live = cell < ncell;
    const int iy = live ? cell / td.nx : 0, ix = live ? cell - iy * td.nx : 0;
    const int x = td.x0 + ix * step, y = td.y0 + iy * step;
    const int half = patch / 2;
    const int cx = x + half, cy = y + half;
    const int shift = half / 2;                       // int(patch // 2 * 0.5)

    auto scan = [&](int off, int cnt, auto&& per_edge) {
        // edges (v[i-1] -> v[i]) for i in [0, cnt), v[-1] = v[cnt-1]
        for (int base = 0; base < cnt; base += kChunk) {
            const int m = min(kChunk, cnt - base);
            __syncthreads();
            for (int i = threadIdx.x; i <= m; i += 256) {
                int src = base + i - 1;               // sv[0] = vertex before the chunk
                if (src < 0) src = cnt - 1;
                sv[i] = verts[off + src];
            }
            __syncthreads();
            for (int i = 0; i < m; ++i) per_edge(sv[i], sv[i + 1]);
        }
    };

    bool in_hole = false;
    for (int hI = 0; hI < td.hole_cnt; ++hI) {
        const HoleDesc hd = holes[td.hole_first + hI];
        PipState s{0, 0};
        scan(hd.poly_off, hd.poly_cnt, [&](int2 a, int2 b) { pip_edge(cx, cy, a.x, a.y, b.x, b.y, s); });
        if (pip_result(s) > 0) in_hole = true;
    }
    PipState p0{0, 0}, p1{0, 0}, p2{0, 0}, p3{0, 0};
    scan(td.poly_off, td.poly_cnt, [&](int2 a, int2 b) {
        pip_edge(cx - shift, cy - shift, a.x, a.y, b.x, b.y, p0);
        if (shift > 0) {
            pip_edge(cx + shift, cy + shift, a.x, a.y, b.x, b.y, p1);
            pip_edge(cx + shift, cy - shift, a.x, a.y, b.x, b.y, p2);
            pip_edge(cx - shift, cy + shift, a.x, a.y, b.x, b.y, p3);
        }
    });
    bool keep = pip_result(p0) >= 0;
    if (shift > 0) keep = keep || pip_result(p1) >= 0 || pip_result(p2) >= 0 || pip_result(p3) >= 0;
    if (live) flags[td.cell_off + cell] = (keep && !in_hole) ? 1 : 0;
}

// ---- row-bucketed variant (the product path)
struct RowBlock { int tissue; int iy; int ix0; int cnt; };   // cnt cells of grid row iy starting at column ix0
struct PolyRows { int ptr_off; };                             // row pointers of a polygon: rowptr[ptr_off + iy .. + 1]

__global__ __launch_bounds__(256) void grid_rows_kernel(const int4* __restrict__ edges, const int* __restrict__ rowptr,
                                                        const TissueDesc* __restrict__ tissues,
                                                        const HoleDesc* __restrict__ holes,
                                                        const PolyRows* __restrict__ tissue_rows,
                                                        const PolyRows* __restrict__ hole_rows,
                                                        const RowBlock* __restrict__ blocks,
                                                        int patch, int step, uint8_t* __restrict__ flags) {
    __shared__ int4 se[kChunk];
    const RowBlock bd = blocks[blockIdx.x];
    const TissueDesc td = tissues[bd.tissue];
    const bool live = (int)threadIdx.x < bd.cnt;
    const int ix = bd.ix0 + (live ? (int)threadIdx.x : 0);
    const int x = td.x0 + ix * step, y = td.y0 + bd.iy * step;
    const int half = patch / 2;
    const int cx = x + half, cy = y + half;
    const int shift = half / 2;                       // int(patch // 2 * 0.5)

    auto scan = [&](int ptr_off, auto&& per_edge) {
        const int e0 = rowptr[ptr_off + bd.iy], e1 = rowptr[ptr_off + bd.iy + 1];
        for (int base = e0; base < e1; base += kChunk) {
            const int m = min(kChunk, e1 - base);
            __syncthreads();
            for (int i = threadIdx.x; i < m; i += 256) se[i] = edges[base + i];
            __syncthreads();
            for (int i = 0; i < m; ++i) per_edge(se[i]);
        }
    };

    bool in_hole = false;
    for (int hI = 0; hI < td.hole_cnt; ++hI) {
        PipState s{0, 0};
        scan(hole_rows[td.hole_first + hI].ptr_off, [&](int4 e) { pip_edge(cx, cy, e.x, e.y, e.z, e.w, s); });
        if (pip_result(s) > 0) in_hole = true;
    }
    PipState p0{0, 0}, p1{0, 0}, p2{0, 0}, p3{0, 0};
    scan(tissue_rows[bd.tissue].ptr_off, [&](int4 e) {
        pip_edge(cx - shift, cy - shift, e.x, e.y, e.z, e.w, p0);
        if (shift > 0) {
            pip_edge(cx + shift, cy + shift, e.x, e.y, e.z, e.w, p1);
            pip_edge(cx + shift, cy - shift, e.x, e.y, e.z, e.w, p2);
            pip_edge(cx - shift, cy + shift, e.x, e.y, e.z, e.w, p3);
        }
    });
    bool keep = pip_result(p0) >= 0;
    if (shift > 0) keep = keep || pip_result(p1) >= 0 || pip_result(p2) >= 0 || pip_result(p3) >= 0;
    if (live) flags[td.cell_off + (long)bd.iy * td.nx + ix] = (keep && !in_hole) ? 1 : 0;
}

// kept counts / compaction over row blocks (same ballot + popcount ranks as the 256-consecutive-cell blocks below)
__global__ __launch_bounds__(256) void rowblock_count_kernel(const uint8_t* __restrict__ flags,
                                                             const TissueDesc* __restrict__ tissues,
                                                             const RowBlock* __restrict__ blocks,
                                                             unsigned* __restrict__ counts) {
    __shared__ unsigned wsum[4];
    const RowBlock bd = blocks[blockIdx.x];
    const TissueDesc td = tissues[bd.tissue];
    const bool f = (int)threadIdx.x < bd.cnt && flags[td.cell_off + (long)bd.iy * td.nx + bd.ix0 + threadIdx.x];
    const unsigned long long b = __ballot(f);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = (unsigned)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(256) void rowblock_compact_kernel(const uint8_t* __restrict__ flags,
                                                               const TissueDesc* __restrict__ tissues,
                                                               const RowBlock* __restrict__ blocks,
                                                               const unsigned long long* __restrict__ offsets,
                                                               int step, int rw, int rh, int level,
                                                               int32_t* __restrict__ rows, unsigned long long cap) {
    __shared__ unsigned wsum[4];
    const RowBlock bd = blocks[blockIdx.x];
    const TissueDesc td = tissues[bd.tissue];
    const int ix = bd.ix0 + (int)threadIdx.x;
    const bool f = (int)threadIdx.x < bd.cnt && flags[td.cell_off + (long)bd.iy * td.nx + ix];
    const unsigned long long b = __ballot(f);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wsum[wave] = (unsigned)__popcll(b);
    __syncthreads();
    unsigned before = 0;
    for (int wv = 0; wv < wave; ++wv) before += wsum[wv];
    const unsigned rank = before + (unsigned)__popcll(b & ((1ull << lane) - 1ull));
    if (f) {
        const unsigned long long r = offsets[blockIdx.x] + rank;
        if (r < cap) {
            int32_t* o = rows + r * 5;
            o[0] = td.x0 + ix * step; o[1] = td.y0 + bd.iy * step; o[2] = rw; o[3] = rh; o[4] = level;
        }
    }
}

// per-block kept counts (wave ballot + popcount)
__global__ __launch_bounds__(256) void block_count_kernel(const uint8_t* __restrict__ flags,
                                                          const TissueDesc* __restrict__ tissues,
                                                          const BlockDesc* __restrict__ blocks,
                                                          unsigned* __restrict__ counts) {
    __shared__ unsigned wsum[4];
    const BlockDesc bd = blocks[blockIdx.x];
    const TissueDesc td = tissues[bd.tissue];
    const int cell = bd.cell0 + threadIdx.x;
    const bool f = cell < td.nx * td.ny && flags[td.cell_off + cell];
    const unsigned long long b = __ballot(f);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = (unsigned)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// single-workgroup exclusive scan of the block counts (<= a few thousand entries)
__global__ __launch_bounds__(1024) void scan_kernel(const unsigned* __restrict__ counts,
                                                    unsigned long long* __restrict__ offsets, int n,
                                                    unsigned long long* __restrict__ total) {
    __shared__ unsigned long long part[1024];
    const int per = (n + 1023) / 1024;
    const int b = threadIdx.x * per, e = min(n, b + per);
    unsigned long long s = 0;
    for (int i = b; i < e; ++i) s += counts[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        unsigned long long v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned long long run = threadIdx.x ? part[threadIdx.x - 1] : 0;
    for (int i = b; i < e; ++i) { offsets[i] = run; run += counts[i]; }
    if (threadIdx.x == 1023) *total = part[1023];
}

__global__ __launch_bounds__(256) void compact_kernel(const uint8_t* __restrict__ flags,
                                                      const TissueDesc* __restrict__ tissues,
                                                      const BlockDesc* __restrict__ blocks,
                                                      const unsigned long long* __restrict__ offsets,
                                                      int step, int rw, int rh, int level,
                                                      int32_t* __restrict__ rows, unsigned long long cap) {
    __shared__ unsigned wsum[4];
    const BlockDesc bd = blocks[blockIdx.x];
    const TissueDesc td = tissues[bd.tissue];
    const int cell = bd.cell0 + threadIdx.x;
    const bool f = cell < td.nx * td.ny && flags[td.cell_off + cell];
    const unsigned long long b = __ballot(f);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wsum[wave] = (unsigned)__popcll(b);
    __syncthreads();
    unsigned before = 0;
    for (int wv = 0; wv < wave; ++wv) before += wsum[wv];
    const unsigned rank = before + (unsigned)__popcll(b & ((1ull << lane) - 1ull));
    if (f) {
        const unsigned long long r = offsets[blockIdx.x] + rank;
        if (r < cap) {
            const int iy = cell / td.nx, ix = cell - iy * td.nx;
            int32_t* o = rows + r * 5;
            o[0] = td.x0 + ix * step; o[1] = td.y0 + iy * step; o[2] = rw; o[3] = rh; o[4] = level;
        }
    }
}

}  // namespace
}  // namespace ap

extern "C" {

int ap_contours_from_mask(const float* mask, int h, int w, double tissue_area_thresh,
                          int min_hole_area, int max_n_holes, double sx, double sy,
                          ap_contours** out, ap_stream_t stream) {
    AP_REQUIRE(mask && out, "contours_from_mask: null argument");
    AP_REQUIRE(h > 0 && w > 0 && (long)h * w <= (1l << 28), "contours_from_mask: bad mask shape %dx%d", h, w);
    hipStream_t s = (hipStream_t)stream;
    const size_t count = (size_t)h * w;
    ap::ArenaScope arena_scope;
    ap::DevBuf<float> dmask; ap::DevBuf<uint8_t> dbin;
    int rc;
    if ((rc = dmask.alloc(count)) != AP_OK || (rc = dbin.alloc(count)) != AP_OK) return rc;
    AP_HIP_CHECK(hipMemcpyAsync(dmask.p, mask, count * sizeof(float), hipMemcpyHostToDevice, s));
    ap::threshold_kernel<<<(unsigned)((count / 4 + 256) / 256), 256, 0, s>>>(dmask.p, dbin.p, count);
    AP_HIP_CHECK(hipGetLastError());
    ap_contours* c = new ap_contours();
    // border following on the device (contours_device.hip: component labelling + one thread per border on an LDS bit image);
    // AP_CONTOURS_HOST=1, or a mask too large for the LDS image, takes the host form (contours.cpp) -- same ContourSet
    const bool host_form = getenv("AP_CONTOURS_HOST") != nullptr;          // read per call: the tests flip it in-process
    if (!host_form && ap::contours_device_supported(h, w)) {
        rc = ap::contours_from_binary_device(dbin.p, h, w, tissue_area_thresh, min_hole_area, max_n_holes, sx, sy, c->set, (void*)s);
        if (rc != AP_OK) { delete c; return rc; }
        *out = c;
        return AP_OK;
    }
    std::vector<uint8_t> bin(count);
    AP_HIP_CHECK(hipMemcpyAsync(bin.data(), dbin.p, count, hipMemcpyDeviceToHost, s));
    AP_HIP_CHECK(hipStreamSynchronize(s));
    ap::contours_from_binary(bin.data(), h, w, tissue_area_thresh, min_hole_area, max_n_holes, sx, sy, c->set);
    *out = c;
    return AP_OK;
}

void ap_contours_destroy(ap_contours* c) { delete c; }

int ap_contours_count(const ap_contours* c) { return c ? (int)c->set.tissue.size() : 0; }

int ap_contours_num_holes(const ap_contours* c, int i) {
    if (!c || i < 0 || i >= (int)c->set.tissue.size()) return 0;
    return (int)c->set.tissue[i].holes.size();
}

int ap_contours_points(const ap_contours* c, int i, int hole, int scaled, int32_t* xy, int cap) {
    AP_REQUIRE(c && i >= 0 && i < (int)c->set.tissue.size(), "contours_points: bad contour index %d", i);
    const ap::Tissue& t = c->set.tissue[i];
    int poly = t.poly;
    if (hole >= 0) {
        AP_REQUIRE(hole < (int)t.holes.size(), "contours_points: bad hole index %d", hole);
        poly = t.holes[hole];
    }
    const std::vector<int32_t>& v = scaled ? c->set.polys[poly].scaled : c->set.polys[poly].raw;
    const int n = (int)(v.size() / 2);
    if (xy && cap > 0) {
        const int m = n < cap ? n : cap;
        for (int k = 0; k < 2 * m; ++k) xy[k] = v[k];
    }
    return n;
}

int ap_grid_coords(const ap_contours* c, int patch_size_src, int step_src, int read_w, int read_h,
                   int level, int32_t* coords, size_t cap, size_t* n_rows, ap_stream_t stream) {
    AP_REQUIRE(c && n_rows, "grid_coords: null argument");
    AP_REQUIRE(patch_size_src > 0 && step_src > 0, "grid_coords: patch %d / step %d", patch_size_src, step_src);
    AP_REQUIRE(coords || cap == 0, "grid_coords: null output with non-zero capacity");
    hipStream_t s = (hipStream_t)stream;
    *n_rows = 0;
    const ap::ContourSet& set = c->set;
    if (set.tissue.empty()) return AP_OK;

    // ---- pack polygons and describe the grids (bounding rect of the scaled contour)
    std::vector<int2> verts;
    std::vector<ap::TissueDesc> tds;
    std::vector<ap::HoleDesc> hds;
    std::vector<ap::BlockDesc> bds;
    long cells = 0;
    auto pack = [&](int poly, int& off, int& cnt) {
        const std::vector<int32_t>& v = set.polys[poly].scaled;
        off = (int)verts.size(); cnt = (int)(v.size() / 2);
        for (int k = 0; k < cnt; ++k) verts.push_back(make_int2(v[2 * k], v[2 * k + 1]));
    };
    for (size_t ti = 0; ti < set.tissue.size(); ++ti) {
        const ap::Tissue& t = set.tissue[ti];
        ap::TissueDesc td{};
        pack(t.poly, td.poly_off, td.poly_cnt);
        if (td.poly_cnt == 0) continue;
        int xmin = verts[td.poly_off].x, xmax = xmin, ymin = verts[td.poly_off].y, ymax = ymin;
        for (int k = 1; k < td.poly_cnt; ++k) {
            const int2 p = verts[td.poly_off + k];
            xmin = p.x < xmin ? p.x : xmin; xmax = p.x > xmax ? p.x : xmax;
            ymin = p.y < ymin ? p.y : ymin; ymax = p.y > ymax ? p.y : ymax;
        }
        // boundingRect: w = xmax - xmin + 1; range(x0, x0 + w, step)
        td.x0 = xmin; td.y0 = ymin;
        td.nx = (xmax - xmin + 1 + step_src - 1) / step_src;
        td.ny = (ymax - ymin + 1 + step_src - 1) / step_src;
        td.hole_first = (int)hds.size(); td.hole_cnt = (int)t.holes.size();
        for (int hp : t.holes) { ap::HoleDesc hd{}; pack(hp, hd.poly_off, hd.poly_cnt); hds.push_back(hd); }
        td.cell_off = cells;
        const long nc = (long)td.nx * td.ny;
        AP_REQUIRE(nc < (1l << 31), "grid_coords: grid too large");
        for (long c0 = 0; c0 < nc; c0 += 256) bds.push_back({(int)tds.size(), (int)c0});
        cells += nc;
        tds.push_back(td);
    }
    if (bds.empty()) return AP_OK;

    const bool legacy = getenv("AP_GRID_FLAGS_LEGACY") != nullptr;     // the full-scan kernel, for A/B (tools / tests)
    ap::ArenaScope arena_scope;
    ap::DevBuf<int2> dverts; ap::DevBuf<ap::TissueDesc> dtd; ap::DevBuf<ap::HoleDesc> dhd;
    ap::DevBuf<ap::BlockDesc> dbd; ap::DevBuf<uint8_t> dflags; ap::DevBuf<unsigned> dcounts;
    ap::DevBuf<unsigned long long> doffs; ap::DevBuf<unsigned long long> dtotal; ap::DevBuf<int32_t> drows;
    ap::DevBuf<int4> dedges; ap::DevBuf<int> drowptr; ap::DevBuf<ap::PolyRows> dtrows, dhrows; ap::DevBuf<ap::RowBlock> drb;
    int rc;
    unsigned nb = 0;
    if ((rc = dtd.alloc(tds.size())) || (rc = dhd.alloc(hds.size())) || (rc = dflags.alloc((size_t)cells)) || (rc = dtotal.alloc(1)))
        return rc;
    AP_HIP_CHECK(hipMemcpyAsync(dtd.p, tds.data(), tds.size() * sizeof(ap::TissueDesc), hipMemcpyHostToDevice, s));
    if (!hds.empty())
        AP_HIP_CHECK(hipMemcpyAsync(dhd.p, hds.data(), hds.size() * sizeof(ap::HoleDesc), hipMemcpyHostToDevice, s));
    // host vectors of the bucketed path live until the final synchronisation below (asynchronous copies read them)
    std::vector<int4> edges;
    std::vector<int> rowptr;
    std::vector<ap::PolyRows> trows(tds.size()), hrows(hds.size());
    std::vector<ap::RowBlock> rbs;
    if (legacy) {
        nb = (unsigned)bds.size();
        if ((rc = dverts.alloc(verts.size())) || (rc = dbd.alloc(bds.size())) || (rc = dcounts.alloc(nb)) || (rc = doffs.alloc(nb)))
            return rc;
        AP_HIP_CHECK(hipMemcpyAsync(dverts.p, verts.data(), verts.size() * sizeof(int2), hipMemcpyHostToDevice, s));
        AP_HIP_CHECK(hipMemcpyAsync(dbd.p, bds.data(), bds.size() * sizeof(ap::BlockDesc), hipMemcpyHostToDevice, s));
        ap::grid_flags_kernel<<<nb, 256, 0, s>>>(dverts.p, dtd.p, dhd.p, dbd.p, patch_size_src, step_src, dflags.p);
        AP_HIP_CHECK(hipGetLastError());
        ap::block_count_kernel<<<nb, 256, 0, s>>>(dflags.p, dtd.p, dbd.p, dcounts.p);
        AP_HIP_CHECK(hipGetLastError());
    } else {
        // ---- bucket every polygon's edges by the grid rows whose probe band [cy - shift, cy + shift] their y-range meets
        //      (cy = y0 + iy * step + half): an edge outside that band cannot change a probe's crossing count or put the
        //      probe on the boundary (cv::pointPolygonTest skips it without the on-vertex case)
        const int half = patch_size_src / 2, shift = half / 2;
        auto rows_of = [&](const ap::TissueDesc& td, int ya, int yb, int& r0, int& r1) {
            const long ylo = ya < yb ? ya : yb, yhi = ya < yb ? yb : ya;
            const long a = ylo - shift - td.y0 - half, b = yhi + shift - td.y0 - half;
            r0 = a <= 0 ? 0 : (int)((a + step_src - 1) / step_src);
            r1 = b < 0 ? -1 : (int)(b / step_src);
            if (r1 > td.ny - 1) r1 = td.ny - 1;
        };
        std::vector<int> fill;
        auto bucket = [&](const ap::TissueDesc& td, int off, int cnt) {
            const int ptr_off = (int)rowptr.size();
            rowptr.resize(rowptr.size() + td.ny + 1, 0);
            fill.assign(td.ny + 1, 0);
            for (int i = 0; i < cnt; ++i) {
                const int2 v0 = verts[off + (i ? i - 1 : cnt - 1)], v = verts[off + i];
                int r0, r1;
                rows_of(td, v0.y, v.y, r0, r1);
                for (int r = r0; r <= r1; ++r) ++fill[r + 1];
            }
            const int base = (int)edges.size();
            for (int r = 0; r < td.ny; ++r) fill[r + 1] += fill[r];
            for (int r = 0; r <= td.ny; ++r) rowptr[ptr_off + r] = base + fill[r];
            edges.resize(edges.size() + (size_t)fill[td.ny]);
            for (int i = 0; i < cnt; ++i) {
                const int2 v0 = verts[off + (i ? i - 1 : cnt - 1)], v = verts[off + i];
                int r0, r1;
                rows_of(td, v0.y, v.y, r0, r1);
                for (int r = r0; r <= r1; ++r) edges[(size_t)base + fill[r]++] = make_int4(v0.x, v0.y, v.x, v.y);
            }
            return ptr_off;
        };
        for (size_t ti = 0; ti < tds.size(); ++ti) {
            const ap::TissueDesc& td = tds[ti];
            trows[ti].ptr_off = bucket(td, td.poly_off, td.poly_cnt);
            for (int h = 0; h < td.hole_cnt; ++h)
                hrows[td.hole_first + h].ptr_off = bucket(td, hds[td.hole_first + h].poly_off, hds[td.hole_first + h].poly_cnt);
            for (int iy = 0; iy < td.ny; ++iy)
                for (int ix0 = 0; ix0 < td.nx; ix0 += 256)
                    rbs.push_back({(int)ti, iy, ix0, td.nx - ix0 < 256 ? td.nx - ix0 : 256});
        }
        AP_REQUIRE(edges.size() < (1ull << 31) && rowptr.size() < (1ull << 31), "grid_coords: edge table too large");
        nb = (unsigned)rbs.size();
        if ((rc = dedges.alloc(edges.size())) || (rc = drowptr.alloc(rowptr.size())) || (rc = dtrows.alloc(trows.size())) ||
            (rc = dhrows.alloc(hrows.size())) || (rc = drb.alloc(rbs.size())) || (rc = dcounts.alloc(nb)) || (rc = doffs.alloc(nb)))
            return rc;
        if (!edges.empty())
            AP_HIP_CHECK(hipMemcpyAsync(dedges.p, edges.data(), edges.size() * sizeof(int4), hipMemcpyHostToDevice, s));
        AP_HIP_CHECK(hipMemcpyAsync(drowptr.p, rowptr.data(), rowptr.size() * sizeof(int), hipMemcpyHostToDevice, s));
        AP_HIP_CHECK(hipMemcpyAsync(dtrows.p, trows.data(), trows.size() * sizeof(ap::PolyRows), hipMemcpyHostToDevice, s));
        if (!hrows.empty())
            AP_HIP_CHECK(hipMemcpyAsync(dhrows.p, hrows.data(), hrows.size() * sizeof(ap::PolyRows), hipMemcpyHostToDevice, s));
        AP_HIP_CHECK(hipMemcpyAsync(drb.p, rbs.data(), rbs.size() * sizeof(ap::RowBlock), hipMemcpyHostToDevice, s));
        ap::grid_rows_kernel<<<nb, 256, 0, s>>>(dedges.p, drowptr.p, dtd.p, dhd.p, dtrows.p, dhrows.p, drb.p, patch_size_src,
                                                step_src, dflags.p);
        AP_HIP_CHECK(hipGetLastError());
        ap::rowblock_count_kernel<<<nb, 256, 0, s>>>(dflags.p, dtd.p, drb.p, dcounts.p);
        AP_HIP_CHECK(hipGetLastError());
    }
    ap::scan_kernel<<<1, 1024, 0, s>>>(dcounts.p, doffs.p, (int)nb, dtotal.p);
    AP_HIP_CHECK(hipGetLastError());
    unsigned long long total = 0;
    AP_HIP_CHECK(hipMemcpyAsync(&total, dtotal.p, sizeof(total), hipMemcpyDeviceToHost, s));
    AP_HIP_CHECK(hipStreamSynchronize(s));
    *n_rows = (size_t)total;
    if (total == 0) return AP_OK;
    if (total > cap) {
        ap::set_error("grid_coords: %llu rows exceed the output capacity %zu", total, cap);
        return AP_ERR_CAPACITY;
    }
    if ((rc = drows.alloc((size_t)total * 5))) return rc;
    if (legacy)
        ap::compact_kernel<<<nb, 256, 0, s>>>(dflags.p, dtd.p, dbd.p, doffs.p, step_src, read_w, read_h, level, drows.p, total);
    else
        ap::rowblock_compact_kernel<<<nb, 256, 0, s>>>(dflags.p, dtd.p, drb.p, doffs.p, step_src, read_w, read_h, level, drows.p,
                                                       total);
    AP_HIP_CHECK(hipGetLastError());
    AP_HIP_CHECK(hipMemcpyAsync(coords, drows.p, (size_t)total * 5 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    AP_HIP_CHECK(hipStreamSynchronize(s));
    return AP_OK;
}

}  // extern "C"
