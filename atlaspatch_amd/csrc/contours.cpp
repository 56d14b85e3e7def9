// Border following for the tissue mask (host C++, inherently sequential raster scan).
//
// Replaces cv2.findContours(mask_u8, RETR_CCOMP, CHAIN_APPROX_NONE) + the area / hole filters of
// /root/reference/atlas_patch/utils/contours.py:41-116 and scale_contours (:119-131).
//
// Algorithm: Suzuki & Abe (1985) "Topological structural analysis of digitized binary images by
// border following", Algorithm 1, in the form OpenCV's raster scanner runs it: the binary image
// is framed with one zero pixel; foreground is 8-connected, holes 4-connected; every pixel of a
// followed border is marked with +nbd, or -nbd when its east neighbour was examined as a zero
// pixel, so that no border is followed twice; the parent of a new border is derived from LNBD
// (the last border number met on the current row) with Suzuki's table.  RETR_CCOMP flattening:
// outer borders are top level, a hole's parent is the outer border of its component; each new
// node is pushed at the head of its parent's child list and the output is a pre-order walk, so
// top-level contours come out in reverse discovery order, each followed by its holes in reverse
// discovery order.
//
// Unlike OpenCV (signed-char marks, 127 border numbers, bounding-box parent search) the marks
// here are int32 border numbers, which makes Suzuki's parent rule exact.  This file is a
// different formulation from oracle/cv2_restated.py (which scans only 0/1 transitions and takes
// parents from a connected-component labelling); the tests check they agree.
//
// The mask is <= 1024 x 1024 (SegmentationConfig.thumbnail_max), ~1 ms of host work per slide;
// the per-cell point-in-polygon scan that dominates (extraction.py:83-128) runs on the GPU
// (coords.hip).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include "coords_internal.h"

namespace ap {

namespace {

const int kDX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
const int kDY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

struct RawBorder {
    bool is_hole;
    int parent;                 // discovery index of the parent border, -1 = frame
    std::vector<int32_t> xy;    // x0, y0, x1, y1, ... (unpadded mask coordinates)
};

// Follows one border starting at padded position `start`; marks visited pixels with +-nbd.
void follow(std::vector<int32_t>& img, int stride, int start, bool is_hole, int nbd,
            std::vector<int32_t>& out_xy) {
    int delta[8];
    for (int k = 0; k < 8; ++k) delta[k] = kDX[k] + kDY[k] * stride;
    const int i0 = start;
    int s_end = is_hole ? 0 : 4, s = s_end, i1 = 0;
    do {
        s = (s - 1) & 7;
        i1 = i0 + delta[s];
    } while (img[i1] == 0 && s != s_end);

    auto emit = [&](int pos) {
        out_xy.push_back(pos % stride - 1);
        out_xy.push_back(pos / stride - 1);
    };
    if (s == s_end) {           // isolated pixel
        img[i0] = -nbd;
        emit(i0);
        return;
    }
    int i3 = i0;
    for (;;) {
        s_end = s;
        int i4;
        for (;;) {
            ++s;
            i4 = i3 + delta[s & 7];
            if (img[i4] != 0) break;
        }
        s &= 7;
        if (s >= 1 && s <= s_end)          // the search wrapped past east: east was a zero pixel
            img[i3] = -nbd;
        else if (img[i3] == 1)
            img[i3] = nbd;
        emit(i3);
        if (i4 == i0 && i3 == i1) break;
        i3 = i4;
        s = (s + 4) & 7;
    }
}

}  // namespace

// binary: h*w bytes (non-zero = tissue).  Fills `out` with the filtered, ordered contour set.
void contours_from_binary(const uint8_t* binary, int h, int w, double tissue_area_thresh,
                          int min_hole_area, int max_n_holes, double sx, double sy,
                          ContourSet& out) {
    const int stride = w + 2;
    std::vector<int32_t> img((size_t)(h + 2) * stride, 0);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) img[(size_t)(y + 1) * stride + x + 1] = binary[(size_t)y * w + x] ? 1 : 0;

    std::vector<RawBorder> found;      // discovery order; border number nbd = index + 2
    for (int y = 1; y <= h; ++y) {
        int lnbd = 1;                  // 1 = the frame
        int32_t* row = img.data() + (size_t)y * stride;
        for (int x = 1; x <= w; ++x) {
            const int32_t p = row[x], prev = row[x - 1];
            bool start = false, is_hole = false;
            int origin = 0;
            if (p != prev) {
                if (prev == 0 && p == 1) {
                    start = true; origin = y * stride + x;
                } else if (p == 0 && prev >= 1) {
                    start = true; is_hole = true; origin = y * stride + x - 1;
                    // lnbd already reflects pixel x-1 when that pixel carries a border number
                }
            }
            if (start) {
                RawBorder b;
                b.is_hole = is_hole;
                // Suzuki's parent table, B' = border numbered lnbd
                if (lnbd <= 1) {
                    b.parent = -1;
                } else {
                    const RawBorder& bp = found[lnbd - 2];
                    b.parent = (bp.is_hole == is_hole) ? bp.parent : (lnbd - 2);
                }
                const int nbd = (int)found.size() + 2;
                follow(img, stride, origin, is_hole, nbd, b.xy);
                found.push_back(std::move(b));
            }
            const int32_t now = row[x];
            if (now != 0 && now != 1) lnbd = std::abs(now);
        }
    }

    std::vector<BorderSummary> summary(found.size());
    for (size_t k = 0; k < found.size(); ++k) {
        const std::vector<int32_t>& xy = found[k].xy;
        const size_t n = xy.size() / 2;
        long long a2 = 0;
        if (n) {
            long long px = xy[2 * (n - 1)], py = xy[2 * (n - 1) + 1];
            for (size_t i = 0; i < n; ++i) {
                const long long x = xy[2 * i], y = xy[2 * i + 1];
                a2 += px * y - py * x;
                px = x; py = y;
            }
        }
        summary[k] = {0, found[k].is_hole ? 1 : 0, (int)n, found[k].parent, a2};
    }
    Selection sel;
    select_contours(summary, h, w, tissue_area_thresh, min_hole_area, max_n_holes, sel);
    build_contour_set(sel, sx, sy, [&](int d) { return found[d].xy; }, out);
}

void select_contours(const std::vector<BorderSummary>& found, int h, int w, double tissue_area_thresh, int min_hole_area,
                     int max_n_holes, Selection& sel) {
    // RETR_CCOMP: a hole's parent (Suzuki tree) is already the outer border of its component.
    // flat order: outer borders in reverse discovery order, each followed by its holes reversed.
    const int nb = (int)found.size();
    std::vector<std::vector<int>> kids(nb);
    std::vector<int> top;
    for (int k = 0; k < nb; ++k) {
        if (!found[k].is_hole) { top.push_back(k); continue; }
        int par = found[k].parent;
        if (par < 0 || par >= nb || found[par].is_hole) {       // cannot happen for a well-formed scan; stay safe
            par = -1;
            for (int j = k - 1; j >= 0 && par < 0; --j) if (!found[j].is_hole) par = j;
            if (par < 0) continue;
        }
        kids[par].push_back(k);
    }
    std::reverse(top.begin(), top.end());
    struct Flat { int disc; int parent_flat; };
    std::vector<Flat> flat;
    for (int k : top) {
        const int me = (int)flat.size();
        flat.push_back({k, -1});
        for (auto it = kids[k].rbegin(); it != kids[k].rend(); ++it) flat.push_back({*it, me});
    }

    // ---- mask_to_contours filters (contours.py:80-114).  cv2.contourArea = |shoelace / 2| in float64: the sums are integers
    // far below 2^53, so the exact integer sum converted once equals OpenCV's running double sum
    const double min_area = tissue_area_thresh * (double)((double)h * (double)w);
    const double hole_thr = (double)min_hole_area;
    std::vector<double> area(flat.size());
    std::vector<int> tissue_flat;                       // flat indices kept as tissue
    std::vector<int> hole_flat;                         // holes passing the area test, flat order
    for (size_t i = 0; i < flat.size(); ++i) {
        area[i] = std::fabs((double)found[flat[i].disc].area2 * 0.5);
        if (flat[i].parent_flat < 0) {
            if (area[i] >= min_area) tissue_flat.push_back((int)i);
        } else if (area[i] >= hole_thr) {
            hole_flat.push_back((int)i);
        }
    }
    std::vector<char> allowed(flat.size(), 1);
    if (max_n_holes > 0 && (int)hole_flat.size() > max_n_holes) {
        std::vector<int> ranked = hole_flat;            // stable, descending by area
        std::stable_sort(ranked.begin(), ranked.end(), [&](int a, int b) { return area[a] > area[b]; });
        std::fill(allowed.begin(), allowed.end(), 0);
        for (int i = 0; i < max_n_holes; ++i) allowed[ranked[i]] = 1;
    }
    sel.tissue.clear();
    sel.holes.clear();
    for (int ti : tissue_flat) {
        sel.tissue.push_back(flat[ti].disc);
        std::vector<int> hs;
        for (int hf : hole_flat)
            if (flat[hf].parent_flat == ti && allowed[hf]) hs.push_back(flat[hf].disc);
        sel.holes.push_back(std::move(hs));
    }
}

}  // namespace ap
