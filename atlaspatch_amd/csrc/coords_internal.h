// Shared between contours.cpp (host border following) and coords.hip (device grid scan).
#pragma once
#include <cstdint>
#include <vector>

namespace ap {

struct Polygon {
    std::vector<int32_t> raw;      // x, y pairs in mask space
    std::vector<int32_t> scaled;   // x, y pairs at level 0 (scale_contours semantics)
};

struct Tissue {
    int poly = -1;                 // index into ContourSet::polys
    std::vector<int> holes;        // indices into ContourSet::polys
};

struct ContourSet {
    std::vector<Polygon> polys;
    std::vector<Tissue> tissue;    // in the reference's output order
};

void contours_from_binary(const uint8_t* binary, int h, int w, double tissue_area_thresh,
                          int min_hole_area, int max_n_holes, double sx, double sy, ContourSet& out);

// One discovered border (outer border of a foreground component, or border of an enclosed hole) before the filters, in
// DISCOVERY order (raster order of the scan position where Suzuki's algorithm starts it).
struct BorderSummary {
    int origin;          // start pixel (unpadded linear index y * w + x)
    int is_hole;
    int n;               // number of points
    int parent;          // holes: discovery index of the component's outer border; outer borders: -1
    long long area2;     // twice the signed shoelace sum over the point sequence (an exact integer: |coordinates| < 2^15)
};
// What mask_to_contours keeps (contours.py:80-114), in its output order: RETR_CCOMP flattening (outer borders in reverse
// discovery order, each followed by its holes reversed), tissue = top level with area >= thresh * h * w, holes with area >=
// min_hole_area, at most max_n_holes holes overall (largest first, stable).
struct Selection {
    std::vector<int> tissue;                 // discovery indices
    std::vector<std::vector<int>> holes;     // per tissue contour, discovery indices
};
// device form (contours_device.hip): dbin = device uint8 [h * w]; `stream` is a hipStream_t.  Supported while the padded bit
// image fits the LDS (<= ~1100 x 1100; the masks are <= 1024 x 1024, SegmentationConfig.thumbnail_max)
bool contours_device_supported(int h, int w);
int contours_from_binary_device(const uint8_t* dbin, int h, int w, double tissue_area_thresh, int min_hole_area, int max_n_holes,
                                double sx, double sy, ContourSet& out, void* stream);
void select_contours(const std::vector<BorderSummary>& found, int h, int w, double tissue_area_thresh, int min_hole_area,
                     int max_n_holes, Selection& sel);
// ContourSet from the selection; points(d) = x, y pairs of border d (only selected borders are asked for)
template <typename PointsFn>
void build_contour_set(const Selection& sel, double sx, double sy, PointsFn points, ContourSet& out) {
    out.polys.clear();
    out.tissue.clear();
    auto push_poly = [&](int d) {
        Polygon pg;
        pg.raw = points(d);
        pg.scaled.resize(pg.raw.size());
        const float fsx = (float)sx, fsy = (float)sy;      // numpy: f32 array *= python float
        for (size_t i = 0; i + 1 < pg.raw.size(); i += 2) {
            volatile float vx = (float)pg.raw[i] * fsx;
            volatile float vy = (float)pg.raw[i + 1] * fsy;
            pg.scaled[i] = (int32_t)vx;                      // astype(int32): truncation
            pg.scaled[i + 1] = (int32_t)vy;
        }
        out.polys.push_back(std::move(pg));
        return (int)out.polys.size() - 1;
    };
    for (size_t t = 0; t < sel.tissue.size(); ++t) {
        Tissue ts;
        ts.poly = push_poly(sel.tissue[t]);
        for (int hd : sel.holes[t]) ts.holes.push_back(push_poly(hd));
        out.tissue.push_back(std::move(ts));
    }
}


}  // namespace ap
