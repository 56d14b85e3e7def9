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

}  // namespace ap
