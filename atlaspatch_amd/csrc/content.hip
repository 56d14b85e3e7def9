// Per-tile content statistics for the reference's --no-fast-mode filters (utils/image.py:7-41):
//   black:  gray = cv2.cvtColor(RGB2GRAY) < black_thresh            (is_black_patch)
//   white:  HSV saturation < sat_thresh  AND  value >= value_thresh  (is_white_patch)
// counted per tile; the host applies `count / pixels >= 0.7` exactly as the reference does.
//
// OpenCV's 8-bit conversions are integer fixed point and are restated bit-exactly:
//   gray = (4899 R + 9617 G + 1868 B + 8192) >> 14                       (RGB2GRAY, 14-bit coefficients)
//   V = max(R,G,B);  S = (diff * sdiv[V] + 2048) >> 12,  diff = V - min(R,G,B),
//   sdiv[0] = 0, sdiv[v] = round_half_even(1044480 / v)                  (RGB2HSV, hsv_shift = 12)
//
// One workgroup per (tile, slice): a lane takes 16 pixels as three 16-byte loads (48 contiguous bytes),
// predicates are counted with wave ballots + popcount, one atomicAdd per wave.  HBM-bound (3 B / pixel).
#include "ap_common.h"

namespace ap {
namespace {

__global__ __launch_bounds__(256) void tile_content_kernel(const uint8_t* __restrict__ tiles, int groups_per_tile,
                                                           int black_thresh, int sat_thresh, int value_thresh,
                                                           unsigned* __restrict__ counts) {
    __shared__ int sdiv[256];
    for (int v = threadIdx.x; v < 256; v += 256)
        sdiv[v] = v == 0 ? 0 : (int)__builtin_rint(1044480.0 / (double)v);      // cvRound: half to even
    __syncthreads();
    const int tile = blockIdx.y;
    const u32x4* src = (const u32x4*)(tiles + (size_t)tile * groups_per_tile * 48);
    unsigned nblack = 0, nwhite = 0;
    for (int gidx = blockIdx.x * 256 + threadIdx.x; gidx < groups_per_tile; gidx += gridDim.x * 256) {
        const u32x4 a = src[gidx * 3], b = src[gidx * 3 + 1], c = src[gidx * 3 + 2];
        const uint32_t w[12] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], c[0], c[1], c[2], c[3]};
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int byte0 = p * 3;
            auto byte_at = [&](int i) { return (int)((w[i >> 2] >> ((i & 3) * 8)) & 0xffu); };
            const int r = byte_at(byte0), g = byte_at(byte0 + 1), bl = byte_at(byte0 + 2);
            const int gray = (4899 * r + 9617 * g + 1868 * bl + 8192) >> 14;
            const int vmax = max(r, max(g, bl)), vmin = min(r, min(g, bl));
            const int sat = ((vmax - vmin) * sdiv[vmax] + 2048) >> 12;
            nblack += gray < black_thresh ? 1u : 0u;
            nwhite += (sat < sat_thresh && vmax >= value_thresh) ? 1u : 0u;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        nblack += __shfl_xor(nblack, off, 64);
        nwhite += __shfl_xor(nwhite, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&counts[tile * 2], nblack);
        atomicAdd(&counts[tile * 2 + 1], nwhite);
    }
}

}  // namespace

int tile_content_counts(const uint8_t* tiles, int n, int h, int w, int black_thresh, int sat_thresh,
                        int value_thresh, unsigned* counts, hipStream_t stream) {
    AP_REQUIRE(tiles && counts, "tile_content: null pointer");
    AP_REQUIRE(n >= 0 && h > 0 && w > 0 && ((size_t)h * w) % 16 == 0, "tile_content: %d x %d tiles (pixels per tile "
               "must be a multiple of 16)", h, w);
    AP_REQUIRE(((uintptr_t)tiles & 15) == 0, "tile_content: tiles must be 16-byte aligned");
    if (n == 0) return AP_OK;
    AP_HIP_CHECK(hipMemsetAsync(counts, 0, (size_t)n * 2 * sizeof(unsigned), stream));
    const int groups = (int)((size_t)h * w / 16);
    int slices = (groups + 255) / 256;
    if (slices > 4) slices = 4;
    dim3 grid(slices, n), block(256);
    tile_content_kernel<<<grid, block, 0, stream>>>(tiles, groups, black_thresh, sat_thresh, value_thresh, counts);
    AP_HIP_CHECK(hipGetLastError());
    return AP_OK;
}

}  // namespace ap
