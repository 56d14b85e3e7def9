"""CPU restatement of the five OpenCV primitives the AtlasPatch hot path calls.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: opencv-python
(``opencv-python>=4.7.0``, /root/reference/pyproject.toml:35) is not installed in
this image, so these functions restate OpenCV's published algorithms and are
anchored on the reference's call sites:

  * ``findContours(mask_u8, RETR_CCOMP, CHAIN_APPROX_NONE)``
        atlas_patch/utils/contours.py:59
  * ``contourArea(cnt)``                       atlas_patch/utils/contours.py:91,104
  * ``pointPolygonTest(cnt, (x, y), False)``   atlas_patch/utils/contours.py:37,
                                               atlas_patch/services/extraction.py:79
  * ``boundingRect(cnt)``                      atlas_patch/services/extraction.py:94

Algorithms restated
  * findContours: Suzuki & Abe 1985 border following exactly as OpenCV's raster
    scanner runs it (modules/imgproc/src/contours.cpp: cvFindNextContour +
    icvFetchContour): the image is binarised (non-zero -> 1) and framed with one
    zero pixel; foreground is 8-connected, holes are 4-connected; an outer border
    starts where ``prev == 0 and p == 1``, a hole border where ``p == 0 and
    prev >= 1`` (origin = the foreground pixel to the left); visited pixels are
    marked ``nbd`` (or negative when their east neighbour was examined as zero) so
    no border is followed twice.  RETR_CCOMP: every outer border is top level,
    every hole border is a child of the outer border of its own component.
    Every new node is inserted at the HEAD of its parent's child list and the flat
    output is a pre-order walk (cvInsertNodeIntoTree / cvTreeToNodeSeq), so
    top-level contours come out in reverse discovery order, each followed by its
    holes in reverse discovery order.  hierarchy rows = [next, prev, child, parent].
  * contourArea: |1/2 * sum(x_{i-1} * y_i - y_{i-1} * x_i)| in float64.
  * boundingRect (int32 points): (min x, min y, max x - min x + 1, max y - min y + 1).
  * pointPolygonTest, measureDist=False, integral point, int32 contour: OpenCV's
    "purely integer" branch (modules/imgproc/src/geometry.cpp).
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage as _ndi

RETR_EXTERNAL = 0
RETR_LIST = 1
RETR_CCOMP = 2
RETR_TREE = 3
CHAIN_APPROX_NONE = 1
CHAIN_APPROX_SIMPLE = 2

# 8 directions, OpenCV order: 0=E, 1=NE, 2=N, 3=NW, 4=W, 5=SW, 6=S, 7=SE (y grows down)
_DX = (1, 1, 0, -1, -1, -1, 0, 1)
_DY = (0, -1, -1, -1, 0, 1, 1, 1)

_POS_MARK = 2      # any value > 1  ("nbd")
_NEG_MARK = -126   # any value < 0  ("nbd | -128")


def _fetch_contour(img: list, stride: int, start: int, is_hole: bool) -> list:
    """icvFetchContour for CHAIN_APPROX_NONE on a flat framed image (list of ints).

    Returns the flat indices of the border pixels in visiting order and applies
    the visit marks to ``img`` in place.
    """
    deltas = tuple(_DX[k] + _DY[k] * stride for k in range(8))
    i0 = start
    s_end = s = 0 if is_hole else 4
    while True:
        s = (s - 1) & 7
        i1 = i0 + deltas[s]
        if img[i1] != 0 or s == s_end:
            break
    if s == s_end:                       # isolated pixel
        img[i0] = _NEG_MARK
        return [i0]

    out = []
    i3 = i0
    while True:
        s_end = s
        while True:
            s += 1
            i4 = i3 + deltas[s & 7]
            if img[i4] != 0:
                break
        s &= 7
        # east neighbour examined as zero during this visit <=> the search wrapped
        if 1 <= s <= s_end:
            img[i3] = _NEG_MARK
        elif img[i3] == 1:
            img[i3] = _POS_MARK
        out.append(i3)
        if i4 == i0 and i3 == i1:
            break
        i3 = i4
        s = (s + 4) & 7
    return out


def findContours(image, mode, method):
    """Restates cv2.findContours for mode=RETR_CCOMP, method=CHAIN_APPROX_NONE."""
    if mode != RETR_CCOMP or method != CHAIN_APPROX_NONE:
        raise NotImplementedError("oracle covers RETR_CCOMP + CHAIN_APPROX_NONE only")
    src = np.asarray(image)
    if src.ndim != 2:
        raise ValueError("findContours expects a single-channel image")
    h, w = src.shape
    binary = np.zeros((h + 2, w + 2), dtype=np.int8)
    binary[1:-1, 1:-1] = (src != 0)
    stride = w + 2

    # component labels (8-connected foreground) give each hole its CCOMP parent
    labels, _ = _ndi.label(binary, structure=np.ones((3, 3), dtype=np.uint8))

    img = binary.ravel().tolist()
    # positions where the binary value changes along a row are the only places a
    # border can start (outer: 0->1, hole: 1->0)
    diff = binary[:, 1:] != binary[:, :-1]          # diff[y, x-1] <=> b[y,x] != b[y,x-1]
    ys, xs = np.nonzero(diff[1:h + 1, :w])          # x-1 in [0, w) -> x in [1, w]
    ys = ys + 1
    xs = xs + 1

    found = []   # (is_hole, component label, [flat indices])
    for y, x in zip(ys.tolist(), xs.tolist()):
        pos = y * stride + x
        p = img[pos]
        prev = img[pos - 1]
        if p == prev:
            continue
        if prev == 0 and p == 1:
            is_hole = False
            origin = pos
        elif p == 0 and prev >= 1:
            is_hole = True
            origin = pos - 1
        else:
            continue
        pts = _fetch_contour(img, stride, origin, is_hole)
        found.append((is_hole, int(labels[origin // stride, origin % stride]), pts))

    if not found:
        return (), None

    outer_of_label = {}
    for k, (is_hole, lab, _) in enumerate(found):
        if not is_hole:
            outer_of_label[lab] = k
    children = {k: [] for k, f in enumerate(found) if not f[0]}
    for k, (is_hole, lab, _) in enumerate(found):
        if is_hole:
            children[outer_of_label[lab]].append(k)

    top = [k for k, f in enumerate(found) if not f[0]][::-1]
    order = []          # flat pre-order: discovery ids
    for k in top:
        order.append(k)
        order.extend(children[k][::-1])
    flat_index = {k: i for i, k in enumerate(order)}

    hierarchy = np.full((len(order), 4), -1, dtype=np.int32)
    for pos_in_top, k in enumerate(top):
        i = flat_index[k]
        if pos_in_top + 1 < len(top):
            hierarchy[i, 0] = flat_index[top[pos_in_top + 1]]
        if pos_in_top > 0:
            hierarchy[i, 1] = flat_index[top[pos_in_top - 1]]
        kids = children[k][::-1]
        if kids:
            hierarchy[i, 2] = flat_index[kids[0]]
        for pos_in_kids, c in enumerate(kids):
            j = flat_index[c]
            hierarchy[j, 3] = i
            if pos_in_kids + 1 < len(kids):
                hierarchy[j, 0] = flat_index[kids[pos_in_kids + 1]]
            if pos_in_kids > 0:
                hierarchy[j, 1] = flat_index[kids[pos_in_kids - 1]]

    contours = []
    for k in order:
        idx = np.asarray(found[k][2], dtype=np.int64)
        pts = np.empty((idx.size, 1, 2), dtype=np.int32)
        pts[:, 0, 0] = idx % stride - 1
        pts[:, 0, 1] = idx // stride - 1
        contours.append(pts)
    return tuple(contours), hierarchy[None, :, :]


def contourArea(contour, oriented: bool = False) -> float:
    pts = np.asarray(contour).reshape(-1, 2).astype(np.float64)
    if pts.shape[0] == 0:
        return 0.0
    prev = np.roll(pts, 1, axis=0)
    a00 = float(np.sum(prev[:, 0] * pts[:, 1] - prev[:, 1] * pts[:, 0])) * 0.5
    return a00 if oriented else abs(a00)


def boundingRect(contour):
    pts = np.asarray(contour).reshape(-1, 2)
    if pts.shape[0] == 0:
        return (0, 0, 0, 0)
    x0 = int(pts[:, 0].min())
    y0 = int(pts[:, 1].min())
    x1 = int(pts[:, 0].max())
    y1 = int(pts[:, 1].max())
    return (x0, y0, x1 - x0 + 1, y1 - y0 + 1)


def pointPolygonTest(contour, pt, measureDist: bool) -> float:
    """Integer branch of cv::pointPolygonTest (int32 contour, integral point)."""
    if measureDist:
        raise NotImplementedError("oracle covers measureDist=False only")
    cnt = np.asarray(contour)
    if cnt.dtype != np.int32:
        raise NotImplementedError("oracle covers int32 contours only")
    pts = cnt.reshape(-1, 2)
    total = pts.shape[0]
    if total == 0:
        return -1.0
    fx = np.float32(pt[0])
    fy = np.float32(pt[1])
    px = int(np.rint(fx))
    py = int(np.rint(fy))
    if float(px) != float(fx) or float(py) != float(fy):
        raise NotImplementedError("oracle covers integral query points only")

    vx = pts[:, 0].astype(np.int64)
    vy = pts[:, 1].astype(np.int64)
    v0x = np.roll(vx, 1)
    v0y = np.roll(vy, 1)

    skip = ((v0y <= py) & (vy <= py)) | ((v0y > py) & (vy > py)) | ((v0x < px) & (vx < px))
    on_skip = skip & (vy == py) & (
        (vx == px)
        | ((v0y == py) & (((v0x <= px) & (px <= vx)) | ((vx <= px) & (px <= v0x))))
    )
    if bool(on_skip.any()):
        # an earlier non-skipped edge may also report 0; either way the answer is 0
        return 0.0
    dist = (py - v0y) * (vx - v0x) - (px - v0x) * (vy - v0y)
    live = ~skip
    if bool((live & (dist == 0)).any()):
        return 0.0
    dist = np.where(vy < v0y, -dist, dist)
    counter = int(np.count_nonzero(live & (dist > 0)))
    return -1.0 if counter % 2 == 0 else 1.0


def pointPolygonTest_scalar(contour, pt) -> int:
    """Edge-by-edge loop form of the same branch (used to cross-check the vector form)."""
    pts = np.asarray(contour).reshape(-1, 2).tolist()
    total = len(pts)
    if total == 0:
        return -1
    px, py = int(pt[0]), int(pt[1])
    counter = 0
    vx, vy = pts[-1]
    for i in range(total):
        v0x, v0y = vx, vy
        vx, vy = pts[i]
        if (v0y <= py and vy <= py) or (v0y > py and vy > py) or (v0x < px and vx < px):
            if py == vy and (px == vx or (py == v0y and
                                          ((v0x <= px <= vx) or (vx <= px <= v0x)))):
                return 0
            continue
        dist = (py - v0y) * (vx - v0x) - (px - v0x) * (vy - v0y)
        if dist == 0:
            return 0
        if vy < v0y:
            dist = -dist
        counter += dist > 0
    return -1 if counter % 2 == 0 else 1


# ----------------------------------------------------------------------------- colour conversions (8-bit)
# Used by atlas_patch/utils/image.py:14,33 (is_black_patch / is_white_patch), which
# services/extraction.py:112-116 applies to every candidate tile when fast_mode is off.
# OpenCV's 8-bit paths are integer fixed point (modules/imgproc/src/color_yuv / color_hsv):
#   RGB2GRAY: gray = descale(R*4899 + G*9617 + B*1868, 14) = (... + (1 << 13)) >> 14
#   RGB2HSV (hrange 180):  V = max, diff = V - min,
#             S = (diff * sdiv_table[V] + (1 << 11)) >> 12,  sdiv_table[v] = saturate_cast<int>((255 << 12) / (1. * v))
#             (saturate_cast<int>(double) = cvRound: round half to even), sdiv_table[0] = 0.
_SDIV = np.zeros(256, np.int64)
_SDIV[1:] = np.rint((255 << 12) / np.arange(1, 256, dtype=np.float64)).astype(np.int64)


def cvtColor_RGB2GRAY(rgb: np.ndarray) -> np.ndarray:
    a = rgb.astype(np.int64)
    return ((a[..., 0] * 4899 + a[..., 1] * 9617 + a[..., 2] * 1868 + (1 << 13)) >> 14).astype(np.uint8)


def cvtColor_RGB2HSV_sv(rgb: np.ndarray):
    """(S, V) planes of cv2.cvtColor(rgb, COLOR_RGB2HSV) for uint8 input (H is not used by the path)."""
    a = rgb.astype(np.int64)
    v = a.max(-1)
    diff = v - a.min(-1)
    s = (diff * _SDIV[v] + (1 << 11)) >> 12
    return s.astype(np.uint8), v.astype(np.uint8)


def is_black_patch(patch: np.ndarray, rgb_thresh: int = 40, min_fraction: float = 0.7) -> bool:
    """utils/image.py:7-18."""
    gray = cvtColor_RGB2GRAY(patch)
    return bool(float((gray < rgb_thresh).mean()) >= float(min_fraction))


def is_white_patch(patch: np.ndarray, sat_thresh: int = 5, min_fraction: float = 0.7, value_thresh: int = 200) -> bool:
    """utils/image.py:21-41."""
    s, v = cvtColor_RGB2HSV_sv(patch)
    return bool(float(((s < sat_thresh) & (v >= value_thresh)).mean()) >= float(min_fraction))
