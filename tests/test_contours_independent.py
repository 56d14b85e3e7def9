"""Independent characterisation of what ``cv2.findContours(RETR_CCOMP, CHAIN_APPROX_NONE)`` returns, checked against the oracle's
restatement (oracle/cv2_restated.py -- the thing G4 and every device test lean on; reference call site utils/contours.py:59).

OpenCV is absent from this image, so the restatement cannot be pinned against cv2 itself.  What CAN be stated without any border
follower, from connected-component labelling alone (scipy.ndimage.label -- a different algorithm from a different library):

  * with 8-connected foreground and 4-connected background (a zero frame added), every pair (foreground component k, background
    component r) that touch through a 4-neighbourhood gives exactly ONE border, whose point SET is
    S(k, r) = {p in k : some 4-neighbour of p lies in r};
  * it is an outer border iff r is the exterior of k -- the background component of the pixel to the left of k's raster-first
    pixel -- and a hole border otherwise; RETR_CCOMP makes every outer border top level and every hole a child of the outer
    border of its own foreground component.

The traversal order inside a border and the order of the borders are OpenCV's and stay with the restatement; the sets, the
count, the outer / hole split and the parent links are checked here.  Second independent check: the integer
``pointPolygonTest`` against Pillow's polygon rasteriser for points away from the boundary.
"""
from __future__ import annotations

import numpy as np
import pytest
from scipy import ndimage as ndi

from oracle import cv2_restated as cv2r

N4 = ((0, 1), (0, -1), (1, 0), (-1, 0))


def _masks():
    rng = np.random.default_rng(11)
    out = {}
    yy, xx = np.mgrid[0:96, 0:128]
    blob = ((yy - 48) ** 2 / 38.0 ** 2 + (xx - 64) ** 2 / 52.0 ** 2) <= 1.0
    out["blob"] = blob
    holes = blob & ~(((yy - 40) ** 2 + (xx - 50) ** 2) <= 9 ** 2) & ~(((yy - 58) ** 2 + (xx - 86) ** 2) <= 12 ** 2)
    out["blob_with_holes"] = holes
    island = holes | (((yy - 58) ** 2 + (xx - 86) ** 2) <= 5 ** 2)
    out["island_in_hole"] = island
    out["touching_border"] = np.pad(np.ones((40, 50), bool), ((0, 56), (0, 78))) | (xx > 120)
    out["noise_40"] = rng.random((72, 80)) < 0.40
    out["noise_60"] = rng.random((72, 80)) < 0.60
    out["diagonals"] = (np.eye(48, dtype=bool) | np.eye(48, dtype=bool)[::-1])          # 8-connected only through corners
    out["checker"] = ((yy[:32, :32] + xx[:32, :32]) % 2 == 0)
    out["ragged"] = ndi.binary_opening(rng.random((96, 128)) < 0.55, iterations=1) | (rng.random((96, 128)) < 0.03)
    return out


def _independent_borders(mask: np.ndarray):
    """{(fg label, bg label): frozenset of (x, y)} plus {fg label: exterior bg label}, from labelling alone."""
    h, w = mask.shape
    pad = np.zeros((h + 2, w + 2), bool)
    pad[1:-1, 1:-1] = mask
    fg, nfg = ndi.label(pad, structure=np.ones((3, 3), int))
    bg, _ = ndi.label(~pad, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    borders: dict = {}
    ys, xs = np.nonzero(pad)
    for y, x in zip(ys.tolist(), xs.tolist()):
        k = int(fg[y, x])
        for dy, dx in N4:
            r = int(bg[y + dy, x + dx])
            if r:
                borders.setdefault((k, r), set()).add((x - 1, y - 1))
    exterior = {}
    for k in range(1, nfg + 1):
        y, x = np.argwhere(fg == k)[0]                       # raster-first pixel (argwhere is row-major)
        exterior[k] = int(bg[y, x - 1])
        assert exterior[k] != 0
    return {key: frozenset(v) for key, v in borders.items()}, exterior, fg


@pytest.mark.parametrize("name", list(_masks()))
def test_border_sets_counts_and_parents_follow_from_component_labelling(name):
    mask = _masks()[name]
    contours, hierarchy = cv2r.findContours(mask.astype(np.uint8) * 255, cv2r.RETR_CCOMP, cv2r.CHAIN_APPROX_NONE)
    hier = np.asarray(hierarchy).reshape(-1, 4) if len(contours) else np.zeros((0, 4), int)
    borders, exterior, fg = _independent_borders(mask)
    assert len(contours) == len(borders)                                  # one border per touching (component, region) pair
    got = {}
    for idx, c in enumerate(contours):
        pts = frozenset(map(tuple, np.asarray(c).reshape(-1, 2).tolist()))
        comps = {int(fg[y + 1, x + 1]) for x, y in pts}
        assert len(comps) == 1 and 0 not in comps                         # a border lies on ONE foreground component
        got[idx] = (comps.pop(), pts)
    remaining = dict(borders)
    outer_of = {}
    for idx, (k, pts) in got.items():
        matches = [key for key, s in remaining.items() if key[0] == k and s == pts]
        assert matches, (name, idx, "no (component, region) pair has this point set")
        key = matches[0]
        del remaining[key]
        is_outer = key[1] == exterior[k]
        assert is_outer == (hier[idx, 3] == -1)                           # outer <=> top level
        if is_outer:
            outer_of[k] = idx
    assert not remaining
    assert len(outer_of) == int(fg.max())                                 # one outer border per foreground component
    for idx, (k, _) in got.items():
        if hier[idx, 3] != -1:
            assert hier[idx, 3] == outer_of[k]                            # a hole's parent = the outer border of its component


@pytest.mark.parametrize("name", ["blob", "blob_with_holes", "island_in_hole", "noise_60", "ragged"])
def test_point_polygon_test_agrees_with_pillow_rasteriser_away_from_the_boundary(name):
    """pointPolygonTest(measureDist=False) is +1 / 0 / -1 (utils/contours.py:37, services/extraction.py:79).  Pillow fills a polygon
    with its own scanline code; at integer points more than one pixel away from every contour vertex both must agree on inside /
    outside (the boundary convention itself is OpenCV's and stays with the restatement)."""
    from PIL import Image, ImageDraw
    mask = _masks()[name]
    contours, _ = cv2r.findContours(mask.astype(np.uint8) * 255, cv2r.RETR_CCOMP, cv2r.CHAIN_APPROX_NONE)
    h, w = mask.shape
    rng = np.random.default_rng(5)
    checked = 0
    for c in sorted(contours, key=len, reverse=True)[:6]:
        poly = np.asarray(c).reshape(-1, 2)
        if len(poly) < 8:
            continue
        img = Image.new("L", (w, h), 0)
        ImageDraw.Draw(img).polygon([tuple(p) for p in poly.tolist()], fill=1, outline=1)
        inside = np.asarray(img).astype(bool)
        near = np.zeros((h, w), bool)
        near[poly[:, 1], poly[:, 0]] = True
        near = ndi.binary_dilation(near, structure=np.ones((3, 3), bool), iterations=2)
        for x, y in zip(rng.integers(0, w, 400).tolist(), rng.integers(0, h, 400).tolist()):
            if near[y, x]:
                continue
            r = cv2r.pointPolygonTest(c, (float(x), float(y)), False)
            assert (r > 0) == bool(inside[y, x]) and r != 0, (name, x, y, r)
            checked += 1
    assert checked > 200


def test_contour_area_satisfies_picks_theorem_on_simple_borders():
    """contourArea = |shoelace| over the vertex list (utils/contours.py:91,104).  A border traced with CHAIN_APPROX_NONE is a lattice
    polygon whose boundary lattice points are exactly its vertices, so for a SIMPLE border Pick's theorem gives the area without
    the shoelace sum: A = I + B / 2 - 1, with I counted from the foreground pixels strictly inside (labelling + the border set)."""
    mask = _masks()["blob_with_holes"]
    contours, hierarchy = cv2r.findContours(mask.astype(np.uint8) * 255, cv2r.RETR_CCOMP, cv2r.CHAIN_APPROX_NONE)
    hier = np.asarray(hierarchy).reshape(-1, 4)
    h, w = mask.shape
    done = 0
    for idx, c in enumerate(contours):
        poly = np.asarray(c).reshape(-1, 2)
        pts = set(map(tuple, poly.tolist()))
        if len(pts) != len(poly):
            continue                                            # the border revisits a pixel: not a simple polygon
        on = np.zeros((h, w), bool)
        on[poly[:, 1], poly[:, 0]] = True
        # lattice points strictly inside the closed curve = everything the curve separates from the frame, minus the curve
        outside, _ = ndi.label(~np.pad(on, 1), structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
        interior = int(((outside != outside[0, 0]) & ~np.pad(on, 1)).sum())
        assert cv2r.contourArea(c) == interior + len(poly) / 2.0 - 1.0, (idx, hier[idx].tolist())
        done += 1
    assert done >= 3                                            # the blob's outer border and its two holes
