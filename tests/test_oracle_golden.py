"""CPU tests: the oracle against the golden vectors produced by running the reference itself
(tests/golden/gen_golden.py), plus self-consistency of the restated OpenCV primitives."""
import json
import os

import numpy as np
import pytest

from oracle import coords_oracle, cv2_restated as cv2r
from tests import helpers


def test_oracle_geometry_table(golden_dir):
    table = json.load(open(os.path.join(golden_dir, "geometry.json")))
    for row in table["geometry"]:
        try:
            lv, (rw, rh), pss, ss, p0 = coords_oracle.prepare_geometry(row["ds"], row["mag"], row["tgt"], row["ps"], row["step"])
            got = [lv, rw, rh, pss, ss, p0]
        except ValueError as exc:
            got = {"error": str(exc)}
        assert got == row["out"], row
    for row in table["levels"]:
        if "error" in row:
            with pytest.raises(ValueError):
                coords_oracle.optimal_level(row["ds"], row["target"])
        else:
            lv, extra = coords_oracle.optimal_level(row["ds"], row["target"])
            assert lv == row["level"] and extra == row["extra"], row


def test_oracle_scale_contours(golden_dir):
    g = np.load(os.path.join(golden_dir, "scale_contours.npz"))
    k = 0
    while f"c{k}_in" in g:
        W, H, mw, mh = (int(v) for v in g[f"c{k}_dims"])
        out = coords_oracle.scale_contours([g[f"c{k}_in"].reshape(-1, 1, 2)], W / float(mw), H / float(mh))[0]
        assert np.array_equal(out.reshape(-1, 2), g[f"c{k}_out"]), k
        k += 1
    assert k == 15
    for W in (99999, 100000, 40000, 46000):
        c = np.stack([np.arange(1024), np.arange(1024)], -1).astype(np.int32).reshape(-1, 1, 2)
        out = coords_oracle.scale_contours([c], W / 1024.0, W / 1024.0)[0]
        assert np.array_equal(out.reshape(-1, 2), g[f"sweep{W}"])
    # the float32 truncation really differs from float64 math for W = 99 999
    f64 = (np.arange(1024) * (99999 / 1024.0)).astype(np.int64)
    assert (f64 != g["sweep99999"][:, 0]).sum() > 0


@pytest.mark.parametrize("name", sorted(helpers.load_coords_cases().keys()))
def test_oracle_coords_match_reference(name):
    case = helpers.load_coords_cases()[name]
    cfg = case["info"]["config"]
    coords, geom = coords_oracle.coords_from_mask(
        case["mask"], level0_wh=(cfg["width"], cfg["height"]), downsamples=cfg["downsamples"], src_mag=cfg["mag"],
        tgt_mag=cfg["target_mag"], patch_size=cfg["patch_size"], step_size=cfg["step_size"],
        tissue_thresh=cfg["tissue_thresh"])
    assert coords.dtype == np.int32
    assert np.array_equal(coords, case["coords"])
    g = case["info"]["geometry"]
    assert [geom["level"], list(geom["read_wh"]), geom["patch_size_src"], geom["step_src"], geom["patch_size_level0"]] == g
    n = coords.shape[0]
    if n:
        stem = name
        first = coords_oracle.passport(stem, *coords[0].tolist(), cfg["mag"], cfg["target_mag"], n)
        last = coords_oracle.passport(stem, *coords[-1].tolist(), cfg["mag"], cfg["target_mag"], n)
        assert first == case["info"]["passport_first"] and last == case["info"]["passport_last"]


def test_oracle_contours_match_reference():
    for name, case in helpers.load_coords_cases().items():
        cfg = case["info"]["config"]
        tissue, holes = coords_oracle.mask_to_contours(case["mask"], tissue_area_thresh=cfg["tissue_thresh"])
        want_t, want_h = helpers.load_contour_case(name)
        assert len(tissue) == len(want_t) == case["info"]["n_tissue"], name
        for a, b in zip(tissue, want_t):
            assert np.array_equal(a.reshape(-1, 2), b), name
        assert [len(h) for h in holes] == case["info"]["n_holes"], name


# ----------------------------------------------------------------------------- primitive self-checks
def _random_blobs(rng, h, w, p=0.5, rounds=2):
    m = (rng.random((h, w)) > p).astype(np.float32)
    for _ in range(rounds):
        m = (m + np.roll(m, 1, 0) + np.roll(m, -1, 0) + np.roll(m, 1, 1) + np.roll(m, -1, 1)) / 5.0
    return (m > 0.5).astype(np.uint8) * 255


def test_find_contours_counts_match_region_labelling():
    """#outer borders = #8-connected components; #hole borders = #4-connected enclosed
    background regions (independent check with scipy.ndimage.label)."""
    from scipy import ndimage as ndi
    rng = np.random.default_rng(5)
    for trial in range(12):
        m = _random_blobs(rng, 60 + trial * 7, 90 - trial * 3, p=0.45 + 0.01 * trial)
        contours, hier = cv2r.findContours(m, cv2r.RETR_CCOMP, cv2r.CHAIN_APPROX_NONE)
        fg = m > 0
        _, n_comp = ndi.label(fg, structure=np.ones((3, 3)))
        padded = np.pad(~fg, 1, constant_values=True)
        lab, n_bg = ndi.label(padded)                       # 4-connected background incl. the frame
        n_holes = n_bg - 1
        if hier is None:
            assert n_comp == 0
            continue
        h = hier[0]
        assert int((h[:, 3] == -1).sum()) == n_comp
        assert int((h[:, 3] != -1).sum()) == n_holes
        # hierarchy is two-level and every hole's parent is an outer border
        for i in range(len(contours)):
            if h[i, 3] != -1:
                assert h[h[i, 3], 3] == -1
        # every contour point is a foreground pixel
        for c in contours:
            pts = c.reshape(-1, 2)
            assert fg[pts[:, 1], pts[:, 0]].all()


def test_point_polygon_test_forms_agree_and_are_rotation_invariant():
    rng = np.random.default_rng(9)
    m = _random_blobs(rng, 80, 100)
    contours, _ = cv2r.findContours(m, cv2r.RETR_CCOMP, cv2r.CHAIN_APPROX_NONE)
    big = max(contours, key=len)
    scaled = coords_oracle.scale_contours([big], 37.3, 12.9)[0]
    pts = np.stack([rng.integers(-50, 4000, 400), rng.integers(-50, 1200, 400)], -1)
    pts = np.concatenate([pts, scaled.reshape(-1, 2)[::7]], 0)       # include on-vertex queries
    vec = coords_oracle._pip_many(scaled, pts[:, 0], pts[:, 1])
    for (x, y), v in zip(pts.tolist(), vec.tolist()):
        s = cv2r.pointPolygonTest_scalar(scaled, (x, y))
        assert s == v == int(cv2r.pointPolygonTest(scaled, (x, y), False))
    rolled = np.roll(scaled, 17, axis=0)
    flipped = scaled[::-1].copy()
    assert np.array_equal(vec, coords_oracle._pip_many(rolled, pts[:, 0], pts[:, 1]))
    assert np.array_equal(vec, coords_oracle._pip_many(flipped, pts[:, 0], pts[:, 1]))


def test_point_polygon_test_vs_raster_truth():
    """Inside/outside against brute-force pixel membership for points strictly off the border."""
    m = np.zeros((40, 50), np.uint8)
    m[5:30, 8:40] = 255
    m[12:20, 15:25] = 0
    contours, hier = cv2r.findContours(m, cv2r.RETR_CCOMP, cv2r.CHAIN_APPROX_NONE)
    outer = [c for c, h in zip(contours, hier[0]) if h[3] == -1][0]
    hole = [c for c, h in zip(contours, hier[0]) if h[3] != -1][0]
    assert cv2r.contourArea(outer) == (24 * 31) and cv2r.boundingRect(outer) == (8, 5, 32, 25)
    assert cv2r.pointPolygonTest(outer, (20, 10), False) == 1
    assert cv2r.pointPolygonTest(outer, (8, 10), False) == 0          # on the border
    assert cv2r.pointPolygonTest(outer, (2, 2), False) == -1
    assert cv2r.pointPolygonTest(hole, (18, 15), False) == 1          # inside the hole polygon
    assert cv2r.pointPolygonTest(hole, (14, 15), False) == 0          # hole border runs on tissue pixels


def test_vit_oracle_matches_reference_extract_batch(golden_dir):
    from oracle import vit_oracle
    g = np.load(os.path.join(golden_dir, "extract_batch.npz"))
    x = vit_oracle.preprocess_center_crop(g["preproc_in"][None])
    assert np.array_equal(x[0].numpy(), g["preproc_out"])              # K1 op order, bit-exact
    model = vit_oracle.make_hf_vit(layers=2)
    sd = dict(model.state_dict())
    ns = (0, 1, 5, 32, 33)
    patches = helpers.golden_patches(ns)
    for n in ns:
        out = vit_oracle.extract_batch(sd, patches[n], heads=12, batch_size=32)
        assert out.dtype == np.float32 and out.shape == (n, 768)
        if n:
            ref = g[f"L2_n{n}_out"]
            assert np.linalg.norm(out - ref) / np.linalg.norm(ref) < 2e-5


def test_cv2_colour_restatement_known_answers():
    """Known-answer values of OpenCV's 8-bit RGB2GRAY / RGB2HSV (documented BT.601 weights and the
    255 * (max - min) / max saturation, both in fixed point with round-to-nearest)."""
    from oracle import cv2_restated as cv2r
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0], [128, 64, 64], [10, 200, 30],
                    [201, 200, 199], [199, 199, 199]]], np.uint8)
    assert cv2r.cvtColor_RGB2GRAY(px)[0].tolist() == [76, 150, 29, 255, 0, 83, 124, 200, 199]
    s, v = cv2r.cvtColor_RGB2HSV_sv(px)
    assert v[0].tolist() == [255, 255, 255, 255, 0, 128, 200, 201, 199]
    assert s[0].tolist() == [255, 255, 255, 0, 0, 128, 242, 3, 0]
    white = np.full((8, 8, 3), 240, np.uint8)
    black = np.full((8, 8, 3), 20, np.uint8)
    assert cv2r.is_white_patch(white, sat_thresh=15) and not cv2r.is_black_patch(white, rgb_thresh=50)
    assert cv2r.is_black_patch(black, rgb_thresh=50) and not cv2r.is_white_patch(black, sat_thresh=15)
    mixed = white.copy(); mixed[:3] = (200, 30, 30)             # 62.5 % white < 0.7
    assert not cv2r.is_white_patch(mixed, sat_thresh=15)
