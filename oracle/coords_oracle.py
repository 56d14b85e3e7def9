"""CPU oracle: tissue mask -> patch coordinates.  TEST INFRASTRUCTURE ONLY.

Restates, function by function, the reference's coordinate path (all citations are
into /root/reference/atlas_patch):

  optimal_level        core/wsi/iwsi.py:325-358
  prepare_geometry     services/extraction.py:44-64
  mask_to_contours     utils/contours.py:41-116
  scale_contours       utils/contours.py:119-131 (+ sx, sy from extraction.py:35-41)
  in_tissue            services/extraction.py:67-81 + utils/contours.py:22-38
  iter_coords          services/extraction.py:83-103 (fast-mode rows)
  passport             services/storage.py:387-392

Pinned by tests/golden (reference modules executed unmodified with cv2 replaced
by oracle/cv2_restated.py); the cv2 primitives themselves are parity-unpinned.
"""
from __future__ import annotations

from typing import Sequence

import numpy as np

from . import cv2_restated as cv2


# --------------------------------------------------------------------------- geometry
def optimal_level(downsamples: Sequence[float], target_ds: float):
    """iwsi.py:325-358."""
    downsamples = list(downsamples) or [1.0]
    for i, d in enumerate(downsamples):
        if abs(d - target_ds) < 0.01:
            return i, 1.0
    if target_ds >= downsamples[0]:
        best_i, best_d = 0, downsamples[0]
        for i, d in enumerate(downsamples):
            if d <= target_ds:
                best_i, best_d = i, d
            else:
                break
        return best_i, target_ds / best_d
    for i, d in enumerate(downsamples):
        if d >= target_ds:
            return i, d / target_ds
    raise ValueError(f"No level for target downsample {target_ds}")


def prepare_geometry(downsamples, src_mag, tgt_mag, patch_size, step_size=None):
    """extraction.py:44-64 -> (level, (read_w, read_h), patch_size_src, step_src, patch_size_level0)."""
    if src_mag is None:
        raise ValueError("WSI base magnification is required for patch extraction.")
    if int(tgt_mag) > int(src_mag):
        raise ValueError(f"Requested magnification {tgt_mag}x exceeds available {src_mag}x.")
    desired = float(src_mag) / float(tgt_mag)
    level, _ = optimal_level(downsamples, desired)
    level_ds = float((list(downsamples) or [1.0])[level])
    patch_size_src = int(round(patch_size * desired))
    step_src = int(round((step_size or patch_size) * desired))
    patch_size_level0 = int(patch_size * int(src_mag) // int(tgt_mag))
    read_w = max(1, int(round(patch_size_src / level_ds)))
    return level, (read_w, read_w), patch_size_src, step_src, patch_size_level0


# --------------------------------------------------------------------------- contours
def mask_to_contours(mask: np.ndarray, *, tissue_area_thresh: float = 0.01,
                     a_h: int = 16, max_n_holes: int = 10):
    """contours.py:41-116."""
    mask_u8 = (mask > 0.5).astype(np.uint8) * 255
    contours, hierarchy = cv2.findContours(mask_u8, cv2.RETR_CCOMP, cv2.CHAIN_APPROX_NONE)
    if hierarchy is None or len(contours) == 0:
        return [], []
    hierarchy = hierarchy[0]
    H, W = mask.shape[:2]
    min_area = tissue_area_thresh * float(H * W)
    hole_thr = float(a_h)

    tissue_idx = []
    holes_by_parent: dict[int, list] = {}
    for i, cont in enumerate(contours):
        area = cv2.contourArea(cont)
        parent = int(hierarchy[i][3])
        if parent == -1:
            if area >= min_area:
                tissue_idx.append(i)
        elif area >= hole_thr:
            holes_by_parent.setdefault(parent, []).append(cont)

    all_holes = [h for hs in holes_by_parent.values() for h in hs]
    if max_n_holes > 0 and len(all_holes) > max_n_holes:
        ranked = sorted(all_holes, key=cv2.contourArea, reverse=True)   # stable
        allowed = set(map(id, ranked[:max_n_holes]))
        for parent, hs in list(holes_by_parent.items()):
            holes_by_parent[parent] = [h for h in hs if id(h) in allowed]

    tissue = [contours[i] for i in tissue_idx]
    holes = [list(holes_by_parent.get(i, [])) for i in tissue_idx]
    return tissue, holes


def scale_contours(contours, sx: float, sy: float):
    """contours.py:119-131: trunc(fl32(fl32(x) * fl32(s))) per coordinate."""
    out = []
    for c in contours:
        c = c.astype(np.float32)
        c[:, :, 0] *= sx
        c[:, :, 1] *= sy
        out.append(c.astype(np.int32))
    return out


# --------------------------------------------------------------------------- point in polygon
def _pip_many(contour: np.ndarray, px: np.ndarray, py: np.ndarray, chunk: int = 1024) -> np.ndarray:
    """Vectorised integer pointPolygonTest over many points -> int8 in {-1, 0, +1}."""
    pts = contour.reshape(-1, 2).astype(np.int64)
    vx, vy = pts[:, 0][None, :], pts[:, 1][None, :]
    v0x, v0y = np.roll(pts[:, 0], 1)[None, :], np.roll(pts[:, 1], 1)[None, :]
    out = np.empty(px.shape[0], dtype=np.int8)
    for s in range(0, px.shape[0], chunk):
        x = px[s:s + chunk].astype(np.int64)[:, None]
        y = py[s:s + chunk].astype(np.int64)[:, None]
        skip = ((v0y <= y) & (vy <= y)) | ((v0y > y) & (vy > y)) | ((v0x < x) & (vx < x))
        on_skip = skip & (vy == y) & (
            (vx == x) | ((v0y == y) & (((v0x <= x) & (x <= vx)) | ((vx <= x) & (x <= v0x)))))
        dist = (y - v0y) * (vx - v0x) - (x - v0x) * (vy - v0y)
        live = ~skip
        on_edge = on_skip.any(axis=1) | (live & (dist == 0)).any(axis=1)
        dist = np.where(vy < v0y, -dist, dist)
        odd = (np.count_nonzero(live & (dist > 0), axis=1) & 1) == 1
        out[s:s + chunk] = np.where(on_edge, 0, np.where(odd, 1, -1))
    return out


def iter_coords(tissue_contours, holes_contours, *, level, read_wh, patch_size_src, step_src):
    """extraction.py:83-103 fast-mode rows: int32 [N, 5] = (x, y, read_w, read_h, level).

    Per tissue contour (in order): grid over its bounding rect, row-major, no
    clipping to the slide and no de-duplication across contours; a cell is kept
    iff its centre is not strictly inside any hole and any of the four diagonal
    probes at +-(patch//2)//2 lies inside or on the contour.
    """
    rows = []
    ps = int(patch_size_src)
    half = ps // 2
    shift = int(half * 0.5)
    for contour, holes in zip(tissue_contours, holes_contours):
        x0, y0, ww, hh = cv2.boundingRect(contour)
        xs = np.arange(x0, x0 + ww, step_src, dtype=np.int64)
        ys = np.arange(y0, y0 + hh, step_src, dtype=np.int64)
        if xs.size == 0 or ys.size == 0:
            continue
        gx, gy = np.meshgrid(xs, ys)          # row-major: y outer, x inner
        gx = gx.ravel()
        gy = gy.ravel()
        cx = gx + half
        cy = gy + half
        alive = np.ones(gx.shape[0], dtype=bool)
        for hole in holes:
            idx = np.flatnonzero(alive)
            if idx.size == 0:
                break
            inside = _pip_many(hole, cx[idx], cy[idx]) > 0
            alive[idx[inside]] = False
        keep = np.zeros(gx.shape[0], dtype=bool)
        if shift > 0:
            probes = ((-shift, -shift), (shift, shift), (shift, -shift), (-shift, shift))
        else:
            probes = ((0, 0),)
        for dx, dy in probes:
            idx = np.flatnonzero(alive & ~keep)
            if idx.size == 0:
                break
            ok = _pip_many(contour, cx[idx] + dx, cy[idx] + dy) >= 0
            keep[idx[ok]] = True
        sel = np.flatnonzero(keep)
        block = np.empty((sel.size, 5), dtype=np.int64)
        block[:, 0] = gx[sel]
        block[:, 1] = gy[sel]
        block[:, 2] = read_wh[0]
        block[:, 3] = read_wh[1]
        block[:, 4] = level
        rows.append(block)
    if not rows:
        return np.empty((0, 5), dtype=np.int32)
    return np.concatenate(rows, axis=0).astype(np.int32)


def coords_from_mask(mask: np.ndarray, *, level0_wh, downsamples, src_mag, tgt_mag,
                     patch_size, step_size=None, tissue_thresh=0.01):
    """PatchExtractionService.extract's coordinate content (extraction.py:30-42,131-197)."""
    tissue_t, holes_t = mask_to_contours(mask, tissue_area_thresh=tissue_thresh)
    W, H = level0_wh
    mh, mw = mask.shape[:2]
    sx = W / float(mw)
    sy = H / float(mh)
    tissue = scale_contours(tissue_t, sx, sy)
    holes = [scale_contours(hs, sx, sy) for hs in holes_t]
    level, read_wh, ps_src, step_src, ps_l0 = prepare_geometry(
        downsamples, src_mag, tgt_mag, patch_size, step_size)
    coords = iter_coords(tissue, holes, level=level, read_wh=read_wh,
                         patch_size_src=ps_src, step_src=step_src)
    return coords, dict(level=level, read_wh=read_wh, patch_size_src=ps_src,
                        step_src=step_src, patch_size_level0=ps_l0)


def passport(stem, x, y, rw, rh, lv, level0_mag, target_mag, total):
    """storage.py:387-392."""
    mag_val = level0_mag if level0_mag else "na"
    tgt_val = target_mag if target_mag else "na"
    return f"{stem}__x{x}_y{y}_rw{rw}_rh{rh}_lv{lv}_mag{mag_val}_tmag{tgt_val}_total{total}"
