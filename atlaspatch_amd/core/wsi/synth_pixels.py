"""Deterministic synthetic whole-slide content.

A synthetic slide is a pure function of ``(seed, width, height)``: a near-white
background with a union of axis-aligned ellipses of pink/purple "tissue".  All
arithmetic is integer (a 32-bit mixing hash and int64 ellipse tests in units of
16 level-0 pixels) so that the NumPy implementation here and the HIP tile
synthesis kernel (csrc/synth.hip) produce identical bytes.

The same ellipses rasterised on the <=1024 thumbnail grid give the analytic
tissue mask used when SAM2 weights are unavailable.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

UNIT_SHIFT = 4           # ellipse geometry lives on a 16-pixel lattice
_M32 = np.uint64(0xFFFFFFFF)


@dataclass(frozen=True)
class SynthSpec:
    width: int
    height: int
    seed: int = 1234
    n_ellipses: int = 12
    mag: int = 20
    mpp: float = 0.5
    downsamples: tuple = (1.0, 4.0, 16.0)

    def ellipses(self) -> np.ndarray:
        """int64 [K, 4] = (cx, cy, a, b) in 16-pixel units, seeded."""
        return make_ellipses(self.width, self.height, self.seed, self.n_ellipses)


def mix32(a: np.ndarray) -> np.ndarray:
    """32-bit integer finaliser (lowbias32); uint64 carrier, values stay < 2**32."""
    a = a & _M32
    a ^= a >> np.uint64(16)
    a = (a * np.uint64(0x7FEB352D)) & _M32
    a ^= a >> np.uint64(15)
    a = (a * np.uint64(0x846CA68B)) & _M32
    a ^= a >> np.uint64(16)
    return a


def make_ellipses(width: int, height: int, seed: int, k: int) -> np.ndarray:
    uw = max(1, width >> UNIT_SHIFT)
    uh = max(1, height >> UNIT_SHIFT)
    idx = np.arange(k, dtype=np.uint64)
    base = mix32(idx * np.uint64(4) + np.uint64(seed & 0xFFFFFFFF) * np.uint64(0x9E3779B1))
    r0 = mix32(base + np.uint64(1))
    r1 = mix32(base + np.uint64(2))
    r2 = mix32(base + np.uint64(3))
    r3 = mix32(base + np.uint64(4))
    out = np.empty((k, 4), dtype=np.int64)
    # centres in the middle 70 % of the slide, radii 6..22 % of the short side
    out[:, 0] = (uw * 15) // 100 + (r0 % np.uint64(max(1, (uw * 70) // 100))).astype(np.int64)
    out[:, 1] = (uh * 15) // 100 + (r1 % np.uint64(max(1, (uh * 70) // 100))).astype(np.int64)
    short = min(uw, uh)
    lo = max(1, (short * 6) // 100)
    span = max(1, (short * 16) // 100)
    out[:, 2] = lo + (r2 % np.uint64(span)).astype(np.int64)
    out[:, 3] = lo + (r3 % np.uint64(span)).astype(np.int64)
    return out


def inside_any(ux: np.ndarray, uy: np.ndarray, ell: np.ndarray) -> np.ndarray:
    """Point (in 16-px units, int64 arrays) inside the union of ellipses."""
    hit = np.zeros(np.broadcast(ux, uy).shape, dtype=bool)
    for cx, cy, a, b in ell.tolist():
        dx = (ux - cx) * b
        dy = (uy - cy) * a
        hit |= (dx * dx + dy * dy) <= (a * b) * (a * b)
    return hit


def render_region(spec: SynthSpec, x0: int, y0: int, w: int, h: int, level: int = 0) -> np.ndarray:
    """RGB uint8 [h, w, 3] of the region whose top-left is LEVEL-0 (x0, y0), read at ``level``.

    Pixels outside the slide are black (what OpenSlide returns after RGBA->RGB,
    /root/reference/atlas_patch/core/wsi/openslide_wsi.py:198).
    """
    ds = int(round(spec.downsamples[level]))
    xs = x0 + np.arange(w, dtype=np.int64) * ds
    ys = y0 + np.arange(h, dtype=np.int64) * ds
    gx, gy = np.meshgrid(xs, ys)
    tissue = inside_any(gx >> UNIT_SHIFT, gy >> UNIT_SHIFT, spec.ellipses())
    key = (gx.astype(np.uint64) * np.uint64(0x9E3779B1)
           + gy.astype(np.uint64) * np.uint64(0x85EBCA77)
           + np.uint64(spec.seed & 0xFFFFFFFF) + np.uint64(level) * np.uint64(0xC2B2AE3D))
    hsh = mix32(key)
    n0 = (hsh & np.uint64(0xFF)).astype(np.int64)
    n1 = ((hsh >> np.uint64(8)) & np.uint64(0xFF)).astype(np.int64)
    n2 = ((hsh >> np.uint64(16)) & np.uint64(0xFF)).astype(np.int64)
    out = np.empty((h, w, 3), dtype=np.uint8)
    bg = 236 + (n0 & 7)
    out[..., 0] = np.where(tissue, 168 + (n0 >> 2), bg)
    out[..., 1] = np.where(tissue, 72 + (n1 >> 1), bg)
    out[..., 2] = np.where(tissue, 136 + (n2 >> 2), bg)
    oob = (gx < 0) | (gy < 0) | (gx >= spec.width) | (gy >= spec.height)
    out[oob] = 0
    return out


def analytic_mask(spec: SynthSpec, thumb_max: int = 1024) -> np.ndarray:
    """float32 {0,1} mask on the thumbnail grid the reference would segment.

    Thumbnail size follows PIL's ``Image.thumbnail((1024, 1024))`` aspect rule
    applied to the 1.25x power image (/root/reference/atlas_patch/services/
    segmentation.py:202-206): the long side becomes ``thumb_max``.
    """
    ds_target = spec.mag / 1.25
    pw = max(1, int(round(spec.width / ds_target)))
    ph = max(1, int(round(spec.height / ds_target)))
    mw, mh = thumbnail_size(pw, ph, thumb_max)
    # sample each mask pixel at the level-0 coordinate of its centre
    cx = ((np.arange(mw, dtype=np.int64) * 2 + 1) * spec.width) // (2 * mw)
    cy = ((np.arange(mh, dtype=np.int64) * 2 + 1) * spec.height) // (2 * mh)
    # 1-D coordinate vectors broadcast inside inside_any: the per-axis squares are computed on W + H values instead of W x H
    return inside_any((cx >> UNIT_SHIFT)[None, :], (cy >> UNIT_SHIFT)[:, None], spec.ellipses()).astype(np.float32)


def thumbnail_size(w: int, h: int, max_side: int) -> tuple[int, int]:
    """Output size of ``PIL.Image.thumbnail((max_side, max_side))`` (Pillow >= 7 rule)."""
    if w <= max_side and h <= max_side:
        return w, h
    import math

    def _round_aspect(number, key):
        return max(min(math.floor(number), math.ceil(number), key=key), 1)

    x, y = max_side, max_side
    aspect = w / h
    if x / y >= aspect:
        x = _round_aspect(y * aspect, key=lambda n: abs(aspect - n / y))
    else:
        y = _round_aspect(x / aspect, key=lambda n: 0 if n == 0 else abs(aspect - x / n))
    return int(x), int(y)
