"""Patch geometry at a target magnification (reference: services/extraction.py:44-64).

Pure host arithmetic (float64 + Python's banker's ``round``); must stay bit-identical because
it fixes ``level``, ``read_w`` and the grid stride of every coords row.
"""
from __future__ import annotations

from typing import NamedTuple

from ..core.wsi.iwsi import IWSI


class PatchGeometry(NamedTuple):
    level: int
    read_wh: tuple[int, int]
    patch_size_src: int      # patch footprint in level-0 pixels
    step_src: int            # grid stride in level-0 pixels
    patch_size_level0: int


def prepare_geometry(wsi: IWSI, *, patch_size: int, step_size: int | None,
                     target_magnification: int) -> PatchGeometry:
    src_mag = wsi.mag
    if src_mag is None:
        raise ValueError("WSI base magnification is required for patch extraction.")
    if int(target_magnification) > int(src_mag):
        raise ValueError(f"Requested magnification {target_magnification}x exceeds available {src_mag}x.")
    desired_ds = float(src_mag) / float(target_magnification)
    level, _ = wsi.optimal_level(desired_ds)
    level_ds = float((wsi.ds or [1.0])[level])
    footprint = int(round(patch_size * desired_ds))
    stride = int(round((step_size or patch_size) * desired_ds))
    at_level0 = int(patch_size * int(src_mag) // int(target_magnification))
    read = max(1, int(round(footprint / level_ds)))
    return PatchGeometry(level, (read, read), footprint, stride, at_level0)
