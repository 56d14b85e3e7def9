"""Tile content filters on the device (reference: atlas_patch/utils/image.py:7-41).

``is_black_patch`` / ``is_white_patch`` keep the reference's names, arguments and decision rule
(``fraction >= min_fraction`` with the fraction computed as ``count / pixels`` in float64); the per-pixel
work (OpenCV's fixed-point RGB2GRAY / RGB2HSV) runs in ``ap_tile_content_counts``, bit-exactly.
``tile_content_flags`` is the batched form the extraction service uses.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib


def tile_content_counts(tiles: torch.Tensor, *, black_thresh: int, sat_thresh: int, value_thresh: int = 200) -> np.ndarray:
    """tiles: uint8 [n, h, w, 3] on the device -> int64 [n, 2] (black pixels, white pixels)."""
    assert tiles.is_cuda and tiles.dtype == torch.uint8 and tiles.is_contiguous() and tiles.dim() == 4
    n, h, w, _ = tiles.shape
    counts = torch.empty((n, 2), dtype=torch.int32, device=tiles.device)
    if n:
        with torch.cuda.device(tiles.device):
            _lib.check(_lib.load().ap_tile_content_counts(tiles.data_ptr(), n, h, w, int(black_thresh), int(sat_thresh),
                                                          int(value_thresh), counts.data_ptr(),
                                                          _lib.current_stream_ptr(tiles.device)), "ap_tile_content_counts")
    return counts.cpu().numpy().astype(np.int64)


def tile_content_flags(tiles: torch.Tensor, *, black_thresh: int, white_thresh: int, min_fraction: float = 0.7,
                       value_thresh: int = 200):
    """(is_black, is_white) bool arrays for a device batch, the reference's rule per tile."""
    n, h, w, _ = tiles.shape
    counts = tile_content_counts(tiles, black_thresh=black_thresh, sat_thresh=white_thresh, value_thresh=value_thresh)
    frac = counts.astype(np.float64) / float(h * w)
    return frac[:, 0] >= float(min_fraction), frac[:, 1] >= float(min_fraction)


def _one(patch: np.ndarray, device) -> torch.Tensor:
    arr = np.ascontiguousarray(patch)
    if arr.ndim != 3 or arr.shape[2] != 3 or arr.dtype != np.uint8:
        raise ValueError(f"patch must be HWC uint8 RGB, got shape {arr.shape} dtype {arr.dtype}")
    return torch.from_numpy(arr)[None].to(device)


def is_black_patch(patch: np.ndarray, rgb_thresh: int = 40, min_fraction: float = 0.7, *, device="cuda") -> bool:
    black, _ = tile_content_flags(_one(patch, device), black_thresh=rgb_thresh, white_thresh=0, min_fraction=min_fraction)
    return bool(black[0])


def is_white_patch(patch: np.ndarray, sat_thresh: int = 5, min_fraction: float = 0.7, value_thresh: int = 200, *,
                   device="cuda") -> bool:
    _, white = tile_content_flags(_one(patch, device), black_thresh=0, white_thresh=sat_thresh,
                                  min_fraction=min_fraction, value_thresh=value_thresh)
    return bool(white[0])
