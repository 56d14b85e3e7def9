"""Tissue mask -> contours -> patch coordinates through the C ABI.

Host-side mirror of /root/reference/atlas_patch/utils/contours.py (``mask_to_contours``,
``scale_contours``) and of the grid scan in services/extraction.py:67-128.  All arithmetic
happens in ``libatlaspatch_hip.so`` (threshold + grid point-in-polygon kernels on the GPU,
border following in host C++); there is no Python/NumPy fallback.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Optional, Sequence

import numpy as np

from .. import _lib

DEFAULT_FILTER_PARAMS = {"a_h": 16, "max_n_holes": 10}

_tls = threading.local()


def worker_stream() -> int:
    """HIP stream of the coordinate kernels for the calling thread.  The main thread stays on stream 0 (ordered with everything
    else it launches).  A worker thread (the runner's coordinate pool) gets its own NON-blocking stream, created once per
    thread: on the legacy stream its threshold / grid kernels and copies queued behind the segmenter's graph replays -- a
    worker's stream synchronisation then waited for forwards it has nothing to do with (18 ms of thread time per slide for
    2 ms of work).  The inputs are host arrays and every call ends with its own synchronisation, so no cross-stream ordering
    is needed."""
    if threading.current_thread() is threading.main_thread():
        return 0
    st = getattr(_tls, "stream", None)
    if st is None:
        import torch
        if not torch.cuda.is_available():
            return 0
        st = _tls.stream = torch.cuda.Stream()          # hipStreamNonBlocking
    return int(st.cuda_stream)


class DeviceContours:
    """Owns an ``ap_contours`` handle: filtered, ordered tissue contours + holes, with their
    level-0 scaled copies (``scale_contours`` semantics)."""

    def __init__(self, mask: np.ndarray, *, tissue_area_thresh: float = 0.01,
                 filter_params: Optional[dict] = None, sx: float = 1.0, sy: float = 1.0,
                 stream: Optional[int] = None) -> None:
        self.lib = _lib.load()
        self._stream = worker_stream() if stream is None else int(stream)
        params = dict(DEFAULT_FILTER_PARAMS if filter_params is None else filter_params)
        mask = np.ascontiguousarray(mask, dtype=np.float32)
        if mask.ndim != 2:
            raise ValueError(f"mask must be 2-D, got shape {mask.shape}")
        self.shape = mask.shape
        handle = C.c_void_p()
        _lib.check(self.lib.ap_contours_from_mask(
            mask.ctypes.data_as(C.c_void_p), mask.shape[0], mask.shape[1], float(tissue_area_thresh),
            int(params.get("a_h", 0)), int(params.get("max_n_holes", 0)), float(sx), float(sy),
            C.byref(handle), C.c_void_p(self._stream)), "ap_contours_from_mask")
        self._handle = handle

    def __len__(self) -> int:
        return int(self.lib.ap_contours_count(self._handle))

    def num_holes(self, i: int) -> int:
        return int(self.lib.ap_contours_num_holes(self._handle, i))

    def points(self, i: int, hole: int = -1, scaled: bool = False) -> np.ndarray:
        n = self.lib.ap_contours_points(self._handle, i, hole, 1 if scaled else 0, None, 0)
        if n < 0:
            _lib.check(n, "ap_contours_points")
        out = np.empty((n, 1, 2), dtype=np.int32)
        if n:
            got = self.lib.ap_contours_points(self._handle, i, hole, 1 if scaled else 0,
                                              out.ctypes.data_as(C.c_void_p), n)
            if got < 0:
                _lib.check(got, "ap_contours_points")
        return out

    def as_lists(self, scaled: bool = False):
        tissue = [self.points(i, -1, scaled) for i in range(len(self))]
        holes = [[self.points(i, h, scaled) for h in range(self.num_holes(i))] for i in range(len(self))]
        return tissue, holes

    def grid_coords(self, *, patch_size_src: int, step_src: int, read_wh: Sequence[int], level: int,
                    stream: Optional[int] = None) -> np.ndarray:
        """int32 [N, 5] rows (x, y, read_w, read_h, level) in the reference's order."""
        stream = self._stream if stream is None else int(stream)
        total = C.c_size_t(0)
        cap = 1 << 16
        while True:
            out = np.empty((cap, 5), dtype=np.int32)
            rc = self.lib.ap_grid_coords(self._handle, int(patch_size_src), int(step_src), int(read_wh[0]),
                                         int(read_wh[1]), int(level), out.ctypes.data_as(C.c_void_p), cap,
                                         C.byref(total), C.c_void_p(stream))
            if rc == _lib.AP_ERR_CAPACITY:
                cap = int(total.value)
                continue
            _lib.check(rc, "ap_grid_coords")
            return out[: int(total.value)].copy()

    def close(self) -> None:
        if getattr(self, "_handle", None) is not None:
            self.lib.ap_contours_destroy(self._handle)
            self._handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def mask_to_contours(mask: np.ndarray, *, tissue_area_thresh: float = 0.01,
                     filter_params: Optional[dict] = None):
    """(tissue_contours, holes_per_tissue) as lists of int32 [n, 1, 2] arrays (contours.py:41-116)."""
    dc = DeviceContours(mask, tissue_area_thresh=tissue_area_thresh, filter_params=filter_params)
    try:
        return dc.as_lists(scaled=False)
    finally:
        dc.close()


def scale_contours(contours, sx: float, sy: float):
    """``scale_contours`` of the reference (contours.py:119-131) for host-side consumers (the visualisation overlay):
    ``c.astype(float32); c[..., 0] *= sx; c[..., 1] *= sy; c.astype(int32)`` -- float32 multiply, truncation toward zero.
    The coordinate path itself scales inside ``ap_contours_from_mask`` (same arithmetic, golden G3)."""
    out = []
    for c in contours:
        s = np.asarray(c).astype(np.float32)
        s[..., 0] *= sx
        s[..., 1] *= sy
        out.append(s.astype(np.int32))
    return out
