"""Image resampling used outside the per-tile hot loop.

``resize_area_or_cubic`` stands in for the ``cv2.resize`` call of
/root/reference/atlas_patch/core/wsi/iwsi.py:305-321 (INTER_AREA when shrinking, INTER_CUBIC when
enlarging).  OpenCV is not available in this image: PARITY UNPINNED.  The exact-integer-ratio
area average (what pyramid levels produce) is implemented exactly; other ratios use Pillow's
BOX / BICUBIC filters, which differ from OpenCV's by rounding.  Synthetic slides and slides whose
pyramid holds the 1.25x level never reach this function.
"""
from __future__ import annotations

import numpy as np
from PIL import Image


def resize_area_or_cubic(arr: np.ndarray, size_wh, interpolation: str = "optimise") -> np.ndarray:
    out_w, out_h = int(size_wh[0]), int(size_wh[1])
    h, w = arr.shape[:2]
    shrink = out_w < w or out_h < h
    if interpolation == "linear":
        return np.asarray(Image.fromarray(arr).resize((out_w, out_h), Image.Resampling.BILINEAR))
    if (interpolation in ("optimise", "area")) and shrink:
        if w % out_w == 0 and h % out_h == 0:
            fx, fy = w // out_w, h // out_h
            acc = arr.reshape(out_h, fy, out_w, fx, -1).astype(np.float64).mean(axis=(1, 3))
            return np.clip(np.rint(acc), 0, 255).astype(np.uint8).reshape(out_h, out_w, *arr.shape[2:])
        return np.asarray(Image.fromarray(arr).resize((out_w, out_h), Image.Resampling.BOX))
    return np.asarray(Image.fromarray(arr).resize((out_w, out_h), Image.Resampling.BICUBIC))


# ----------------------------------------------------------------------------- Pillow resampling tables
# libImaging/Resample.c (8 bits per channel): precompute_coeffs + normalize_coeffs_8bpc, in the same double
# arithmetic and operation order, so the device kernel (ap_resample_u8) reproduces PIL.Image.resize bit for bit.
_PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _bilinear(x: float) -> float:
    if x < 0.0:
        x = -x
    return 1.0 - x if x < 1.0 else 0.0


_FILTERS = {"bicubic": (_bicubic, 2.0), "bilinear": (_bilinear, 1.0)}


def pillow_resample_tables(in_size: int, out_size: int, filter_name: str = "bicubic"):
    """(bounds int32 [out, 2], coeffs int32 [out, ksize], ksize) for one axis of ``Image.resize``."""
    import math
    fn, support0 = _FILTERS[filter_name]
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    coeffs = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [fn((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in k:
            ww += v
        for x in range(xmax):
            v = k[x] / ww if ww != 0.0 else k[x]
            coeffs[xx, x] = int(-0.5 + v * (1 << _PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << _PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, coeffs, ksize


class DeviceResampler:
    """``PIL.Image.resize((ow, oh), resample)`` for device batches of uint8 HWC tiles (bit-identical)."""

    def __init__(self, in_hw, out_hw, filter_name: str, device) -> None:
        import torch
        self.in_hw, self.out_hw = (int(in_hw[0]), int(in_hw[1])), (int(out_hw[0]), int(out_hw[1]))
        self.device = torch.device(device)
        bx, kx, self.ksx = pillow_resample_tables(self.in_hw[1], self.out_hw[1], filter_name)
        by, ky, self.ksy = pillow_resample_tables(self.in_hw[0], self.out_hw[0], filter_name)
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        self.bx, self.kx, self.by, self.ky = to(bx), to(kx), to(by), to(ky)
        self._tmp = None

    def __call__(self, tiles):
        import torch
        from .. import _lib
        assert tiles.is_cuda and tiles.dtype == torch.uint8 and tiles.is_contiguous() and tuple(tiles.shape[1:3]) == self.in_hw
        n = tiles.shape[0]
        oh, ow = self.out_hw
        need = n * self.in_hw[0] * ow * 3
        if self._tmp is None or self._tmp.numel() < need:
            self._tmp = torch.empty(need, dtype=torch.uint8, device=self.device)
        out = torch.empty((n, oh, ow, 3), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().ap_resample_u8(tiles.data_ptr(), n, self.in_hw[0], self.in_hw[1], out.data_ptr(), oh, ow,
                                                  self.bx.data_ptr(), self.kx.data_ptr(), self.ksx, self.by.data_ptr(),
                                                  self.ky.data_ptr(), self.ksy, self._tmp.data_ptr(),
                                                  _lib.current_stream_ptr(self.device)), "ap_resample_u8")
        return out
