"""Image resampling on the device: the two resamplers the reference's hot path calls.

``cv2_resize_device`` / ``cv2_resize_array`` = ``cv2.resize`` for uint8 RGB (``ap_cv2_resize_u8``: OpenCV's 8-bit
arithmetic restated, see csrc/cv2resize.hip) -- the per-tile ``cv2.resize(patch, (ps, ps))`` of
/root/reference/atlas_patch/services/feature_embedding.py:94-95 and services/extraction.py:112-113, and the thumbnail
resize of core/wsi/iwsi.py:305-321 (INTER_AREA when shrinking, INTER_CUBIC when enlarging).  OpenCV is not
available in this image: PARITY UNPINNED against cv2 itself, bit-exact against ``oracle/cv2_resize.py``.

``DeviceResampler`` = ``PIL.Image.resize`` (bit-identical; Pillow is present) for the encoders' own transforms.
There is no host fallback: without a HIP device these raise.
"""
from __future__ import annotations

import numpy as np

INTER_LINEAR, INTER_CUBIC, INTER_AREA = 1, 2, 3      # cv2's constants (= AP_CV_INTER_*)


def thumbnail_interpolation(in_hw, out_wh, policy: str = "optimise") -> int:
    """The interpolation iwsi.py:305-319 picks: "optimise" = AREA when either axis shrinks, else CUBIC."""
    h, w = int(in_hw[0]), int(in_hw[1])
    out_w, out_h = int(out_wh[0]), int(out_wh[1])
    if policy == "optimise":
        return INTER_AREA if (out_w < w or out_h < h) else INTER_CUBIC
    return {"area": INTER_AREA, "cubic": INTER_CUBIC, "linear": INTER_LINEAR}.get(policy, INTER_LINEAR)


def cv2_resize_device(tiles, out_wh, interpolation: int = INTER_LINEAR, *, out=None, flags: int = 0):
    """``cv2.resize(tile, (out_w, out_h), interpolation=...)`` on a device batch: uint8 [n, h, w, 3] -> [n, oh, ow, 3]."""
    import torch
    from .. import _lib
    assert tiles.is_cuda and tiles.dtype == torch.uint8 and tiles.dim() == 4 and tiles.shape[3] == 3 and tiles.is_contiguous()
    n, h, w = int(tiles.shape[0]), int(tiles.shape[1]), int(tiles.shape[2])
    ow, oh = int(out_wh[0]), int(out_wh[1])
    if out is None:
        out = torch.empty((n, oh, ow, 3), dtype=torch.uint8, device=tiles.device)
    assert out.is_cuda and out.is_contiguous() and tuple(out.shape) == (n, oh, ow, 3)
    with torch.cuda.device(tiles.device):
        _lib.check(_lib.load().ap_cv2_resize_u8(tiles.data_ptr(), n, h, w, out.data_ptr(), oh, ow, int(interpolation),
                                                int(flags), _lib.current_stream_ptr(tiles.device)), "ap_cv2_resize_u8")
    return out


def cv2_resize_array(arr: np.ndarray, out_wh, interpolation: int = INTER_LINEAR, *, device=None) -> np.ndarray:
    """Host uint8 [h, w, 3] -> host uint8 [oh, ow, 3] through the device kernel (thumbnails, single tiles)."""
    import torch
    from .. import _lib
    if not torch.cuda.is_available():
        raise _lib.HipLibraryError("cv2-exact resize runs on the HIP device; no HIP device is available "
                                   "(there is no CPU fallback)")
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    src = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.uint8)).to(dev)
    return cv2_resize_device(src[None], out_wh, interpolation)[0].cpu().numpy()


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
