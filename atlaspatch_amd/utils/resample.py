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


def _bicubic_v(x: np.ndarray) -> np.ndarray:
    a = -0.5
    x = np.abs(x)
    near = ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    far = (((x - 5) * x + 8) * x - 4) * a
    return np.where(x < 1.0, near, np.where(x < 2.0, far, 0.0))


def _bilinear_v(x: np.ndarray) -> np.ndarray:
    x = np.abs(x)
    return np.where(x < 1.0, 1.0 - x, 0.0)


_FILTERS_V = {"bicubic": _bicubic_v, "bilinear": _bilinear_v}
_TABLE_CACHE: dict = {}


def pillow_resample_tables(in_size: int, out_size: int, filter_name: str = "bicubic", box=None):
    """(bounds int32 [out, 2], coeffs int32 [out, ksize], ksize) for one axis of ``Image.resize``.
    ``box`` = (in0, in1): the source interval of ``resize(..., box=...)`` along this axis (C floats in Pillow: the values
    are rounded to float32 first, their difference is taken in float32, everything after that is double).
    Vectorised over the output positions with the SAME double operations in the same order as precompute_coeffs (the
    window sum is a sequential cumulative sum, not a pairwise one); ``_pillow_resample_tables_scalar`` is the loop form
    the tests compare it with.  Cached per argument tuple (the arrays are read-only)."""
    import math
    key = (int(in_size), int(out_size), filter_name, None if box is None else (float(box[0]), float(box[1])))
    hit = _TABLE_CACHE.get(key)
    if hit is not None:
        return hit
    fn, support0 = _FILTERS_V[filter_name], _FILTERS[filter_name][1]
    in0, in1 = (np.float32(0.0), np.float32(in_size)) if box is None else (np.float32(box[0]), np.float32(box[1]))
    scale = filterscale = float(np.float32(in1 - in0)) / out_size
    in0 = float(in0)
    if filterscale < 1.0:
        filterscale = 1.0
    support = support0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    xx = np.arange(out_size, dtype=np.float64)
    center = in0 + (xx + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)          # (int) truncates toward zero; operands here are > -1
    xmin = np.where(center - support + 0.5 < 0, 0, xmin)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    taps = np.arange(ksize, dtype=np.int64)[None, :]
    live = taps < xmax[:, None]
    arg = ((taps + xmin[:, None]).astype(np.float64) - center[:, None] + 0.5) * ss
    k = np.where(live, fn(arg), 0.0)
    ww = np.cumsum(k, axis=1)[:, -1]                                          # sequential, like the C loop (zeros add nothing)
    v = np.where(ww[:, None] != 0.0, k / np.where(ww == 0.0, 1.0, ww)[:, None], k)
    fixed = np.where(v < 0, np.trunc(-0.5 + v * (1 << _PRECISION_BITS)), np.trunc(0.5 + v * (1 << _PRECISION_BITS)))
    coeffs = np.where(live, fixed, 0.0).astype(np.int32)
    bounds = np.stack([xmin, xmax], 1).astype(np.int32)
    coeffs.setflags(write=False); bounds.setflags(write=False)
    if len(_TABLE_CACHE) > 256:
        _TABLE_CACHE.clear()
    _TABLE_CACHE[key] = (bounds, coeffs, ksize)
    return bounds, coeffs, ksize


def _pillow_resample_tables_scalar(in_size: int, out_size: int, filter_name: str = "bicubic", box=None):
    """Loop form of ``pillow_resample_tables`` (libImaging/Resample.c precompute_coeffs + normalize_coeffs_8bpc line by line)."""
    import math
    fn, support0 = _FILTERS[filter_name]
    in0, in1 = (np.float32(0.0), np.float32(in_size)) if box is None else (np.float32(box[0]), np.float32(box[1]))
    scale = filterscale = float(np.float32(in1 - in0)) / out_size
    in0 = float(in0)
    if filterscale < 1.0:
        filterscale = 1.0
    support = support0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    coeffs = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
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

    def __init__(self, in_hw, out_hw, filter_name: str, device, box=None) -> None:
        """``box`` = (x0, y0, x1, y1) floats: the source region of ``resize(size, resample, box=box)``."""
        import torch
        self.in_hw, self.out_hw = (int(in_hw[0]), int(in_hw[1])), (int(out_hw[0]), int(out_hw[1]))
        self.device = torch.device(device)
        bx, kx, self.ksx = pillow_resample_tables(self.in_hw[1], self.out_hw[1], filter_name,
                                                  None if box is None else (box[0], box[2]))
        by, ky, self.ksy = pillow_resample_tables(self.in_hw[0], self.out_hw[0], filter_name,
                                                  None if box is None else (box[1], box[3]))
        to = lambda a: torch.from_numpy(np.array(a, order="C")).to(self.device)      # a writable copy (tables may be cached views)
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


# ----------------------------------------------------------------------------- Pillow thumbnail (reduce + resize) on the device
import threading

_FILTER_SUPPORT = {"bicubic": 2.0, "bilinear": 1.0}
_RESAMPLER_CACHE: dict = {}          # (in shape, out size, filter, box, device) -> DeviceResampler (its scratch is per instance:
_RESAMPLER_LOCK = threading.Lock()   #  calls are serialised by the lock; they only enqueue)


def pillow_thumbnail_plan(size_wh, max_wh, filter_name: str = "bicubic", reducing_gap: float = 2.0):
    """What ``Image.thumbnail(max_wh, resample, reducing_gap)`` does to an image of ``size_wh`` (Pillow 12
    Image.thumbnail / Image.resize / Image._get_safe_box, restated from the installed package's Python source):
    ``None`` when the image already fits, else ``(final_wh, reduce)`` with ``reduce = None`` or
    ``((fx, fy), reduce_box, resize_box)``: first ``reduce((fx, fy), box=reduce_box)``, then
    ``resize(final_wh, resample, box=resize_box)`` on the reduced image."""
    import math
    width, height = int(size_wh[0]), int(size_wh[1])
    x, y = math.floor(max_wh[0]), math.floor(max_wh[1])
    if x >= width and y >= height:
        return None

    def round_aspect(number, key):
        return max(min(math.floor(number), math.ceil(number), key=key), 1)

    aspect = width / height
    if x / y >= aspect:
        x = round_aspect(y * aspect, key=lambda n: abs(aspect - n / y))
    else:
        y = round_aspect(x / aspect, key=lambda n: 0 if n == 0 else abs(aspect - x / n))
    final = (x, y)
    if (width, height) == final:
        return None
    box = (0, 0, width, height)
    if reducing_gap is None:
        return final, None
    fx = int((box[2] - box[0]) / final[0] / reducing_gap) or 1
    fy = int((box[3] - box[1]) / final[1] / reducing_gap) or 1
    if fx <= 1 and fy <= 1:
        return final, None
    support = _FILTER_SUPPORT[filter_name] - 0.5
    sx, sy = (box[2] - box[0]) / final[0] * support, (box[3] - box[1]) / final[1] * support
    rbox = (max(0, int(box[0] - sx)), max(0, int(box[1] - sy)), min(width, math.ceil(box[2] + sx)), min(height, math.ceil(box[3] + sy)))
    resize_box = ((box[0] - rbox[0]) / fx, (box[1] - rbox[1]) / fy, (box[2] - rbox[0]) / fx, (box[3] - rbox[1]) / fy)
    return final, ((fx, fy), rbox, resize_box)


def pillow_reduce_device(img, factor, box=None):
    """``Image.reduce(factor, box)`` for a device image uint8 [h, w, 3] -> uint8 [ceil(bh / fy), ceil(bw / fx), 3]
    (libImaging/Reduce.c: box average ``((sum + n / 2) * mult(n)) >> 24`` with ``mult(n) = (uint32)(2^32 / (256 n))``
    computed in float32, partial blocks at the right / bottom edge averaged over their own pixel count)."""
    import torch
    from .. import _lib
    assert img.is_cuda and img.dtype == torch.uint8 and img.dim() == 3 and img.shape[2] == 3 and img.is_contiguous()
    h, w = int(img.shape[0]), int(img.shape[1])
    fx, fy = (int(factor), int(factor)) if not isinstance(factor, (tuple, list)) else (int(factor[0]), int(factor[1]))
    x0, y0, x1, y1 = (0, 0, w, h) if box is None else (int(v) for v in box)
    if not (0 <= x0 < x1 <= w and 0 <= y0 < y1 <= h and fx >= 1 and fy >= 1):
        raise ValueError(f"reduce: box {box} / factor {factor} outside the image {w}x{h}")
    ow, oh = -(-(x1 - x0) // fx), -(-(y1 - y0) // fy)
    out = torch.empty((oh, ow, 3), dtype=torch.uint8, device=img.device)
    with torch.cuda.device(img.device):
        _lib.check(_lib.load().ap_pillow_reduce_u8(img.data_ptr(), h, w, x0, y0, x1 - x0, y1 - y0, fx, fy, out.data_ptr(),
                                                   _lib.current_stream_ptr(img.device)), "ap_pillow_reduce_u8")
    return out


def pillow_thumbnail_device(img, max_wh, filter_name: str = "bicubic", reducing_gap: float = 2.0):
    """``Image.thumbnail(max_wh)`` (BICUBIC, reducing_gap 2.0 = Pillow 12's defaults, which the reference relies on:
    services/segmentation.py:202-206) for a device image uint8 [h, w, 3]; bit-identical to Pillow (tested against it)."""
    plan = pillow_thumbnail_plan((int(img.shape[1]), int(img.shape[0])), max_wh, filter_name, reducing_gap)
    if plan is None:
        return img
    final, red = plan
    box = None
    if red is not None:
        factor, rbox, box = red
        img = pillow_reduce_device(img, factor, rbox)
    key = (int(img.shape[0]), int(img.shape[1]), final, filter_name, box, str(img.device))
    with _RESAMPLER_LOCK:
        rs = _RESAMPLER_CACHE.get(key)
        if rs is None:
            if len(_RESAMPLER_CACHE) >= 32:
                _RESAMPLER_CACHE.pop(next(iter(_RESAMPLER_CACHE)))
            rs = _RESAMPLER_CACHE[key] = DeviceResampler(key[:2], (final[1], final[0]), filter_name, img.device, box=box)
        return rs(img[None])[0]


def pillow_reduce_numpy(arr: np.ndarray, factor, box=None) -> np.ndarray:
    """NumPy statement of ``ap_pillow_reduce_u8`` (host logic tests compare it with ``Image.reduce`` itself)."""
    h, w = arr.shape[:2]
    fx, fy = (int(factor), int(factor)) if not isinstance(factor, (tuple, list)) else (int(factor[0]), int(factor[1]))
    x0, y0, x1, y1 = (0, 0, w, h) if box is None else (int(v) for v in box)
    bw, bh = x1 - x0, y1 - y0
    ow, oh = -(-bw // fx), -(-bh // fy)
    out = np.empty((oh, ow) + arr.shape[2:], np.uint8)
    src = arr[y0:y1, x0:x1].astype(np.uint64)
    for oy in range(oh):
        ys = slice(oy * fy, min(bh, (oy + 1) * fy))
        for ox in range(ow):
            xs = slice(ox * fx, min(bw, (ox + 1) * fx))
            blk = src[ys, xs]
            n = blk.shape[0] * blk.shape[1]
            mult = np.uint64(np.uint32(np.float32(4294967296.0) / np.float32(256 * n)))
            ssum = blk.reshape(n, -1).sum(0) + np.uint64(n // 2)
            out[oy, ox] = ((ssum * mult) & np.uint64(0xFFFFFFFF)) >> np.uint64(24)
    return out


def pillow_nearest_index(in_size: int, out_size: int) -> np.ndarray:
    """Source index per output position of ``Image.resize(..., NEAREST)`` along one axis (libImaging/Geometry.c,
    ImagingScaleAffine: ``xo = a / 2`` then ``xo += a`` per output pixel, ACCUMULATED in double, index = (int)xo);
    validated against Pillow in the host-logic tests."""
    a = float(np.float32(in_size)) / out_size
    xo = a * 0.5
    out = np.empty(out_size, np.int32)
    for x in range(out_size):
        out[x] = min(in_size - 1, int(xo))
        xo += a
    return out
