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
