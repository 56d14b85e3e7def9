"""Whole-slide tile source interface (input side of the hot path).

Same abstract surface as /root/reference/atlas_patch/core/wsi/iwsi.py:9-395 so that
backends written for the reference plug in unchanged: subclasses implement
``_setup, _extract_mpp, _extract_mag, extract, get_size, get_thumb, cleanup`` and
fill ``w, h, nlvl, ds, dims, meta, mpp, mag``.  The level/geometry arithmetic
(``optimal_level``, iwsi.py:325-358) is float64 host math and must stay
bit-identical because it selects ``level`` and ``read_w`` for every coords row.
"""
from __future__ import annotations

import abc
from typing import Any, Dict, Literal, Mapping, Optional, Sequence, Tuple, Union

import numpy as np
from PIL import Image

_MAG_BY_MPP = ((0.16, 80), (0.2, 60), (0.3, 40), (0.6, 20), (1.2, 10), (2.4, 5))

_VENDOR_KEYS = ("openslide.vendor", "tiff.make", "tiff.model", "hamamatsu.model", "leica.scanner")
_INSTITUTION_KEYS = ("tiff.institution", "tiff.institutionname", "aperio.institution",
                     "openslide.institution", "dicom.institutionname")
_STAIN_KEYS = ("aperio.stain", "aperio.staindescription", "openslide.stain", "hamamatsu.stain",
               "philips.stain")


class IWSI(abc.ABC):
    MPP_MIN = 0.1
    MPP_MAX = 10.0

    def __init__(self, path: str, mpp: Optional[float] = None):
        self.path = path
        self._mpp_manual = mpp
        self._loaded = False
        self.w: Optional[int] = None
        self.h: Optional[int] = None
        self.nlvl: Optional[int] = None
        self.ds: Optional[list[float]] = None
        self.dims: Optional[list[Tuple[int, int]]] = None
        self.meta: Optional[Dict[str, Any]] = None
        self.mpp: Optional[float] = None
        self.mag: Optional[int] = None

    # ------------------------------------------------------------------ backend hooks
    @abc.abstractmethod
    def _setup(self) -> None: ...

    @abc.abstractmethod
    def _extract_mpp(self) -> Optional[float]: ...

    @abc.abstractmethod
    def _extract_mag(self) -> Optional[int]: ...

    @abc.abstractmethod
    def extract(self, xy: Tuple[int, int], lv: int, wh: Tuple[int, int], *,
                mode: Literal["array", "image"] = "array") -> Union[np.ndarray, Image.Image]:
        """Region with top-left LEVEL-0 ``xy``, read at pyramid level ``lv``, size ``wh``."""

    @abc.abstractmethod
    def get_size(self, lv: int = 0) -> Tuple[int, int]: ...

    @abc.abstractmethod
    def get_thumb(self, max_hw: Tuple[int, int]) -> Image.Image: ...

    @abc.abstractmethod
    def cleanup(self) -> None: ...

    # ------------------------------------------------------------------ shared logic
    def _ensure_loaded(self) -> None:
        if not self._loaded:
            self._setup()
            self._loaded = True

    @classmethod
    def validate_mpp(cls, mpp: float, *, source: str = "metadata") -> float:
        if not (cls.MPP_MIN <= mpp <= cls.MPP_MAX):
            raise ValueError(
                f"MPP value {mpp} from {source} is outside valid range "
                f"[{cls.MPP_MIN}, {cls.MPP_MAX}] µm/pixel. "
                "This may indicate corrupted metadata or incorrect input. "
                "If this value is intentional, please verify your data source.")
        return mpp

    def _infer_mag(self, m: float) -> int:
        for upper, magnification in _MAG_BY_MPP:
            if m < upper:
                return magnification
        raise ValueError(f"Cannot infer magnification from mpp {m}")

    def optimal_level(self, target_ds: float) -> Tuple[int, float]:
        """(level, residual downsample) for a target downsample (iwsi.py:325-358).

        1. a level within 0.01 of the target wins outright (first such level);
        2. otherwise the LAST level whose downsample is <= target (scan stops at the
           first larger one), residual = target / that;
        3. if the target is below level 0's downsample: first level >= target.
        """
        self._ensure_loaded()
        levels = self.ds or [1.0]
        for index, value in enumerate(levels):
            if abs(value - target_ds) < 0.01:
                return index, 1.0
        if target_ds >= levels[0]:
            chosen = 0
            for index, value in enumerate(levels):
                if value > target_ds:
                    break
                chosen = index
            return chosen, target_ds / levels[chosen]
        for index, value in enumerate(levels):
            if value >= target_ds:
                return index, value / target_ds
        raise ValueError(f"No level for target downsample {target_ds}")

    @staticmethod
    def _meta_lookup(meta: Mapping[str, Any], keys: Sequence[str], token: str) -> Optional[str]:
        """First non-empty value among ``keys`` (case-insensitive), else the first key (sorted)
        containing ``token`` (iwsi.py:167-198)."""
        lowered: dict[str, Any] = {}
        for key, value in (meta or {}).items():
            if value is None:
                continue
            lowered.setdefault(str(key).lower(), value)

        def clean(value: Any) -> Optional[str]:
            if value is None:
                return None
            text = str(value).strip()
            return text or None

        for key in keys:
            text = clean(lowered.get(key.lower()))
            if text:
                return text
        for key in sorted(lowered):
            if token in key:
                text = clean(lowered[key])
                if text:
                    return text
        return None

    def metadata_attrs(self) -> Dict[str, Any]:
        """Optional H5 file attrs: mpp, magnification, vendor, institution, stain (iwsi.py:200-244)."""
        self._ensure_loaded()
        meta = self.meta or {}
        out: Dict[str, Any] = {}
        if self.mpp is not None:
            out["mpp"] = self.mpp
        if self.mag is not None:
            out["magnification"] = int(self.mag)
        for label, keys in (("vendor", _VENDOR_KEYS), ("institution", _INSTITUTION_KEYS),
                            ("stain", _STAIN_KEYS)):
            found = self._meta_lookup(meta, keys, label)
            if found:
                out[label] = found
        return out

    def _thumbnail_geometry(self, power: float):
        """(level, read_wh, out_wh) of the power-based thumbnail (iwsi.py:246-303)."""
        self._ensure_loaded()
        if self.mag is None:
            raise ValueError(
                "WSI base magnification is unknown; cannot generate power-based thumbnail.")
        width0, height0 = self.get_size(lv=0)
        if width0 <= 0 or height0 <= 0:
            raise ValueError("Invalid WSI dimensions.")
        if float(power) <= 0:
            raise ValueError("thumbnail power must be positive")
        ds_target = max(1e-6, float(self.mag) / float(power))
        level, _ = self.optimal_level(ds_target)
        ds_level = float((self.ds or [1.0])[level])
        read_wh = (max(1, int(round(width0 / ds_level))), max(1, int(round(height0 / ds_level))))
        out_wh = (max(1, int(round(width0 / ds_target))), max(1, int(round(height0 / ds_target))))
        return level, read_wh, out_wh

    def read_level_device(self, level: int, wh: Tuple[int, int], device):
        """Optional capability: the region (0, 0, wh) of pyramid ``level`` as uint8 [h, w, 3] IN HBM, or None when the
        backend has nothing better than ``extract`` (the caller then reads on the host and uploads).  Backends that can
        produce pixels on the device (synthetic slides) or read strips in parallel outside the interpreter lock
        (OpenSlide through the native hook) override it: the whole-level read is what the reference's thumbnail spends
        its time on (iwsi.py:296-303: one ``read_region`` of the full level)."""
        return None

    def get_thumbnail_at_power_device(self, *, power: float = 1.25, interpolation: str = "optimise", device=None):
        """``get_thumbnail_at_power`` with the image left in HBM: uint8 [out_h, out_w, 3] on ``device``.  Same level
        choice, same ``cv2.resize`` (AREA when shrinking / CUBIC when enlarging, iwsi.py:305-321, ``ap_cv2_resize_u8``);
        the segmentation path continues on the device from here (Pillow ``thumbnail``, the SAM2 input resize)."""
        import torch
        from atlaspatch_amd.utils.resample import cv2_resize_device, thumbnail_interpolation
        level, read_wh, (out_w, out_h) = self._thumbnail_geometry(power)
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        region = self.read_level_device(level, read_wh, dev)
        if region is None:
            arr = self.extract((0, 0), lv=level, wh=read_wh, mode="array")
            if not isinstance(arr, np.ndarray):
                raise RuntimeError("Failed to read thumbnail region as array")
            region = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.uint8)).to(dev)
        if region.shape[1] != out_w or region.shape[0] != out_h:
            region = cv2_resize_device(region[None], (out_w, out_h),
                                       thumbnail_interpolation(tuple(region.shape[:2]), (out_w, out_h), interpolation))[0]
        return region

    def get_thumbnail_at_power(self, *, power: float = 1.25,
                               interpolation: str = "optimise") -> Image.Image:
        """Whole-slide RGB image at objective ``power`` (iwsi.py:246-323).

        Reads the pyramid level chosen by ``optimal_level(mag / power)`` in full and, when
        that level is not already the exact ``(W0/ds, H0/ds)`` size, resamples it with
        ``cv2.resize`` semantics (AREA when shrinking / CUBIC when enlarging, iwsi.py:305-321)
        on the device (``ap_cv2_resize_u8``).
        """
        level, read_wh, (out_w, out_h) = self._thumbnail_geometry(power)
        region = self.extract((0, 0), lv=level, wh=read_wh, mode="array")
        if not isinstance(region, np.ndarray):
            raise RuntimeError("Failed to read thumbnail region as array")
        if region.shape[1] != out_w or region.shape[0] != out_h:
            from atlaspatch_amd.utils.resample import cv2_resize_array, thumbnail_interpolation
            region = cv2_resize_array(region, (out_w, out_h),
                                      thumbnail_interpolation(region.shape[:2], (out_w, out_h), interpolation))
        return Image.fromarray(region)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.cleanup()

    def __repr__(self) -> str:
        state = f"{self.w}x{self.h}" if self._loaded else "loading pending"
        return f"<{type(self).__name__}: {state}>"
