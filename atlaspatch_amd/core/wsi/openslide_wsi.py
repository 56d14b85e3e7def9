"""OpenSlide backend, active only when ``openslide`` is importable (it is not in the build image).

Mirrors /root/reference/atlas_patch/core/wsi/openslide_wsi.py:71-205: MPP from metadata keys /
free text / TIFF resolution / 10 / magnification (rounded to 4 decimals), magnification from the
objective-power property or inferred from MPP, ``read_region(...).convert("RGB")`` tiles.
"""
from __future__ import annotations

import re
from typing import Literal, Optional, Tuple, Union

import numpy as np
from PIL import Image

from .iwsi import IWSI

try:  # pragma: no cover - optional dependency
    import openslide
except Exception:  # noqa: BLE001
    openslide = None

# key tables and their priority order as the reference holds them (openslide_wsi.py:19-32)
_MPP_KEYS = ("openslide.mpp-x", "openslide.mpp-y", "openslide.mirax.MPP", "aperio.MPP", "hamamatsu.XResolution")
_MPP_TEXT_KEYS = ("openslide.comment", "tiff.ImageDescription")
_MAG_KEYS = ("aperio.AppMag", "openslide.objective-power", "hamamatsu.SourceLens")
_MPP_PATTERNS = (r"mpp\s*[:=]\s*([0-9]*\.?[0-9]+)", r"microns?\s+per\s+pixel[^0-9]*([0-9]*\.?[0-9]+)")


def _mpp_from_text(text) -> Optional[float]:
    """openslide_wsi.py:149-182: first pattern that matches AND parses."""
    if not text:
        return None
    for pattern in _MPP_PATTERNS:
        match = re.search(pattern, text, flags=re.IGNORECASE)
        if match:
            try:
                return float(match.group(1))
            except ValueError:
                continue
    return None


def mpp_from_properties(meta: dict) -> Optional[float]:
    """MPP lookup of the reference (openslide_wsi.py:71-128), on a plain property dict: direct keys in priority
    order -> free-text fields -> TIFF resolution -> 10 / magnification; rounded to 4 decimals."""
    for key in _MPP_KEYS:
        if key in meta:
            try:
                return round(float(meta[key]), 4)
            except (TypeError, ValueError):
                continue
    for key in _MPP_TEXT_KEYS:
        parsed = _mpp_from_text(meta.get(key))
        if parsed is not None:
            return round(parsed, 4)
    try:
        res, unit = meta.get("tiff.XResolution"), meta.get("tiff.ResolutionUnit")
        if res and unit:
            res_f = float(res)
            if unit.lower() == "centimeter":
                return round(10000 / res_f, 4)
            elif unit.lower() == "inch":
                return round(25400 / res_f, 4)
    except (TypeError, ValueError):
        pass
    for key in _MAG_KEYS:
        value = meta.get(key)
        if value is not None:
            try:
                mag = float(value)
                if mag > 0:
                    return round(10.0 / mag, 4)
            except (TypeError, ValueError):
                continue
    return None


def mag_from_properties(meta: dict, mpp: Optional[float], infer_mag) -> Optional[int]:
    """Magnification (openslide_wsi.py:130-147): the objective-power property, else inferred from MPP."""
    power = meta.get("openslide.objective-power")
    if power:
        try:
            return int(float(power))
        except (TypeError, ValueError):
            pass
    if mpp is not None:
        try:
            return infer_mag(mpp)
        except ValueError:
            pass
    return None


class OpenSlideWSI(IWSI):
    def __init__(self, path: str, mpp: Optional[float] = None, **_: object) -> None:
        if openslide is None:
            raise RuntimeError("openslide-python is not installed; the OpenSlide backend is unavailable")
        super().__init__(path=path, mpp=mpp)
        self._slide = None

    def _setup(self) -> None:
        self._slide = openslide.OpenSlide(self.path)
        self.w, self.h = self._slide.dimensions
        self.nlvl = self._slide.level_count
        self.ds = [float(d) for d in self._slide.level_downsamples]
        self.dims = [tuple(d) for d in self._slide.level_dimensions]
        self.meta = dict(self._slide.properties)
        if self._mpp_manual is not None:
            self.mpp = self.validate_mpp(float(self._mpp_manual), source="user-provided mpp")
        else:
            found = self._extract_mpp()
            self.mpp = self.validate_mpp(found, source="slide metadata") if found is not None else None
        self.mag = self._extract_mag()

    def _extract_mpp(self) -> Optional[float]:
        return mpp_from_properties(self.meta or {})

    def _extract_mag(self) -> Optional[int]:
        return mag_from_properties(self.meta or {}, self.mpp, self._infer_mag)

    def extract(self, xy: Tuple[int, int], lv: int, wh: Tuple[int, int], *,
                mode: Literal["array", "image"] = "array") -> Union[np.ndarray, Image.Image]:
        self._ensure_loaded()
        region = self._slide.read_region(xy, lv, wh).convert("RGB")
        if mode == "image":
            return region
        if mode == "array":
            return np.array(region)
        raise ValueError(f"Invalid mode: {mode}")

    def get_size(self, lv: int = 0) -> Tuple[int, int]:
        self._ensure_loaded()
        if lv < 0 or lv >= self.nlvl:
            raise IndexError(f"Level {lv} out of range")
        return self.dims[lv]

    def get_thumb(self, max_hw: Tuple[int, int]) -> Image.Image:
        self._ensure_loaded()
        return self._slide.get_thumbnail(max_hw).convert("RGB")

    def cleanup(self) -> None:
        if self._slide is not None:
            try:
                self._slide.close()
            finally:
                self._slide = None
        self._loaded = False
