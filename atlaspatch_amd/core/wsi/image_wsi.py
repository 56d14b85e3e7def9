"""Single-level Pillow image backend (reference: core/wsi/image_wsi.py:9-147); ``mpp`` is required."""
from __future__ import annotations

from typing import Any, Literal, Optional, Tuple, Union

import numpy as np
from PIL import Image

from .iwsi import IWSI


class ImageWSI(IWSI):
    def __init__(self, **kwargs: Any) -> None:
        mpp = kwargs.get("mpp")
        if mpp is None:
            raise ValueError("mpp parameter is required for standard images")
        if mpp <= 0:
            raise ValueError(f"mpp must be positive, got {mpp}")
        super().__init__(**kwargs)
        self._image: Optional[Image.Image] = None
        self._mpp_value = self.validate_mpp(mpp, source="user-provided mpp")

    def _setup(self) -> None:
        try:
            if self._image is None:
                self._image = Image.open(self.path).convert("RGB")
        except FileNotFoundError as exc:
            raise FileNotFoundError(f"Image not found: {self.path}") from exc
        except Exception as exc:  # noqa: BLE001
            raise RuntimeError(f"Setup failed: Cannot open: {self.path}: {exc}") from exc
        self.w, self.h = self._image.size
        self.nlvl, self.ds, self.dims = 1, [1.0], [(self.w, self.h)]
        self.meta = {"format": self._image.format or "unknown", "mode": self._image.mode}
        self.mpp = self._mpp_value
        self.mag = self._extract_mag()

    def _extract_mpp(self) -> Optional[float]:
        return self._mpp_value

    def _extract_mag(self) -> Optional[int]:
        try:
            return self._infer_mag(self.mpp) if self.mpp is not None else None
        except ValueError:
            return None

    def extract(self, xy: Tuple[int, int], lv: int, wh: Tuple[int, int], *,
                mode: Literal["array", "image"] = "array") -> Union[np.ndarray, Image.Image]:
        self._ensure_loaded()
        if lv != 0:
            raise ValueError("Standard images only support level 0")
        x, y = xy
        region = self._image.crop((x, y, x + wh[0], y + wh[1])).convert("RGB")
        if mode == "image":
            return region
        if mode == "array":
            return np.array(region)
        raise ValueError(f"Invalid mode: {mode}")

    def get_size(self, lv: int = 0) -> Tuple[int, int]:
        self._ensure_loaded()
        if lv != 0:
            raise ValueError("Standard images only support level 0")
        return (self.w, self.h)

    def get_thumb(self, max_hw: Tuple[int, int]) -> Image.Image:
        self._ensure_loaded()
        thumb = self._image.copy()
        thumb.thumbnail(max_hw, Image.Resampling.LANCZOS)
        return thumb

    def cleanup(self) -> None:
        if self._image is not None:
            try:
                self._image.close()
            finally:
                self._image = None
        self._loaded = False
