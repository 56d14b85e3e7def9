"""Tissue segmentation services (reference: services/segmentation.py:195-236).

The reference segments a <=1024-px thumbnail with a fine-tuned SAM2 (Hiera-T) whose package and
weights (``AtlasAnalyticsLab/AtlasPatch:model.pth``) are not available offline, so this build
ships the seam plus two implementations:

* ``AnalyticSegmentationService`` -- for synthetic slides: rasterises the slide's own ellipses on
  the thumbnail grid (SURVEY.md 8d "synthetic masks"); lets ``process`` run end to end.
* ``SAM2SegmentationService`` -- keeps the reference's constructor and thumbnail preparation
  (``get_thumbnail_at_power(1.25)`` + ``PIL.thumbnail(1024)``) but refuses to predict until the
  Hiera-T image path lands.  Any other segmenter plugs in through
  ``SegmentationService``.
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor
from typing import Sequence

import numpy as np

from ..core.config import SegmentationConfig
from ..core.models import Mask
from ..core.wsi.iwsi import IWSI
from .interfaces import SegmentationService


def prepare_thumbnail(wsi: IWSI, cfg: SegmentationConfig):
    """1.25x power image, then Pillow ``thumbnail((max, max))`` (segmentation.py:202-206)."""
    thumb = wsi.get_thumbnail_at_power(power=cfg.thumbnail_power, interpolation="optimise")
    if cfg.thumbnail_max:
        thumb.thumbnail((cfg.thumbnail_max, cfg.thumbnail_max))
    return thumb


class AnalyticSegmentationService(SegmentationService):
    def __init__(self, thumbnail_max: int = 1024) -> None:
        self.thumbnail_max = thumbnail_max

    def segment_thumbnail(self, wsi: IWSI) -> Mask:
        if not hasattr(wsi, "tissue_mask"):
            raise TypeError(f"{type(wsi).__name__} has no analytic tissue mask; use a SAM2 or custom "
                            "SegmentationService for real slides")
        data = np.asarray(wsi.tissue_mask(self.thumbnail_max), dtype=np.float32)
        return Mask(data=data, source_shape=(int(data.shape[0]), int(data.shape[1])))

    def segment_batch(self, wsis: Sequence[IWSI]) -> list[Mask]:
        workers = max(1, min(8, len(wsis), os.cpu_count() or 8))
        with ThreadPoolExecutor(max_workers=workers, thread_name_prefix="thumb") as pool:
            return list(pool.map(self.segment_thumbnail, wsis))

    def close(self) -> None:
        pass


class SAM2SegmentationService(SegmentationService):
    def __init__(self, cfg: SegmentationConfig) -> None:
        self.cfg = cfg

    def segment_thumbnail(self, wsi: IWSI) -> Mask:
        raise NotImplementedError(
            "SAM2 (Hiera-T) tissue segmentation is not part of this build yet: the sam2 package and "
            "the AtlasAnalyticsLab/AtlasPatch:model.pth weights are unavailable offline. Use "
            "AnalyticSegmentationService (synthetic slides) or plug a SegmentationService in.")

    def segment_batch(self, wsis: Sequence[IWSI]) -> list[Mask]:
        return [self.segment_thumbnail(w) for w in wsis]

    def close(self) -> None:
        pass
