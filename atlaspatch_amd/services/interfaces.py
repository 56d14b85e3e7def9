"""Service seams (reference: services/interfaces.py:12-40): segmenters, extractors, embedders,
WSI loaders and MPP resolvers plug in here without touching their callers."""
from __future__ import annotations

import abc
from typing import Optional, Protocol, Sequence

import numpy as np

from ..core.models import ExtractionResult, Mask, Slide
from ..core.wsi.iwsi import IWSI


class SegmentationService(abc.ABC):
    @abc.abstractmethod
    def segment_thumbnail(self, wsi: IWSI) -> Mask: ...

    @abc.abstractmethod
    def segment_batch(self, wsis: Sequence[IWSI]) -> list[Mask]: ...


class ExtractionService(abc.ABC):
    @abc.abstractmethod
    def extract(self, wsi: IWSI, mask: np.ndarray, *, slide: Slide) -> ExtractionResult: ...


class FeatureEmbeddingService(abc.ABC):
    @abc.abstractmethod
    def embed_features(self, result: ExtractionResult, *, wsi: IWSI) -> ExtractionResult: ...


class VisualizationService(abc.ABC):
    @abc.abstractmethod
    def visualize(self, result: ExtractionResult, *, wsi: IWSI, mask: np.ndarray) -> None: ...


class MPPResolver(Protocol):
    def resolve(self, slide: Slide) -> Optional[float]: ...


class WSILoader(Protocol):
    def open(self, slide: Slide) -> IWSI: ...
