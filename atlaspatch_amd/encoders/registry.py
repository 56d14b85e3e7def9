"""Name -> lazy builder registry (reference: models/patch/registry.py:11-44).

Names are case-folded; registering a name twice is a ``ValueError``; creating an unknown
name is a ``KeyError`` listing what is available; builder exceptions are logged and re-raised.
"""
from __future__ import annotations

import logging
from typing import Callable, Iterable, Mapping

from .base import FeatureExtractor

logger = logging.getLogger(__name__)

Builder = Callable[[], FeatureExtractor]


class PatchFeatureExtractorRegistry:
    def __init__(self) -> None:
        self._builders: dict[str, Builder] = {}

    def register(self, name: str, builder: Builder) -> None:
        folded = name.lower()
        if folded in self._builders:
            raise ValueError(f"Feature extractor '{name}' already registered.")
        self._builders[folded] = builder

    def available(self) -> list[str]:
        return sorted(self._builders)

    def create(self, name: str) -> FeatureExtractor:
        builder = self._builders.get(name.lower())
        if builder is None:
            raise KeyError(f"Unknown feature extractor '{name}'. Available: {self.available()}")
        try:
            return builder()
        except Exception:
            logger.exception("Failed to create feature extractor '%s'", name)
            raise

    def create_many(self, names: Iterable[str]) -> list[FeatureExtractor]:
        return [self.create(name) for name in names]

    def as_mapping(self) -> Mapping[str, Builder]:
        return dict(self._builders)
