"""Result types passed between the services (reference: core/models.py:10-36)."""
from __future__ import annotations

import dataclasses as _dc
from pathlib import Path
from typing import Any

import numpy as np


@_dc.dataclass(frozen=True)
class Slide:
    path: Path
    mpp: float | None = None
    backend: str | None = None

    @property
    def stem(self) -> str:
        return self.path.stem


@_dc.dataclass
class Mask:
    data: np.ndarray                 # float32 [h, w] in {0, 1}
    source_shape: tuple[int, int]


@_dc.dataclass
class ExtractionResult:
    slide: Slide
    h5_path: Path
    num_patches: int
    image_dir: Path | None = None
    visualizations: dict[str, Path] = _dc.field(default_factory=dict)
    metadata: dict[str, Any] = _dc.field(default_factory=dict)
    coords: np.ndarray | None = None
    patch_size_level0: int | None = None
