"""Validated run configuration (drop-in field names and defaults).

Mirrors the dataclass surface of /root/reference/atlas_patch/core/config.py:40-179
(field names, defaults, error types and the ``validated() -> self`` convention) so
that callers and plugins written against the reference keep working.  Device
strings stay ``cpu`` / ``cuda`` / ``cuda:<n>`` (config.py:25-37): on PyTorch-ROCm
``cuda`` *is* the HIP device, so the flag surface is unchanged on MI355X.
"""
from __future__ import annotations

import dataclasses as _dc
from pathlib import Path

_PRECISIONS = ("bfloat16", "float16", "float32")


def _require(cond: bool, message: str, exc=ValueError) -> None:
    if not cond:
        raise exc(message)


def normalise_device(device: str) -> str:
    """``cpu`` | ``cuda`` | ``cuda:<index>`` (lower-cased); anything else is a ValueError."""
    text = str(device).strip().lower()
    if text in ("cpu", "cuda"):
        return text
    head, sep, tail = text.partition(":")
    if head == "cuda" and sep:
        _require(tail == "" or tail.isdigit(),
                 f"Invalid CUDA device specification '{device}'. Use 'cuda' or 'cuda:<index>'.")
        return text
    raise ValueError(f"device must be 'cpu', 'cuda', or 'cuda:<index>', got {device}")


@_dc.dataclass
class SegmentationConfig:
    checkpoint_path: Path | None
    config_path: Path
    device: str = "cuda"
    thumbnail_power: float = 1.25
    thumbnail_max: int = 1024
    batch_size: int = 1
    mask_threshold: float = 0.0

    def validated(self) -> "SegmentationConfig":
        if self.checkpoint_path is not None:
            _require(self.checkpoint_path.exists(),
                     f"Checkpoint not found: {self.checkpoint_path}", FileNotFoundError)
        _require(self.config_path.exists(),
                 f"SAM2 config not found: {self.config_path}", FileNotFoundError)
        self.device = normalise_device(self.device)
        _require(self.thumbnail_max > 0, f"thumbnail_max must be > 0, got {self.thumbnail_max}")
        _require(self.batch_size > 0,
                 f"segmentation batch_size must be > 0, got {self.batch_size}")
        return self


@_dc.dataclass
class ExtractionConfig:
    patch_size: int
    target_magnification: int
    step_size: int | None = None
    workers: int | None = None
    max_open_slides: int | None = None
    tissue_threshold: float = 0.01
    white_threshold: int = 15
    black_threshold: int = 50
    fast_mode: bool = True
    write_batch: int = 8192

    def validated(self) -> "ExtractionConfig":
        _require(self.patch_size > 0, f"patch_size must be > 0, got {self.patch_size}")
        _require(self.target_magnification > 0,
                 f"target_magnification must be > 0, got {self.target_magnification}")
        if self.step_size is None:
            self.step_size = self.patch_size
        _require(self.step_size > 0, f"step_size must be > 0, got {self.step_size}")
        _require(0 <= self.tissue_threshold <= 1,
                 f"tissue_threshold must be between 0 and 1, got {self.tissue_threshold}")
        for label in ("white_threshold", "black_threshold", "write_batch"):
            value = getattr(self, label)
            _require(value > 0, f"{label} must be > 0, got {value}")
        if self.workers is not None:
            _require(self.workers > 0, f"workers must be > 0, got {self.workers}")
        if self.max_open_slides is None:
            self.max_open_slides = 200
        _require(self.max_open_slides > 0,
                 f"max_open_slides must be > 0, got {self.max_open_slides}")
        return self


@_dc.dataclass
class FeatureExtractionConfig:
    extractors: list[str]
    batch_size: int = 32
    device: str = "cuda"
    num_workers: int = 4
    precision: str = "float32"
    plugins: list[Path] = _dc.field(default_factory=list)

    def validated(self) -> "FeatureExtractionConfig":
        _require(bool(self.extractors), "At least one feature extractor must be provided.")
        _require(self.batch_size > 0, f"feature batch_size must be > 0, got {self.batch_size}")
        _require(self.num_workers >= 0,
                 f"feature num_workers must be >= 0, got {self.num_workers}")
        self.device = normalise_device(self.device)
        precision = str(self.precision).lower()
        _require(precision in _PRECISIONS,
                 f"precision must be one of {sorted(_PRECISIONS)}, got {self.precision}")
        self.precision = precision
        resolved = []
        for plugin in self.plugins:
            candidate = Path(plugin)
            _require(candidate.exists(), f"Feature plugin not found: {candidate}",
                     FileNotFoundError)
            resolved.append(candidate.resolve())
        self.plugins = resolved
        return self


@_dc.dataclass
class OutputConfig:
    output_root: Path
    save_images: bool = False
    visualize_grids: bool = False
    visualize_mask: bool = False
    visualize_contours: bool = False
    skip_existing: bool = True

    def validated(self) -> "OutputConfig":
        self.output_root.mkdir(parents=True, exist_ok=True)
        return self


@_dc.dataclass
class ProcessingConfig:
    input_path: Path
    recursive: bool = False
    mpp_csv: Path | None = None

    def validated(self) -> "ProcessingConfig":
        _require(self.input_path.exists(), f"Input path not found: {self.input_path}",
                 FileNotFoundError)
        if self.mpp_csv is not None:
            _require(self.mpp_csv.exists(), f"MPP CSV not found: {self.mpp_csv}",
                     FileNotFoundError)
        return self


@_dc.dataclass
class VisualizationConfig:
    thumbnail_size: int = 1024

    def validated(self) -> "VisualizationConfig":
        _require(self.thumbnail_size > 0,
                 f"thumbnail_size must be > 0, got {self.thumbnail_size}")
        return self


@_dc.dataclass
class AppConfig:
    processing: ProcessingConfig
    segmentation: SegmentationConfig
    extraction: ExtractionConfig
    output: OutputConfig
    features: FeatureExtractionConfig | None = None
    visualization: VisualizationConfig = _dc.field(default_factory=VisualizationConfig)
    device: str = "cuda"

    def validated(self) -> "AppConfig":
        for part in ("processing", "segmentation", "extraction", "output", "visualization"):
            setattr(self, part, getattr(self, part).validated())
        if self.features is not None:
            self.features = self.features.validated()
        return self
