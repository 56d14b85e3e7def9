"""Output layout: ``<out>/patches/<stem>.h5`` + ``.lock``, ``images/<stem>/``, ``visualization/``
(reference: core/paths.py:9-42)."""
from __future__ import annotations

from pathlib import Path

from .config import ExtractionConfig, OutputConfig
from .models import Slide


def build_run_root(output_cfg: OutputConfig, extraction_cfg: ExtractionConfig) -> Path:
    return output_cfg.output_root


def _patches_dir(output_cfg: OutputConfig, extraction_cfg: ExtractionConfig) -> Path:
    return build_run_root(output_cfg, extraction_cfg) / "patches"


def patch_h5_path(slide: Slide, output_cfg: OutputConfig, extraction_cfg: ExtractionConfig) -> Path:
    return _patches_dir(output_cfg, extraction_cfg) / (slide.stem + ".h5")


def patch_lock_path(slide: Slide, output_cfg: OutputConfig, extraction_cfg: ExtractionConfig) -> Path:
    return _patches_dir(output_cfg, extraction_cfg) / (slide.stem + ".lock")


def find_existing_patch(slide: Slide, output_cfg: OutputConfig,
                        extraction_cfg: ExtractionConfig) -> Path | None:
    candidate = patch_h5_path(slide, output_cfg, extraction_cfg)
    return candidate if candidate.exists() else None


def images_dir(slide: Slide, output_cfg: OutputConfig, extraction_cfg: ExtractionConfig) -> Path:
    return build_run_root(output_cfg, extraction_cfg) / "images" / slide.stem


def visualization_dir(output_cfg: OutputConfig, extraction_cfg: ExtractionConfig) -> Path:
    return build_run_root(output_cfg, extraction_cfg) / "visualization"
