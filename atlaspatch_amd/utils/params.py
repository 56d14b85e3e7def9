"""Slide discovery and the ``wsi,mpp`` CSV override (reference: utils/params.py:27-190)."""
from __future__ import annotations

import csv
import logging
from pathlib import Path
from typing import Dict, Optional

import click

logger = logging.getLogger("atlaspatch_amd.utils")

SUPPORTED_EXTS = {".svs", ".tif", ".tiff", ".ndpi", ".vms", ".vmu", ".scn", ".mrxs", ".bif", ".biff", ".dcm",
                  ".dicom", ".png", ".jpg", ".jpeg", ".bmp", ".webp", ".gif", ".synth"}


def get_wsi_files(path: str, *, recursive: bool = False) -> list[str]:
    root = Path(path)
    if root.is_file():
        if root.suffix.lower() not in SUPPORTED_EXTS:
            logger.warning("File may not be a supported WSI format: %s", root.name)
        return [str(root)]
    walker = root.rglob if recursive else root.glob
    hits: set[Path] = set()
    for ext in SUPPORTED_EXTS:
        hits.update(walker(f"*{ext}"))
        hits.update(walker(f"*{ext.upper()}"))
    files = sorted(hits)
    if not files:
        raise click.ClickException(f"No WSI files found in directory: {path}\n"
                                   "Supported formats: SVS, TIF, TIFF, NDPI, PNG, JPG, etc.")
    return [str(p) for p in files]


def load_mpp_csv(csv_path: str) -> Dict[str, float]:
    """CSV with header ``wsi,mpp``; keys are file names (and stems) -> float mpp."""
    table: Dict[str, float] = {}
    with open(csv_path, newline="") as handle:
        reader = csv.DictReader(handle)
        if reader.fieldnames is None or not {"wsi", "mpp"} <= {f.strip().lower() for f in reader.fieldnames}:
            raise click.ClickException("MPP CSV must have columns 'wsi' and 'mpp'")
        for row in reader:
            norm = {k.strip().lower(): v for k, v in row.items() if k}
            try:
                table[Path(norm["wsi"].strip()).name] = float(norm["mpp"])
            except (KeyError, TypeError, ValueError):
                logger.warning("Skipping malformed MPP CSV row: %s", row)
    return table


def get_mpp_for_wsi(wsi_path: str, mpp_map: Optional[Dict[str, float]]) -> Optional[float]:
    if not mpp_map:
        return None
    p = Path(wsi_path)
    return mpp_map.get(p.name, mpp_map.get(p.stem))
