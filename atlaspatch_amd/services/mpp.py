from __future__ import annotations

from pathlib import Path
from typing import Optional

from ..core.models import Slide
from ..utils.params import get_mpp_for_wsi, load_mpp_csv


class CSVMPPResolver:
    """Optional ``wsi,mpp`` CSV override (reference: services/mpp.py)."""

    def __init__(self, csv_path: Optional[Path]) -> None:
        self._table = load_mpp_csv(str(csv_path)) if csv_path is not None else None

    def resolve(self, slide: Slide) -> Optional[float]:
        return get_mpp_for_wsi(str(slide.path), self._table)
