from __future__ import annotations

from ..core.models import Slide
from ..core.wsi import WSIFactory


class DefaultWSILoader:
    """``WSILoader`` that delegates to the factory (reference: services/wsi_loader.py)."""

    def open(self, slide: Slide):
        return WSIFactory.load(str(slide.path), mpp=slide.mpp, backend=slide.backend)
