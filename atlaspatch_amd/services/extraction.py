"""Patch-coordinate extraction service (reference: services/extraction.py:17-197).

``PatchExtractionService.extract(wsi, mask, *, slide) -> ExtractionResult`` keeps the
reference's signature and H5 output; the work is done by the device path:

    mask --threshold (HIP)--> binary --border following (host C++)--> contours
         --area/hole filters, f32-truncating scale--> level-0 polygons
         --grid point-in-polygon + ballot compaction (HIP)--> coords int32 [N, 5]

No cross-contour de-duplication and no clipping to the slide, exactly like the reference
(SURVEY.md 9.4).  ``fast_mode=False`` content filters and ``--save-images`` are outside
this build's scope and raise.
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Sequence

import numpy as np

from ..core.config import ExtractionConfig, OutputConfig
from ..core.models import ExtractionResult, Slide
from ..core.paths import build_run_root, patch_h5_path
from ..core.wsi.iwsi import IWSI
from ..utils.contours import DeviceContours
from .geometry import PatchGeometry, prepare_geometry
from .interfaces import ExtractionService

logger = logging.getLogger("atlaspatch_amd.extraction_service")


class _StaticLevels(IWSI):
    """Minimal IWSI carrying only pyramid metadata (used by ``coords_from_mask``)."""

    def __init__(self, downsamples: Sequence[float], mag) -> None:
        super().__init__(path="<levels>")
        self.ds = [float(d) for d in downsamples]
        self.mag = mag
        self._loaded = True

    def _setup(self): ...
    def _extract_mpp(self): return None
    def _extract_mag(self): return self.mag
    def extract(self, xy, lv, wh, *, mode="array"): raise NotImplementedError
    def get_size(self, lv=0): raise NotImplementedError
    def get_thumb(self, max_hw): raise NotImplementedError
    def cleanup(self): ...


def device_coords(mask: np.ndarray, *, level0_wh, geometry: PatchGeometry,
                  tissue_thresh: float) -> np.ndarray:
    """int32 [N, 5] rows for a mask, through the C ABI."""
    width, height = level0_wh
    mh, mw = mask.shape[:2]
    contours = DeviceContours(mask, tissue_area_thresh=tissue_thresh,
                              sx=width / float(mw), sy=height / float(mh))
    try:
        return contours.grid_coords(patch_size_src=geometry.patch_size_src, step_src=geometry.step_src,
                                    read_wh=geometry.read_wh, level=geometry.level)
    finally:
        contours.close()


def coords_from_mask(mask: np.ndarray, *, level0_wh, downsamples, src_mag, tgt_mag, patch_size,
                     step_size=None, tissue_thresh=0.01):
    geometry = prepare_geometry(_StaticLevels(downsamples, src_mag), patch_size=patch_size,
                                step_size=step_size, target_magnification=tgt_mag)
    return device_coords(mask, level0_wh=level0_wh, geometry=geometry, tissue_thresh=tissue_thresh), geometry


class PatchExtractionService(ExtractionService):
    def __init__(self, extraction_cfg: ExtractionConfig, output_cfg: OutputConfig) -> None:
        self.cfg = extraction_cfg.validated()
        self.output_cfg = output_cfg.validated()

    def geometry(self, wsi: IWSI) -> PatchGeometry:
        return prepare_geometry(wsi, patch_size=self.cfg.patch_size, step_size=self.cfg.step_size,
                                target_magnification=self.cfg.target_magnification)

    def coords(self, wsi: IWSI, mask: np.ndarray) -> tuple[np.ndarray, PatchGeometry]:
        geometry = self.geometry(wsi)
        return device_coords(mask, level0_wh=wsi.get_size(lv=0), geometry=geometry,
                             tissue_thresh=self.cfg.tissue_threshold), geometry

    def extract(self, wsi: IWSI, mask: np.ndarray, *, slide: Slide) -> ExtractionResult:
        from .storage import H5PatchWriter

        if not self.cfg.fast_mode:
            raise NotImplementedError("--no-fast-mode content filters are not part of this build")
        if self.output_cfg.save_images:
            raise NotImplementedError("--save-images is not part of this build")
        (build_run_root(self.output_cfg, self.cfg) / "patches").mkdir(parents=True, exist_ok=True)
        out_h5 = patch_h5_path(slide, self.output_cfg, self.cfg)
        coords, geometry = self.coords(wsi, mask)
        width0, height0 = wsi.get_size(lv=0)
        step = self.cfg.step_size or self.cfg.patch_size
        extra = {"filename": slide.path.name}
        extra.update(wsi.metadata_attrs())
        writer = H5PatchWriter(chunk_rows=self.cfg.write_batch, patch_size=self.cfg.patch_size,
                               patch_size_level0=geometry.patch_size_level0,
                               level0_mag=int(wsi.mag) if wsi.mag is not None else 0,
                               target_mag=self.cfg.target_magnification,
                               level0_wh=(int(width0), int(height0)),
                               overlap=max(0, int(self.cfg.patch_size) - int(step)),
                               slide_stem=slide.stem, wsi_path=str(wsi.path), extra_file_attrs=extra)
        total = writer.write_coords_array(out_h5, coords)
        logger.debug("Wrote %d coords for %s to %s", total, slide.path.name, out_h5)
        return ExtractionResult(slide=slide, h5_path=Path(out_h5), num_patches=int(total), image_dir=None,
                                coords=None, patch_size_level0=geometry.patch_size_level0)
