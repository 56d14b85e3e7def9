"""Patch-coordinate extraction service (reference: services/extraction.py:17-197).

``PatchExtractionService.extract(wsi, mask, *, slide) -> ExtractionResult`` keeps the
reference's signature and H5 output; the work is done by the device path:

    mask --threshold (HIP)--> binary --border following (host C++)--> contours
         --area/hole filters, f32-truncating scale--> level-0 polygons
         --grid point-in-polygon + ballot compaction (HIP)--> coords int32 [N, 5]

No cross-contour de-duplication and no clipping to the slide, exactly like the reference
(SURVEY.md 9.4).

``fast_mode=False`` (reference extraction.py:105-116) and ``--save-images`` (:112-128, storage.py:163-248) read
every candidate tile: here the tiles go through the pinned tile ring into HBM in batches, the black / white
statistics are computed by ``ap_tile_content_counts`` (OpenCV's fixed-point RGB2GRAY / RGB2HSV, bit exact), rows
are dropped in order with the reference's ``fraction >= 0.7`` rule, and kept tiles are written as
``images/<stem>/<stem>_x{X}_y{Y}.png`` with Pillow exactly like ``H5PatchWriter._save_patch_image``.
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Sequence

import numpy as np

from ..core.config import ExtractionConfig, OutputConfig
from ..core.models import ExtractionResult, Slide
from ..core.paths import build_run_root, images_dir, patch_h5_path
from ..core.wsi.iwsi import IWSI
from ..utils.contours import DeviceContours
from .geometry import PatchGeometry, prepare_geometry
from .interfaces import ExtractionService

from ..utils.stages import stage

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
        self.h5_pool = None            # services/h5_writer_proc.H5WriterPool for cohort runs (set by the runner), else in-process

    def geometry(self, wsi: IWSI) -> PatchGeometry:
        return prepare_geometry(wsi, patch_size=self.cfg.patch_size, step_size=self.cfg.step_size,
                                target_magnification=self.cfg.target_magnification)

    def coords(self, wsi: IWSI, mask: np.ndarray) -> tuple[np.ndarray, PatchGeometry]:
        geometry = self.geometry(wsi)
        return device_coords(mask, level0_wh=wsi.get_size(lv=0), geometry=geometry,
                             tissue_thresh=self.cfg.tissue_threshold), geometry

    def extract(self, wsi: IWSI, mask: np.ndarray, *, slide: Slide) -> ExtractionResult:
        from .storage import H5PatchWriter

        (build_run_root(self.output_cfg, self.cfg) / "patches").mkdir(parents=True, exist_ok=True)
        out_h5 = patch_h5_path(slide, self.output_cfg, self.cfg)
        with stage("contours_and_grid"):
            coords, geometry = self.coords(wsi, mask)
        img_dir = None
        if self.output_cfg.save_images:
            img_dir = images_dir(slide, self.output_cfg, self.cfg)
            img_dir.mkdir(parents=True, exist_ok=True)
        if not self.cfg.fast_mode or img_dir is not None:
            coords = self._filter_and_save(wsi, coords, slide=slide, img_dir=img_dir)
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
        with stage("h5_coords"):
            # cohort runs hand the file to a helper process (libhdf5 is one lock per process: services/h5_writer_proc.py);
            # same writer class, same bytes.  No helper ready / pool off: written here
            total = None
            pool = self.h5_pool
            if pool is not None and pool.ready():
                total = pool.write(writer.to_kwargs(), str(out_h5), coords, writer.passports_array(coords))
            if total is None:
                total = writer.write_coords_array(out_h5, coords)
        logger.debug("Wrote %d coords for %s to %s", total, slide.path.name, out_h5)
        return ExtractionResult(slide=slide, h5_path=Path(out_h5), num_patches=int(total), image_dir=img_dir,
                                coords=None, patch_size_level0=geometry.patch_size_level0)

    # ------------------------------------------------------------------ non-fast mode / --save-images
    def _filter_and_save(self, wsi: IWSI, coords: np.ndarray, *, slide: Slide, img_dir, batch: int = 256) -> np.ndarray:
        """Read every candidate tile (reference extraction.py:105-128), drop black / white ones when fast_mode is
        off, save the kept ones when ``img_dir`` is set.  Returns the kept rows, order preserved."""
        import concurrent.futures as futures

        import torch
        from PIL import Image

        from ..utils.image import tile_content_flags

        if not torch.cuda.is_available():
            from .. import _lib
            raise _lib.HipLibraryError("--no-fast-mode / --save-images read tiles through the HIP device path; "
                                       "no HIP device is available (there is no CPU fallback)")
        device = torch.device("cuda", torch.cuda.current_device())
        ps = int(self.cfg.patch_size)
        n = int(coords.shape[0])
        keep = np.ones(n, dtype=bool)
        if n == 0:
            return np.ascontiguousarray(coords)
        # tiles are read at their level size; cv2.resize(patch, (ps, ps)) (extraction.py:112-113) runs on the device
        rw, rh = int(coords[0, 2]), int(coords[0, 3])
        resized = (rw, rh) != (ps, ps)
        host = torch.empty((batch, rh, rw, 3), dtype=torch.uint8, pin_memory=True)
        view = host.numpy()
        writers = futures.ThreadPoolExecutor(max_workers=max(2, min(8, __import__("os").cpu_count() or 4)),
                                             thread_name_prefix="patch-img") if img_dir is not None else None
        readers = futures.ThreadPoolExecutor(max_workers=4, thread_name_prefix="tile")
        pending = []

        def read(i, row):
            x, y, w_, h_, lv = (int(v) for v in row)
            tile = wsi.extract((x, y), lv=lv, wh=(w_, h_), mode="array")
            if tile.shape[:2] != (rh, rw):
                raise ValueError(f"tile source returned shape {tile.shape}, expected {(rh, rw, 3)}")
            view[i] = tile

        try:
            for lo in range(0, n, batch):
                hi = min(n, lo + batch)
                list(readers.map(lambda i: read(i - lo, coords[i]), range(lo, hi)))
                patches = view
                if resized or not self.cfg.fast_mode:
                    tiles = host[:hi - lo].to(device, non_blocking=False)
                    if resized:
                        from ..utils.resample import INTER_LINEAR, cv2_resize_device
                        tiles = cv2_resize_device(tiles, (ps, ps), INTER_LINEAR)
                        patches = tiles.cpu().numpy() if writers is not None else None
                    if not self.cfg.fast_mode:
                        black, white = tile_content_flags(tiles, black_thresh=self.cfg.black_threshold,
                                                          white_thresh=self.cfg.white_threshold)
                        keep[lo:hi] = ~(black | white)          # is_black first, then is_white (extraction.py:113-116)
                if writers is not None:
                    for i in range(lo, hi):
                        if keep[i]:
                            x, y = int(coords[i, 0]), int(coords[i, 1])
                            pending.append(writers.submit(
                                lambda arr, path: Image.fromarray(arr).save(str(path)), patches[i - lo].copy(),
                                img_dir / f"{slide.stem}_x{x}_y{y}.png"))
            for f in pending:
                f.result()
        finally:
            readers.shutdown(wait=True)
            if writers is not None:
                writers.shutdown(wait=True)
        return np.ascontiguousarray(coords[keep])
