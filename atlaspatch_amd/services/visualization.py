"""Overlay PNGs: ``--visualize-grids / --visualize-mask / --visualize-contours`` (reference:
services/visualization.py:23-102, utils/visualization/{patches,mask,contours}.py).

Off the throughput path (host, Pillow, once per slide).  The mask overlay and the patch-grid overlay are the
reference's own Pillow calls in the same order, so those PNGs are pixel-identical given the same thumbnail, mask and
coords.  The contour overlay differs in ONE primitive: the reference rasterises the outlines with ``cv2.polylines``
(thickness 2 / 1), which this build draws with ``PIL.ImageDraw.line`` -- same vertices, same colours, line rasterisation
not pixel-identical (OpenCV is absent; stated, not hidden).  File names and the ``result.visualizations`` keys are the
reference's.  Failures are logged and swallowed per overlay, like the reference.
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Any, Optional, Sequence

import numpy as np
from PIL import Image, ImageDraw, ImageFont

from ..core.config import ExtractionConfig, OutputConfig, VisualizationConfig
from ..core.models import ExtractionResult
from ..core.paths import visualization_dir
from ..core.wsi.iwsi import IWSI
from .interfaces import VisualizationService

logger = logging.getLogger("atlaspatch_amd.visualization_service")


def _thumb(wsi: IWSI, size: int) -> Image.Image:
    return wsi.get_thumb((size, size)).convert("RGB")


def _info_box(image: Image.Image, text: str, padding: int = 10) -> None:
    """utils/visualization/patches.py:14-44: white box, top right, default bitmap font, 16-px lines."""
    draw = ImageDraw.Draw(image, "RGBA")
    font = ImageFont.load_default()
    lines = text.split("\n")
    line_height = 16
    width = 0
    for line in lines:
        box = draw.textbbox((0, 0), line, font=font)
        width = max(width, box[2] - box[0])
    box_w, box_h = width + 2 * padding, len(lines) * line_height + 2 * padding
    x1, y1 = image.width - box_w - 10, 10
    draw.rectangle(((x1, y1), (image.width - 10, y1 + box_h)), fill=(255, 255, 255, 230), outline=(0, 0, 0, 255), width=2)
    for i, line in enumerate(lines):
        draw.text((x1 + padding, y1 + padding + i * line_height), line, fill=(0, 0, 0, 255), font=font)


def visualize_patches_on_thumbnail(*, coords: np.ndarray, patch_size_level0: int, wsi: IWSI, output_dir: Path,
                                   thumbnail_size: int, info: Optional[dict[str, Any]] = None) -> Path:
    """Patch boxes on the thumbnail (utils/visualization/patches.py:47-90): float32 coordinates divided by the
    level-0 / thumbnail ratio, truncated to int, 1-px black rectangles, info box -> ``<stem>.png``."""
    thumbnail = _thumb(wsi, thumbnail_size)
    width0, height0 = wsi.get_size(lv=0)
    dsx, dsy = width0 / thumbnail.width, height0 / thumbnail.height
    scaled = np.asarray(coords).astype(np.float32)
    scaled[:, 0] = scaled[:, 0] / float(dsx)
    scaled[:, 1] = scaled[:, 1] / float(dsy)
    pw, ph = float(patch_size_level0) / float(dsx), float(patch_size_level0) / float(dsy)
    draw = ImageDraw.Draw(thumbnail, "RGBA")
    for cx, cy in scaled[:, :2].astype(float):
        draw.rectangle(((int(cx), int(cy)), (int(cx + pw), int(cy + ph))), outline=(0, 0, 0), width=1)
    lines = [f"Patches Extracted: {len(coords)}", f"WSI Size: {width0} x {height0}"]
    for key, label in (("patch_size", "Patch Size"), ("step_size", "Step Size"), ("tissue_thresh", "Tissue Threshold")):
        if info and key in info:
            lines.append(f"{label}: {info[key]}")
    _info_box(thumbnail, "\n".join(lines))
    output_dir.mkdir(parents=True, exist_ok=True)
    out = output_dir / f"{Path(wsi.path).stem}.png"
    thumbnail.save(out, quality=95)
    return out


def visualize_mask_on_thumbnail(*, mask: np.ndarray, wsi: IWSI, output_dir: Path, thumbnail_size: int) -> Path:
    """Semi-transparent green mask on the thumbnail + a black / white preview (utils/visualization/mask.py:11-45)."""
    thumb = _thumb(wsi, thumbnail_size)
    binary = (np.asarray(mask).astype(np.float32) > 0.5).astype(np.float32)
    mh, mw = binary.shape[:2]
    if (mw, mh) != (thumb.width, thumb.height):
        as_img = Image.fromarray((binary * 255).astype(np.uint8), mode="L")
        binary = np.asarray(as_img.resize((thumb.width, thumb.height), resample=Image.Resampling.NEAREST),
                            dtype=np.float32) / 255.0
    output_dir.mkdir(parents=True, exist_ok=True)
    stem = Path(wsi.path).stem
    Image.fromarray((binary * 255).astype(np.uint8), mode="L").save(output_dir / f"{stem}_mask_bw.png")
    layer = Image.new("RGBA", thumb.size, (0, 255, 0, 0))
    layer.putalpha(Image.fromarray((binary * 80).astype(np.uint8), mode="L"))
    out = output_dir / f"{stem}_mask.png"
    Image.alpha_composite(thumb.convert("RGBA"), layer).convert("RGB").save(out, quality=95)
    return out


def visualize_contours_on_thumbnail(*, tissue_contours: Sequence[np.ndarray], holes_contours: Sequence[Sequence[np.ndarray]],
                                    wsi: IWSI, output_dir: Path, thumbnail_size: int,
                                    mask_shape: Optional[tuple] = None) -> Path:
    """Tissue outlines in red (2 px), hole outlines in blue (1 px) (utils/visualization/contours.py:14-49); vertices
    scaled with ``scale_contours`` (float32 multiply, truncation) like the reference."""
    from ..utils.contours import scale_contours
    thumb = _thumb(wsi, thumbnail_size)
    if mask_shape is not None:
        sx, sy = float(thumb.width) / float(mask_shape[1]), float(thumb.height) / float(mask_shape[0])
    else:
        width0, height0 = wsi.get_size(lv=0)
        sx, sy = float(thumb.width) / float(width0), float(thumb.height) / float(height0)
    tissue = scale_contours(list(tissue_contours), sx, sy)
    holes = scale_contours([h for hs in holes_contours for h in hs], sx, sy)
    draw = ImageDraw.Draw(thumb)
    for polys, colour, width in ((tissue, (255, 0, 0), 2), (holes, (0, 0, 255), 1)):
        for poly in polys:
            pts = [tuple(int(v) for v in p) for p in np.asarray(poly).reshape(-1, 2)]
            if len(pts) == 1:
                draw.point(pts, fill=colour)
            elif pts:
                draw.line(pts + [pts[0]], fill=colour, width=width)
    output_dir.mkdir(parents=True, exist_ok=True)
    out = output_dir / f"{Path(wsi.path).stem}_contours.png"
    thumb.save(out, quality=95)
    return out


class DefaultVisualizationService(VisualizationService):
    def __init__(self, output_cfg: OutputConfig, extraction_cfg: ExtractionConfig,
                 vis_cfg: Optional[VisualizationConfig] = None) -> None:
        self.output_cfg = output_cfg
        self.extraction_cfg = extraction_cfg
        self.vis_cfg = vis_cfg or VisualizationConfig()

    def visualize(self, result: ExtractionResult, *, wsi: IWSI, mask: np.ndarray) -> None:
        out = self.output_cfg
        if not (out.visualize_grids or out.visualize_mask or out.visualize_contours):
            return
        vis_dir = visualization_dir(self.output_cfg, self.extraction_cfg)
        vis_dir.mkdir(parents=True, exist_ok=True)
        size = self.vis_cfg.thumbnail_size
        if out.visualize_grids:
            try:
                coords, ps0 = result.coords, result.patch_size_level0
                if coords is None or ps0 is None:
                    from ..utils.h5 import h5
                    with h5.File(str(result.h5_path), "r") as fh:
                        coords, ps0 = fh["coords"][:], int(fh.attrs["patch_size_level0"])
                xy = coords[:, :2] if coords.ndim == 2 and coords.shape[1] >= 2 else coords
                info = {"patch_size": self.extraction_cfg.patch_size,
                        "step_size": self.extraction_cfg.step_size or self.extraction_cfg.patch_size,
                        "tissue_thresh": self.extraction_cfg.tissue_threshold}
                result.visualizations["grids"] = visualize_patches_on_thumbnail(
                    coords=xy, patch_size_level0=ps0, wsi=wsi, output_dir=vis_dir, thumbnail_size=size, info=info)
            except Exception as exc:  # noqa: BLE001
                logger.warning("Failed to visualize grids for %s: %s", result.slide.path.name, exc)
        if out.visualize_mask:
            try:
                result.visualizations["mask"] = visualize_mask_on_thumbnail(mask=mask, wsi=wsi, output_dir=vis_dir,
                                                                            thumbnail_size=size)
            except Exception as exc:  # noqa: BLE001
                logger.warning("Failed to visualize mask for %s: %s", result.slide.path.name, exc)
        if out.visualize_contours:
            try:
                from ..utils.contours import mask_to_contours
                tissue, holes = mask_to_contours(mask, tissue_area_thresh=self.extraction_cfg.tissue_threshold)
                result.visualizations["contours"] = visualize_contours_on_thumbnail(
                    tissue_contours=tissue, holes_contours=holes, wsi=wsi, output_dir=vis_dir, thumbnail_size=size,
                    mask_shape=mask.shape)
            except Exception as exc:  # noqa: BLE001
                logger.warning("Failed to visualize contours for %s: %s", result.slide.path.name, exc)
