"""Tissue segmentation services (reference: services/segmentation.py:195-236).

The reference segments a <=1024-px thumbnail with a fine-tuned SAM2 (Hiera-T) whose package and
weights (``AtlasAnalyticsLab/AtlasPatch:model.pth``) are not available offline; two implementations:

* ``AnalyticSegmentationService`` -- for synthetic slides: rasterises the slide's own ellipses on
  the thumbnail grid (SURVEY.md 8d "synthetic masks"); lets ``process`` run end to end.
* ``SAM2SegmentationService`` -- the reference's constructor and thumbnail preparation
  (``get_thumbnail_at_power(1.25)`` + ``PIL.thumbnail(1024)``) around ``sam2_hip.Sam2HipPredictor``: the Hiera-T
  image path on the HIP float32 operator set (weights needed).  The architecture is pinned against an independent
  implementation -- transformers' ``Sam2Model`` on seeded weights through ``services/sam2_keys.py`` (oracle 2.8e-7, device
  <= 1e-4 / 2e-4, tests/test_sam2_hf_pin.py); what stays unpinned is the facebook-side ``sam2`` package itself and the real
  checkpoint ``AtlasAnalyticsLab/AtlasPatch:model.pth``, neither of which exists offline.
  Any other segmenter plugs in through ``SegmentationService``.
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor
from typing import Sequence

import numpy as np

from ..core.config import SegmentationConfig
from ..core.models import Mask
from ..core.wsi.iwsi import IWSI
from ..utils.stages import stage
from .interfaces import SegmentationService


def prepare_thumbnail(wsi: IWSI, cfg: SegmentationConfig):
    """1.25x power image, then Pillow ``thumbnail((max, max))`` (segmentation.py:202-206)."""
    with stage("thumbnail_at_power"):
        thumb = wsi.get_thumbnail_at_power(power=cfg.thumbnail_power, interpolation="optimise")
    if cfg.thumbnail_max:
        with stage("thumbnail_pil"):
            thumb.thumbnail((cfg.thumbnail_max, cfg.thumbnail_max))
    return thumb


def prepare_thumbnail_device(wsi: IWSI, cfg: SegmentationConfig, device):
    """``prepare_thumbnail`` with every pixel operation on the device: the level read where the backend can
    (``read_level_device``), ``cv2.resize`` to the 1.25x size (``ap_cv2_resize_u8``), Pillow's ``thumbnail`` = integer box
    reduce + bicubic resize (``ap_pillow_reduce_u8`` + ``ap_resample_u8``, bit-identical to Pillow).  -> uint8 [h, w, 3] in HBM."""
    from ..utils.resample import pillow_thumbnail_device
    with stage("thumbnail_at_power"):
        thumb = wsi.get_thumbnail_at_power_device(power=cfg.thumbnail_power, interpolation="optimise", device=device)
    if cfg.thumbnail_max:
        with stage("thumbnail_pil"):
            h, w = int(thumb.shape[0]), int(thumb.shape[1])
            if h > 100 * w:                  # Pillow resizes such images in two separate passes (Image.resize): host Pillow
                from PIL import Image
                img = Image.fromarray(thumb.cpu().numpy())
                img.thumbnail((cfg.thumbnail_max, cfg.thumbnail_max))
                import torch
                thumb = torch.from_numpy(np.asarray(img)).to(thumb.device)
            else:
                thumb = pillow_thumbnail_device(thumb, (cfg.thumbnail_max, cfg.thumbnail_max))
    return thumb


class AnalyticSegmentationService(SegmentationService):
    def __init__(self, thumbnail_max: int = 1024) -> None:
        self.thumbnail_max = thumbnail_max

    def segment_thumbnail(self, wsi: IWSI) -> Mask:
        if not hasattr(wsi, "tissue_mask"):
            raise TypeError(f"{type(wsi).__name__} has no analytic tissue mask; use a SAM2 or custom "
                            "SegmentationService for real slides")
        data = np.asarray(wsi.tissue_mask(self.thumbnail_max), dtype=np.float32)
        return Mask(data=data, source_shape=(int(data.shape[0]), int(data.shape[1])))

    def segment_batch(self, wsis: Sequence[IWSI]) -> list[Mask]:
        workers = max(1, min(8, len(wsis), os.cpu_count() or 8))
        with ThreadPoolExecutor(max_workers=workers, thread_name_prefix="thumb") as pool:
            return list(pool.map(self.segment_thumbnail, wsis))

    def close(self) -> None:
        pass


class SAM2SegmentationService(SegmentationService):
    """SAM2.1 Hiera-T tissue segmenter on the HIP operator set (reference: services/segmentation.py:195-236).

    Same constructor, thumbnail preparation (``get_thumbnail_at_power(1.25)`` + ``PIL.thumbnail(1024)``) and
    ``Mask`` output as the reference; the predictor is ``sam2_hip.Sam2HipPredictor``.  Weights: ``cfg.checkpoint_path``
    (the reference's ``model.pth`` layout, ``{"model": state_dict}``), else ``$ATLASPATCH_WEIGHTS_DIR/sam2.{pt,pth}``;
    the reference downloads them from the Hugging Face hub, which is not possible offline.
    ``ATLASPATCH_RANDOM_INIT=<seed>`` builds seeded random weights (plumbing tests / timing only)."""

    def __init__(self, cfg: SegmentationConfig) -> None:
        self.cfg = cfg
        self._predictor = None

    def _build(self):
        import logging

        from .sam2_hip import Sam2HipPredictor, load_sam2_state_dict
        path = self.cfg.checkpoint_path
        if path is None:
            root = os.environ.get("ATLASPATCH_WEIGHTS_DIR")
            for name in ("sam2.pt", "sam2.pth", "model.pth"):
                if root and os.path.exists(os.path.join(root, name)):
                    path = os.path.join(root, name)
                    break
        if path is None and os.environ.get("ATLASPATCH_RANDOM_INIT") in (None, ""):
            # the reference's default (segmentation.py:46-58): AtlasAnalyticsLab/AtlasPatch:model.pth from the Hugging Face
            # hub -- succeeds when the file is in the local hub cache or the host is online, otherwise falls through
            try:
                from huggingface_hub import hf_hub_download
                path = hf_hub_download(repo_id="AtlasAnalyticsLab/AtlasPatch", filename="model.pth")
            except Exception:  # noqa: BLE001
                path = None
        if path is not None:
            sd = load_sam2_state_dict(path)
        elif os.environ.get("ATLASPATCH_RANDOM_INIT") not in (None, ""):
            seed = int(os.environ["ATLASPATCH_RANDOM_INIT"])
            logging.getLogger("atlaspatch_amd.segmentation_service").warning(
                "SAM2: using seeded RANDOM weights (seed %d); masks are not meaningful", seed)
            sd = random_sam2_state_dict(seed)
        else:
            raise FileNotFoundError(
                "No SAM2 weights: pass a checkpoint path, or put sam2.pt / model.pth (AtlasAnalyticsLab/AtlasPatch) into "
                "ATLASPATCH_WEIGHTS_DIR; the Hugging Face download of the reference is not available offline. "
                "(ATLASPATCH_RANDOM_INIT=<seed> gives seeded random weights for plumbing tests.)")
        device = "cuda" if str(self.cfg.device).lower().startswith("cuda") else str(self.cfg.device)
        return Sam2HipPredictor(sd, device=device if ":" in str(self.cfg.device) or device != "cuda" else "cuda",
                                mask_threshold=self.cfg.mask_threshold)

    @property
    def predictor(self):
        if self._predictor is None:
            self._predictor = self._build()
        return self._predictor

    def segment_thumbnail(self, wsi: IWSI) -> Mask:
        return self.segment_prepared(self.prepare_input(wsi))

    def warm_up(self, batch_sizes) -> None:
        """Build the predictor and capture its hipGraphs for the forward batch sizes of the coming run, before any worker
        thread exists (``Sam2HipPredictor.capture_graphs``)."""
        self.predictor.capture_graphs(batch_sizes)

    # The two halves of segment_thumbnail, so that a caller can prepare slide k + 1's input (level read, cv2 / Pillow
    # resizes: host work + short device kernels) on another thread while slide k's forward runs (orchestration/runner.py).
    def prepare_input(self, wsi: IWSI):
        return prepare_thumbnail_device(wsi, self.cfg, self.predictor.device)

    def segment_prepared(self, thumb) -> Mask:
        with stage("sam2_predict"):
            data = self.predictor.predict_device(thumb, resize_to_input=True)
        return Mask(data=data, source_shape=(int(data.shape[0]), int(data.shape[1])))

    def segment_batch(self, wsis: Sequence[IWSI]) -> list[Mask]:
        # thumbnails are prepared on a thread pool like the reference's (segmentation.py:216-220): the level reads of
        # host-decoded backends overlap; every pixel operation after the read is on the device, in stream order
        predictor = self.predictor
        workers = max(1, min(8, len(wsis), os.cpu_count() or 8))
        with ThreadPoolExecutor(max_workers=workers, thread_name_prefix="thumb") as pool:
            thumbs = list(pool.map(lambda w: prepare_thumbnail_device(w, self.cfg, predictor.device), wsis))
        return self.segment_prepared_batch(thumbs)

    def segment_prepared_batch(self, thumbs) -> list[Mask]:
        """One forward for the whole batch (segmentation.py:142-180): the trunk runs on the stacked thumbnails; every mask
        equals the single-slide forward's bit for bit."""
        with stage("sam2_predict"):
            datas = self.predictor.predict_batch_device(list(thumbs), resize_to_input=True)
        return [Mask(data=d, source_shape=(int(d.shape[0]), int(d.shape[1]))) for d in datas]

    def close(self) -> None:
        if self._predictor is not None:
            self._predictor.close()
            self._predictor = None


def random_sam2_state_dict(seed: int = 0) -> dict:
    """Seeded random SAM2.1 Hiera-T image-path parameters under the package's state-dict names."""
    import math

    import torch

    from .sam2_hip import EMBED, WINDOW_SPEC, block_plan
    g = torch.Generator().manual_seed(seed)
    w = lambda *s, sc=0.02: torch.randn(*s, generator=g) * sc
    sd = {}

    def lin(name, o, i):
        sd[name + ".weight"] = w(o, i, sc=1.0 / math.sqrt(i)); sd[name + ".bias"] = w(o)

    def ln(name, d):
        sd[name + ".weight"] = 1.0 + w(d, sc=0.1); sd[name + ".bias"] = w(d)

    t = "image_encoder.trunk."
    sd[t + "patch_embed.proj.weight"] = w(EMBED, 3, 7, 7, sc=0.08); sd[t + "patch_embed.proj.bias"] = w(EMBED)
    sd[t + "pos_embed"] = w(1, EMBED, 7, 7, sc=0.2)
    sd[t + "pos_embed_window"] = w(1, EMBED, WINDOW_SPEC[0], WINDOW_SPEC[0], sc=0.2)
    for i, (din, dout, heads, window, qpool) in enumerate(block_plan()[0]):
        b = f"{t}blocks.{i}."
        ln(b + "norm1", din); lin(b + "attn.qkv", 3 * dout, din); lin(b + "attn.proj", dout, dout); ln(b + "norm2", dout)
        lin(b + "mlp.layers.0", 4 * dout, dout); lin(b + "mlp.layers.1", dout, 4 * dout)
        if din != dout:
            lin(b + "proj", dout, din)
    for n, c in enumerate((768, 384, 192, 96)):
        sd[f"image_encoder.neck.convs.{n}.conv.weight"] = w(256, c, 1, 1, sc=1.0 / math.sqrt(c))
        sd[f"image_encoder.neck.convs.{n}.conv.bias"] = w(256)
    sd["no_mem_embed"] = w(1, 1, 256)
    p = "sam_prompt_encoder."
    sd[p + "pe_layer.positional_encoding_gaussian_matrix"] = torch.randn(2, 128, generator=g)
    for i in range(4):
        sd[p + f"point_embeddings.{i}.weight"] = w(1, 256, sc=0.5)
    sd[p + "not_a_point_embed.weight"] = w(1, 256, sc=0.5); sd[p + "no_mask_embed.weight"] = w(1, 256, sc=0.5)
    d = "sam_mask_decoder."

    def attn(name, internal):
        lin(name + ".q_proj", internal, 256); lin(name + ".k_proj", internal, 256)
        lin(name + ".v_proj", internal, 256); lin(name + ".out_proj", 256, internal)

    for l in range(2):
        b = f"{d}transformer.layers.{l}."
        attn(b + "self_attn", 256); attn(b + "cross_attn_token_to_image", 128); attn(b + "cross_attn_image_to_token", 128)
        for k in range(1, 5):
            ln(b + f"norm{k}", 256)
        lin(b + "mlp.layers.0", 2048, 256); lin(b + "mlp.layers.1", 256, 2048)
    attn(d + "transformer.final_attn_token_to_image", 128); ln(d + "transformer.norm_final_attn", 256)
    sd[d + "iou_token.weight"] = w(1, 256, sc=0.5); sd[d + "mask_tokens.weight"] = w(4, 256, sc=0.5)
    sd[d + "obj_score_token.weight"] = w(1, 256, sc=0.5)
    sd[d + "output_upscaling.0.weight"] = w(256, 64, 2, 2, sc=1.0 / 16); sd[d + "output_upscaling.0.bias"] = w(64)
    ln(d + "output_upscaling.1", 64)
    sd[d + "output_upscaling.3.weight"] = w(64, 32, 2, 2, sc=1.0 / 8); sd[d + "output_upscaling.3.bias"] = w(32)
    sd[d + "conv_s0.weight"] = w(32, 256, 1, 1, sc=1.0 / 16); sd[d + "conv_s0.bias"] = w(32)
    sd[d + "conv_s1.weight"] = w(64, 256, 1, 1, sc=1.0 / 16); sd[d + "conv_s1.bias"] = w(64)
    for k, (o, i) in enumerate(((256, 256), (256, 256), (32, 256))):
        lin(d + f"output_hypernetworks_mlps.0.layers.{k}", o, i)
    return sd
