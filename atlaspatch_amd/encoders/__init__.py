"""Encoder registry + plugin API (reference: atlas_patch/models/patch/__init__.py:47-80)."""
from __future__ import annotations

import torch

from .base import FeatureExtractor, HipViTFeatureExtractor, PatchFeatureExtractor
from .custom import (CustomEncoderComponents, CustomEncoderLoader, register_custom_encoder,
                     register_feature_extractors_from_module, register_hip_vit_encoder)
from .registry import PatchFeatureExtractorRegistry
from .vit import register_clip, register_conch, register_dinov2, register_dinov3, register_more_vits, register_phikon, register_uni, register_vits

__all__ = ["FeatureExtractor", "HipViTFeatureExtractor", "PatchFeatureExtractor",
           "PatchFeatureExtractorRegistry", "build_default_registry", "CustomEncoderComponents",
           "CustomEncoderLoader", "register_custom_encoder", "register_hip_vit_encoder",
           "register_feature_extractors_from_module"]


def build_default_registry(*, device="cuda", num_workers: int = 0,
                           dtype: torch.dtype = torch.float32) -> PatchFeatureExtractorRegistry:
    """Built-in extractors of this build: the ViT family (vit_b_16 / b_32 / l_16 / l_32 / h_14, uni_v1, uni_v2, conch_v1, and the
    transformers-backed dinov2_small / base / large / giant, phikon_v1 / v2, midnight; the timm-hub ViTs h_optimus_0 / 1,
    prov_gigapath, lunit_vit_small_patch16 / 8_dino, pathorchestra; the CLIP towers clip_vit_b_32 / b_16 / l_14 / l_14_336, plip,
    quilt_b_32 / b_16, biomedclip; virchow_v1 / v2, h0_mini; eight dinov3_* names with the rotary embedding) on the
    native HIP path, registered in the reference's order (models/patch/__init__.py:58-80).  Builders are
    lazy (nothing touches the GPU until ``create``), so this is safe on a CPU-only host, exactly
    like the reference's registry which the CLI instantiates at import (cli.py:50)."""
    dev = torch.device(device)
    registry = PatchFeatureExtractorRegistry()
    register_vits(registry, device=dev, num_workers=num_workers, dtype=dtype)
    register_dinov2(registry, device=dev, num_workers=num_workers, dtype=dtype)
    register_dinov3(registry, device=dev, num_workers=num_workers, dtype=dtype)
    register_clip(registry, device=dev, num_workers=num_workers, dtype=dtype)
    register_conch(registry, device=dev, num_workers=num_workers, dtype=dtype)
    register_uni(registry, device=dev, num_workers=num_workers, dtype=dtype)
    register_phikon(registry, device=dev, num_workers=num_workers, dtype=dtype)
    register_more_vits(registry, device=dev, num_workers=num_workers, dtype=dtype)
    return registry
