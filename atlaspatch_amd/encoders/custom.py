"""Plugin API (reference: models/patch/custom.py:19-146), kept verbatim at the boundary:

* a plugin file defines ``register_feature_extractors(registry, device, dtype, num_workers)``;
  it is loaded by path and the hook is invoked WITH KEYWORD ARGUMENTS (custom.py:146);
* inside, the plugin calls ``register_custom_encoder(registry=..., name=..., embedding_dim=...,
  loader=..., device=..., dtype=..., num_workers=0, non_blocking=False)`` whose ``loader(device,
  dtype)`` returns ``CustomEncoderComponents(model, preprocess, forward_fn=None)``.

MI355X addition: ``register_hip_vit_encoder`` lets a plugin hand over a ViT state dict and have
it run in the native HIP kernels instead of as a torch module.
"""
from __future__ import annotations

import dataclasses as _dc
import importlib.util
import logging
from pathlib import Path
from types import ModuleType
from typing import Callable, Optional, Protocol

import torch
from PIL import Image

from .base import PatchFeatureExtractor
from .registry import PatchFeatureExtractorRegistry

logger = logging.getLogger("atlaspatch_amd.encoders.custom")


class CustomEncoderLoader(Protocol):
    def __call__(self, device: torch.device, dtype: torch.dtype) -> "CustomEncoderComponents": ...


@_dc.dataclass
class CustomEncoderComponents:
    model: torch.nn.Module
    preprocess: Callable[[Image.Image], torch.Tensor]
    forward_fn: Optional[Callable[[torch.Tensor], torch.Tensor]] = None


def register_custom_encoder(*, registry: PatchFeatureExtractorRegistry, name: str, embedding_dim: int,
                            loader: CustomEncoderLoader, device: torch.device, dtype: torch.dtype,
                            num_workers: int = 0, non_blocking: bool = False) -> None:
    def build() -> PatchFeatureExtractor:
        parts = loader(device, dtype)
        if not isinstance(parts, CustomEncoderComponents):
            raise TypeError(f"Custom encoder loader for '{name}' must return CustomEncoderComponents, "
                            f"got {type(parts)}.")
        return PatchFeatureExtractor(name=name, model=parts.model, embedding_dim=embedding_dim,
                                     preprocess=parts.preprocess, device=device, dtype=dtype,
                                     num_workers=num_workers, non_blocking=non_blocking,
                                     forward_fn=parts.forward_fn)

    registry.register(name, build)


def register_hip_vit_encoder(*, registry: PatchFeatureExtractorRegistry, name: str,
                             state_dict_loader: Callable[[], dict], arch: str,
                             device: torch.device, dtype: torch.dtype, mean=None, std=None,
                             source: str = "auto", **arch_overrides) -> None:
    """Register a ViT checkpoint to run in the native HIP kernels (lazy, like every builder)."""
    from .vit import build_hip_vit_extractor

    def build():
        return build_hip_vit_extractor(name=name, arch=arch, state_dict=state_dict_loader(),
                                       device=device, dtype=dtype, mean=mean, std=std, source=source,
                                       **arch_overrides)

    registry.register(name, build)


class CustomRegistryHook(Protocol):
    def __call__(self, registry: PatchFeatureExtractorRegistry, device: torch.device,
                 dtype: torch.dtype, num_workers: int) -> None: ...


def _load_module(path: Path) -> ModuleType:
    spec = importlib.util.spec_from_file_location(path.stem, path)
    if spec is None or spec.loader is None:
        raise RuntimeError(f"Failed to load module spec from {path}")
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module


def register_feature_extractors_from_module(module_path, registry: PatchFeatureExtractorRegistry, *,
                                            device: torch.device, dtype: torch.dtype,
                                            num_workers: int = 0) -> None:
    path = Path(module_path).expanduser().resolve()
    hook = getattr(_load_module(path), "register_feature_extractors", None)
    if not callable(hook):
        raise AttributeError(f"Custom encoder module {path} must define a callable "
                             "'register_feature_extractors(registry, device, dtype, num_workers)'.")
    logger.info("Registering custom feature extractors from %s", path)
    hook(registry=registry, device=device, dtype=dtype, num_workers=num_workers)


__all__ = ["CustomEncoderComponents", "CustomEncoderLoader", "CustomRegistryHook",
           "register_custom_encoder", "register_hip_vit_encoder",
           "register_feature_extractors_from_module"]
