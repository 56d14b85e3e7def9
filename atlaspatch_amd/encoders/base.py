"""Encoder operator interface (the drop-in boundary) and its two implementations.

``FeatureExtractor`` is the interface the reference's storage layer calls
(/root/reference/atlas_patch/models/patch/base.py:15-29): attributes ``name`` and
``embedding_dim``, ``extract_batch(patches, *, batch_size=None) -> float32 [len, D]``
(new C-contiguous array; empty input -> ``np.empty((0, D), float32)``) and
``cleanup()``.  The caller owns ``patches`` and clears them after the call
(services/storage.py:384); implementations must not keep references.

* ``HipViTFeatureExtractor``  -- the MI355X-native path: tiles go to HBM as uint8,
  normalise + patch-embed + every transformer block + final LayerNorm run as the
  hand-written HIP kernels behind ``libatlaspatch_hip.so`` (ap_vit_forward_u8).
* ``PatchFeatureExtractor``   -- the generic plugin path for arbitrary ``nn.Module``
  encoders registered through ``register_custom_encoder`` (same contract as
  base.py:48-114 but without the per-batch DataLoader spawn); it exists so that
  third-party plugins keep working and is NOT what the benchmark measures.
"""
from __future__ import annotations

import abc
import logging
from typing import Callable, Optional, Sequence

import numpy as np
import torch
from PIL import Image

from .. import _lib

logger = logging.getLogger("atlaspatch_amd.encoders")


class FeatureExtractor(abc.ABC):
    """Patch-level feature extractor interface (reference: base.py:15-29)."""

    name: str
    embedding_dim: int

    @abc.abstractmethod
    def extract_batch(self, patches: Sequence[np.ndarray], *,
                      batch_size: Optional[int] = None) -> np.ndarray: ...

    @abc.abstractmethod
    def cleanup(self) -> None: ...


def _as_uint8_hwc(patch) -> np.ndarray:
    if isinstance(patch, Image.Image):
        patch = np.asarray(patch.convert("RGB"))
    arr = np.asarray(patch)
    if arr.ndim != 3 or arr.shape[2] != 3 or arr.dtype != np.uint8:
        raise ValueError(f"patch must be HWC uint8 RGB, got shape {arr.shape} dtype {arr.dtype}")
    return arr


class HipViTFeatureExtractor(FeatureExtractor):
    """ViT-family encoder running entirely in hand-written HIP kernels (gfx950).

    Parameters
    ----------
    vit : atlaspatch_amd.encoders.vit.HipViT
        Device-resident encoder (weights already uploaded).
    mean, std : normalisation constants of the reference transform.
    host_resize : optional ``(size, PIL resample)`` applied on the host with Pillow (the
        reference's own resampler) when tiles are not already ``crop_from`` sized, e.g. timm's
        ``Resize(224, bicubic)`` for 256-px tiles.  ``None`` = torchvision's
        ``ImageClassification(crop 224, resize 256)`` on 256-px tiles, which is a pure
        centre crop and runs on the device.
    """

    def __init__(self, *, name: str, vit, mean, std, max_batch: int = 1024,
                 host_resize: Optional[tuple] = None, expect_size: Optional[int] = None) -> None:
        self.name = name
        self.vit = vit
        self.embedding_dim = int(vit.embed_dim)
        self.mean = tuple(float(v) for v in mean)
        self.std = tuple(float(v) for v in std)
        self.max_batch = int(max_batch)
        self.host_resize = host_resize
        self.expect_size = expect_size
        self.device = vit.device

    def _prepare(self, patches: Sequence) -> np.ndarray:
        arrs = [_as_uint8_hwc(p) for p in patches]
        if self.host_resize is not None:
            size, resample = self.host_resize
            out = []
            for a in arrs:
                if min(a.shape[0], a.shape[1]) != size:
                    img = Image.fromarray(a)
                    w, h = img.size
                    # torchvision/timm Resize(int): shorter side -> size, aspect kept
                    if w <= h:
                        nw, nh = size, int(size * h / w)
                    else:
                        nw, nh = int(size * w / h), size
                    a = np.asarray(img.resize((nw, nh), resample))
                out.append(a)
            arrs = out
        shape0 = arrs[0].shape
        if any(a.shape != shape0 for a in arrs):
            raise ValueError("all patches of one batch must have the same shape")
        if self.expect_size is not None and (shape0[0] != self.expect_size or shape0[1] != self.expect_size):
            raise ValueError(
                f"{self.name}: the device preprocess implements the reference transform for "
                f"{self.expect_size}x{self.expect_size} tiles only (got {shape0[1]}x{shape0[0]}); "
                "resampling transforms are not part of this build")
        return np.stack(arrs, axis=0)

    @torch.inference_mode()
    def extract_batch(self, patches: Sequence[np.ndarray], *,
                      batch_size: Optional[int] = None) -> np.ndarray:
        if not patches:
            return np.empty((0, self.embedding_dim), dtype=np.float32)
        host = torch.from_numpy(self._prepare(patches))
        n = host.shape[0]
        # the reference runs chunks of min(len, batch_size) (base.py:83); results are identical
        # for any chunking because every image is independent, so use the larger device chunk
        step = max(1, min(n, self.max_batch))
        out = torch.empty((n, self.embedding_dim), dtype=torch.float32, device=self.device)
        for s in range(0, n, step):
            dev = host[s:s + step].to(self.device, non_blocking=False)
            self.vit.forward_u8(dev, self.mean, self.std, out[s:s + step])
        return out.cpu().numpy()

    @torch.inference_mode()
    def extract_device(self, tiles_u8: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Device-resident variant: tiles uint8 [n, H, W, 3] in HBM -> float32 [n, D] in HBM."""
        n = tiles_u8.shape[0]
        if out is None:
            out = torch.empty((n, self.embedding_dim), dtype=torch.float32, device=tiles_u8.device)
        step = max(1, min(max(n, 1), self.max_batch))
        for s in range(0, n, step):
            self.vit.forward_u8(tiles_u8[s:s + step], self.mean, self.std, out[s:s + step])
        return out

    def cleanup(self) -> None:
        self.vit.release()


class PatchFeatureExtractor(FeatureExtractor):
    """Generic torch-module extractor for plugin encoders (reference: base.py:48-114).

    Same contract and results as the reference's implementation; the per-item transform runs
    in-process (the reference spawns a fresh DataLoader per batch, base.py:84-91) and results
    are returned as float32 numpy.  ``num_workers`` is accepted for API compatibility.
    """

    def __init__(self, *, name: str, model: torch.nn.Module, embedding_dim: int, preprocess,
                 device: torch.device, dtype: torch.dtype = torch.float32, num_workers: int = 0,
                 non_blocking: bool = False, pin_memory: Optional[bool] = None,
                 forward_fn: Optional[Callable[[torch.Tensor], torch.Tensor]] = None) -> None:
        self.name = name
        self.model = model.to(device=device, dtype=dtype).eval()
        self.embedding_dim = int(embedding_dim)
        self.preprocess = preprocess
        self.device = device
        self.dtype = dtype
        self.num_workers = max(0, int(num_workers))
        self.non_blocking = bool(non_blocking)
        self.pin_memory = pin_memory if pin_memory is not None else self.device.type == "cuda"
        self._forward_fn = forward_fn

    @torch.inference_mode()
    def extract_batch(self, patches: Sequence[np.ndarray], *,
                      batch_size: Optional[int] = None) -> np.ndarray:
        if not patches:
            return np.empty((0, self.embedding_dim), dtype=np.float32)
        chunk = min(len(patches), batch_size or len(patches))
        results = []
        for start in range(0, len(patches), chunk):
            items = []
            for patch in patches[start:start + chunk]:
                image = patch if isinstance(patch, Image.Image) else Image.fromarray(patch)
                items.append(self.preprocess(image))
            batch = torch.stack(items, dim=0)
            if self.pin_memory and self.device.type == "cuda":
                batch = batch.pin_memory()
            batch = batch.to(device=self.device, dtype=self.dtype, non_blocking=self.non_blocking)
            feats = self._forward_fn(batch) if self._forward_fn else self.model(batch)
            if feats.ndim > 2:
                feats = torch.flatten(feats, start_dim=1)
            results.append(feats.detach())
        merged = results[0] if len(results) == 1 else torch.cat(results, dim=0)
        return merged.cpu().to(dtype=torch.float32).contiguous().numpy()

    def cleanup(self) -> None:
        try:
            self.model.cpu()
        except Exception:  # noqa: BLE001
            pass
