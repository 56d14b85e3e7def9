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
    resize : optional ``(size, "bicubic" | "bilinear")`` -- the ``Resize(size)`` that opens the reference
        transform (timm's ``Resize(224, bicubic)`` for uni_v1, open_clip's ``Resize(448, bicubic)`` for
        conch_v1): shorter side -> size with the aspect kept, done on the device by ``ap_resample_u8``
        bit-identically to Pillow (the reference's resampler); the centre crop that follows is part of
        the preprocess kernel.  ``None`` = torchvision's ``ImageClassification(crop 224, resize 256)``
        on 256-px tiles, which is a pure centre crop.
    """

    def __init__(self, *, name: str, vit, mean, std, max_batch: int = 1024,
                 resize: Optional[tuple] = None, expect_size: Optional[int] = None) -> None:
        self.name = name
        self.vit = vit
        self.embedding_dim = int(vit.embed_dim)
        self.mean = tuple(float(v) for v in mean)
        self.std = tuple(float(v) for v in std)
        self.max_batch = int(max_batch)
        self.resize = resize
        self.expect_size = expect_size
        self.device = vit.device
        self._resamplers: dict = {}

    def _validated(self, patches: Sequence) -> list:
        arrs = [_as_uint8_hwc(p) for p in patches]
        shape0 = arrs[0].shape
        if any(a.shape != shape0 for a in arrs):
            raise ValueError("all patches of one batch must have the same shape")
        if self.resize is None and self.expect_size is not None and \
                (shape0[0] != self.expect_size or shape0[1] != self.expect_size):
            raise ValueError(
                f"{self.name}: the device preprocess implements the reference transform for "
                f"{self.expect_size}x{self.expect_size} tiles only (got {shape0[1]}x{shape0[0]})")
        return arrs

    def _prepare(self, patches: Sequence) -> np.ndarray:
        """The validated patches gathered into the grow-only PINNED buffer (what np.stack would cost, but the H2D copy that
        follows runs at the pinned rate instead of through the driver's staging buffer).  Safe to reuse: extract_batch ends
        with a synchronous D2H, so every copy out of this buffer has completed when the next call fills it."""
        arrs = self._validated(patches)
        view = self._pinned_view(len(arrs), arrs[0].shape)
        for i, a in enumerate(arrs):
            view[i] = a
        return view

    def _pinned_view(self, n: int, shape0) -> np.ndarray:
        need = n * int(np.prod(shape0))
        pin = getattr(self, "_pin", None)
        if pin is None or pin.numel() < need:
            self._pin = pin = torch.empty(need, dtype=torch.uint8, pin_memory=True)
        return pin[:need].view(n, *shape0).numpy()

    UPLOAD_CHUNK = 8        # tiles per gather + H2D piece (1.5 MB at 256 x 256: the copy engine starts while the host still gathers)

    def _upload(self, arrs: list, dev: torch.Tensor) -> None:
        """arrs -> dev (uint8 [n, H, W, 3] on the device) through the pinned buffer, in pieces: the H2D copy of piece k runs
        while the host gathers piece k + 1 (a 32-patch call: 0.11 ms of gather + 0.12 ms of copy one after the other before; measured
        in one process, tools/extract_batch_probe.py: 2.492 -> 2.46 ms per call, +1.3 % -- the call is its 2.24-ms forward)."""
        n = len(arrs)
        view = self._pinned_view(n, arrs[0].shape)
        pin = torch.from_numpy(view)
        for s in range(0, n, self.UPLOAD_CHUNK):
            e = min(n, s + self.UPLOAD_CHUNK)
            for i in range(s, e):
                view[i] = arrs[i]
            dev[s:e].copy_(pin[s:e], non_blocking=True)

    def resized(self, tiles_u8: torch.Tensor) -> torch.Tensor:
        """The transform's leading ``Resize`` on a device batch uint8 [n, H, W, 3] (identity when absent)."""
        if self.resize is None:
            return tiles_u8
        size, filt = self.resize
        h, w = int(tiles_u8.shape[1]), int(tiles_u8.shape[2])
        if min(h, w) == size:
            return tiles_u8
        rs = self._resamplers.get((h, w))
        if rs is None:
            from ..utils.resample import DeviceResampler
            # torchvision/timm Resize(int): shorter side -> size, aspect kept
            out_hw = (int(size * h / w), size) if w <= h else (size, int(size * w / h))
            rs = self._resamplers[(h, w)] = DeviceResampler((h, w), out_hw, filt, self.device)
        return rs(tiles_u8)

    def forward_device(self, tiles_u8: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        """One device chunk: resize (if the transform has one) + preprocess + encoder, all HIP kernels."""
        n = int(tiles_u8.shape[0])
        step = max(1, min(max(n, 1), self.max_batch))
        for s in range(0, n, step):
            self.vit.forward_u8(self.resized(tiles_u8[s:s + step]), self.mean, self.std, out[s:s + step])
        return out

    @torch.inference_mode()
    def extract_batch(self, patches: Sequence[np.ndarray], *,
                      batch_size: Optional[int] = None) -> np.ndarray:
        if not patches:
            return np.empty((0, self.embedding_dim), dtype=np.float32)
        arrs = self._validated(patches)
        n = len(arrs)
        # the reference runs chunks of min(len, batch_size) (base.py:83); results are identical
        # for any chunking because every image is independent, so use the larger device chunk
        step = max(1, min(n, self.max_batch))
        out = torch.empty((n, self.embedding_dim), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            for s in range(0, n, step):
                part = arrs[s:s + step]
                dev = torch.empty((len(part), *part[0].shape), dtype=torch.uint8, device=self.device)
                if s:
                    torch.cuda.current_stream(self.device).synchronize()    # the pinned buffer is reused: its copies must have left
                self._upload(part, dev)                                      # pinned source, pieces overlapped with the gather
                self.forward_device(dev, out[s:s + step])
        return out.cpu().numpy()

    @torch.inference_mode()
    def extract_device(self, tiles_u8: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Device-resident variant: tiles uint8 [n, H, W, 3] in HBM -> float32 [n, D] in HBM."""
        n = tiles_u8.shape[0]
        if out is None:
            out = torch.empty((n, self.embedding_dim), dtype=torch.float32, device=tiles_u8.device)
        return self.forward_device(tiles_u8, out)

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
