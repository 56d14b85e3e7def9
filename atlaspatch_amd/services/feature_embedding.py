"""Feature embedding service (reference: services/feature_embedding.py:28-316).

Same public surface (``resolve_feature_dtype``, ``PatchFeatureEmbeddingService.embed_features /
embed_all``), same sequencing (extractor-major, slide-minor; one model resident at a time; lock
file per slide; skip feature sets that are already complete; failures collected, not raised).
For extractors on the native HIP path the per-slide work runs through the pinned tile ring
(``tile_ring.TileRing``) and the feature matrix is written in one bulk append; any other
extractor goes through the reference's buffered ``append_features`` loop.
"""
from __future__ import annotations

import logging
import os
import time
from pathlib import Path
from typing import Iterable, Optional

import numpy as np
import torch

from ..core.config import ExtractionConfig, FeatureExtractionConfig, OutputConfig
from ..core.models import ExtractionResult
from ..core.paths import patch_lock_path
from ..core.wsi.iwsi import IWSI
from ..encoders import build_default_registry
from ..encoders.base import HipViTFeatureExtractor
from ..encoders.custom import register_feature_extractors_from_module
from ..encoders.registry import PatchFeatureExtractorRegistry
from ..utils.features import get_existing_features
from ..utils.stages import stage
from .interfaces import FeatureEmbeddingService
from .storage import H5PatchWriter, read_coords

logger = logging.getLogger("atlaspatch_amd.feature_embedding_service")

_PRECISION = {"float32": torch.float32, "float16": torch.float16, "bfloat16": torch.bfloat16}


def resolve_feature_dtype(device: torch.device, precision: str) -> torch.dtype:
    """float16 on a CPU device falls back to float32 (feature_embedding.py:28-39)."""
    dtype = _PRECISION.get(precision, torch.float32)
    if device.type == "cpu" and dtype == torch.float16:
        logger.warning("float16 on CPU is unsupported in many ops; falling back to float32.")
        dtype = torch.float32
    return dtype


def _resize_tile(tile: np.ndarray, size: int) -> np.ndarray:
    """``cv2.resize(tile, (size, size))`` (INTER_LINEAR) for a level read that is not already ``patch_size``
    (feature_embedding.py:94-95), one tile through the device kernel: only the generic (plugin-encoder) loop uses
    this; the native path resizes whole ring batches on the device."""
    from ..utils.resample import INTER_LINEAR, cv2_resize_array
    return cv2_resize_array(tile, (size, size), INTER_LINEAR)


def _to_patch_size(tiles: torch.Tensor, ps: int) -> torch.Tensor:
    """Device batch at its read size -> ``[n, ps, ps, 3]`` with the reference's ``cv2.resize(patch, (ps, ps))``."""
    if tiles.shape[1] == ps and tiles.shape[2] == ps:
        return tiles
    from ..utils.resample import INTER_LINEAR, cv2_resize_device
    return cv2_resize_device(tiles, (ps, ps), INTER_LINEAR)


class PatchFeatureEmbeddingService(FeatureEmbeddingService):
    def __init__(self, extraction_cfg: ExtractionConfig, output_cfg: OutputConfig,
                 feature_cfg: FeatureExtractionConfig,
                 registry: Optional[PatchFeatureExtractorRegistry] = None, *,
                 keep_feature_blocks: Optional[bool] = None) -> None:
        self.cfg = extraction_cfg.validated()
        self.output_cfg = output_cfg.validated()
        self.feature_cfg = feature_cfg.validated()
        wanted = self.feature_cfg.device
        if wanted.startswith("cuda") and not torch.cuda.is_available():
            logger.warning("Feature extraction requested on CUDA but unavailable; using CPU instead.")
            wanted = "cpu"
        self.device = torch.device(wanted)
        self.dtype = resolve_feature_dtype(self.device, self.feature_cfg.precision)
        self.registry = registry or build_default_registry(device=self.device, dtype=self.dtype,
                                                           num_workers=self.feature_cfg.num_workers)
        if registry is None:
            for plugin in self.feature_cfg.plugins:
                register_feature_extractors_from_module(plugin, registry=self.registry, device=self.device,
                                                        dtype=self.dtype,
                                                        num_workers=self.feature_cfg.num_workers)
        self.extractor_names = [name.lower() for name in self.feature_cfg.extractors]
        self._seen: dict[Path, tuple[int | None, set[str]]] = {}
        self._ring = None
        # (h5 path, extractor) -> float32 [N, D] computed in this run; kept only when the rank-sharded gather will use
        # them (--gather-features / ATLASPATCH_GATHER_FEATURES), so that the all-gather does not read the matrices back from disk
        self.feature_blocks: dict = {}
        from ..utils.env import env_flag
        self._keep_blocks = env_flag("ATLASPATCH_GATHER_FEATURES") if keep_feature_blocks is None else bool(keep_feature_blocks)
        # embed_all pipelines slides: slide k's feature matrix lands in one of two grow-only pinned buffers and is written to
        # its H5 by a writer thread while slide k + 1 embeds into the other; the first encoder can be built on a side thread
        # while phase 1 (segmentation + coordinates) still runs (prefetch_extractor)
        self._result_bufs: list = [None, None]
        self._result_busy: list = [None, None]
        self._result_next = 0
        self._h5_pool = None
        self._prefetched = None

    # ------------------------------------------------------------------ bookkeeping
    def _existing(self, h5_path: Path, expected_total: int | None = None) -> set[str]:
        key = Path(h5_path).resolve()
        hit = self._seen.get(key)
        if hit is not None and (expected_total is None or hit[0] == expected_total):
            return set(hit[1])
        found = get_existing_features(key, expected_total=expected_total)
        self._seen[key] = (expected_total, set(found))
        return set(found)

    def _remember(self, h5_path: Path, name: str, total: int) -> None:
        key = Path(h5_path).resolve()
        _, have = self._seen.get(key, (total, set()))
        self._seen[key] = (total, set(have) | {name.lower()})

    def _stamp(self, result: ExtractionResult) -> ExtractionResult:
        have = sorted(self._existing(result.h5_path, expected_total=result.num_patches))
        if have:
            result.metadata["feature_sets"] = have
        return result

    def _lock(self, slide):
        path = patch_lock_path(slide, self.output_cfg, self.cfg)
        path.parent.mkdir(parents=True, exist_ok=True)
        note = f"pid={os.getpid()},time={int(time.time())},slide={slide.path},phase=features"
        try:
            fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_WRONLY)
        except FileExistsError:
            return None, path
        except Exception as exc:  # noqa: BLE001
            raise RuntimeError(f"Failed to create feature lock {path}: {exc}") from exc
        os.write(fd, note.encode())
        os.fsync(fd)
        return fd, path

    @staticmethod
    def _unlock(fd, path) -> None:
        if fd is not None:
            try:
                os.close(fd)
            except OSError:
                pass
        if path is not None:
            try:
                path.unlink()
            except OSError:
                pass

    # ------------------------------------------------------------------ tile access
    def _read_tile(self, wsi: IWSI):
        ps = self.cfg.patch_size

        def read(x, y, rw, rh, lv):
            tile = wsi.extract((x, y), lv=lv, wh=(rw, rh), mode="array")
            if tile.shape[0] != ps or tile.shape[1] != ps:
                tile = _resize_tile(tile, ps)
            return tile

        return read

    def _entries(self, wsi: IWSI, result: ExtractionResult) -> Iterable[tuple]:
        read = self._read_tile(wsi)
        for x, y, rw, rh, lv in read_coords(result.h5_path).tolist():
            yield x, y, rw, rh, lv, read(x, y, rw, rh, lv)

    # ------------------------------------------------------------------ public API
    def embed_features(self, result: ExtractionResult, *, wsi: IWSI) -> ExtractionResult:
        if not self.extractor_names:
            return result
        extractor = self.registry.create(self.extractor_names[0])
        try:
            return self._embed_with_extractor(result=result, wsi=wsi, extractor=extractor)
        finally:
            try:
                extractor.cleanup()
            except Exception:  # noqa: BLE001
                pass

    def _writer(self, result: ExtractionResult, wsi: IWSI) -> H5PatchWriter:
        step = self.cfg.step_size or self.cfg.patch_size
        return H5PatchWriter(chunk_rows=self.cfg.write_batch, patch_size=self.cfg.patch_size,
                             patch_size_level0=result.patch_size_level0 or 0,
                             level0_mag=int(wsi.mag) if wsi.mag is not None else 0,
                             target_mag=self.cfg.target_magnification, level0_wh=wsi.get_size(lv=0),
                             overlap=max(0, int(self.cfg.patch_size) - int(step)),
                             slide_stem=result.slide.stem, wsi_path=str(wsi.path))

    def _embed_with_extractor(self, *, result: ExtractionResult, wsi: IWSI, extractor, defer=None) -> ExtractionResult:
        """``defer`` (embed_all only): a list that receives ``(slide, future)`` when the H5 write of this slide's feature
        matrix was handed to the writer thread; the lock is released and the bookkeeping done there, after the write."""
        fd, lock_path = self._lock(result.slide)
        if fd is None:
            logger.info("Skipping feature embedding for %s (locked by another process).", result.slide.path.name)
            return self._stamp(result)
        deferred = False
        try:
            if extractor.name.lower() in self._existing(result.h5_path, expected_total=result.num_patches):
                logger.info("Skipping feature embedding for %s (feature '%s' already exists).",
                            result.slide.path.name, extractor.name)
                return self._stamp(result)
            attrs = {"name": extractor.name, "embedding_dim": extractor.embedding_dim}
            writer = self._writer(result, wsi)
            if isinstance(extractor, HipViTFeatureExtractor):
                slot = self._result_slot() if defer is not None else None
                with stage("embed_matrix"):
                    feats = self.embed_matrix(result, wsi, extractor, slot=slot)
                if self._keep_blocks:
                    self.feature_blocks[(str(result.h5_path), extractor.name.lower())] = np.array(feats, copy=True)

                def write_and_finish(feats=feats):
                    try:
                        with stage("h5_features"):
                            writer.append_feature_matrix(output_path=result.h5_path, feature_name=extractor.name,
                                                         features=feats, feature_attrs=attrs,
                                                         feature_batch=self.feature_cfg.batch_size,
                                                         expected_total=result.num_patches)
                        self._remember(result.h5_path, extractor.name, result.num_patches)
                        self._finish(result, extractor.name)
                    finally:
                        self._unlock(fd, lock_path)

                if defer is not None:
                    if self._h5_pool is None:
                        import concurrent.futures as futures
                        self._h5_pool = futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix="h5-features")
                    fut = self._h5_pool.submit(write_and_finish)
                    self._result_busy[slot] = fut
                    defer.append((result.slide, fut))
                    deferred = True
                    return result
                deferred = True                      # write_and_finish releases the lock itself
                write_and_finish()
                return result
            writer.append_features(output_path=result.h5_path, entries=self._entries(wsi, result),
                                   feature_name=extractor.name,
                                   feature_fn=lambda patches, ex=extractor: ex.extract_batch(
                                       patches, batch_size=self.feature_cfg.batch_size),
                                   feature_attrs=attrs, feature_batch=self.feature_cfg.batch_size,
                                   expected_total=result.num_patches)
            self._remember(result.h5_path, extractor.name, result.num_patches)
        finally:
            if not deferred:
                self._unlock(fd, lock_path)
        return self._finish(result, extractor.name)

    def _finish(self, result: ExtractionResult, name: str) -> ExtractionResult:
        known = result.metadata.get("feature_sets", [])
        merged = list(dict.fromkeys([*known, name])) if isinstance(known, list) else [name]
        result.metadata["feature_sets"] = merged
        return self._stamp(result)

    def _result_slot(self) -> int:
        """Index of the pinned result buffer the next slide embeds into; waits for the writer that still reads it."""
        slot = self._result_next
        self._result_next = 1 - slot
        busy = self._result_busy[slot]
        if busy is not None:
            try:
                busy.result()
            except Exception:  # noqa: BLE001 -- reported through the deferred list
                pass
            self._result_busy[slot] = None
        return slot

    def _result_buffer(self, slot: int, rows: int, dim: int) -> torch.Tensor:
        buf = self._result_bufs[slot]
        if buf is None or buf.shape[0] < rows or buf.shape[1] != dim:
            self._result_bufs[slot] = None
            buf = self._result_bufs[slot] = torch.empty((max(rows, 2048), dim), dtype=torch.float32, pin_memory=True)
        return buf

    def prefetch_extractor(self) -> None:
        """Build the first encoder (checkpoint load, upload, weight folding: ~0.1-0.2 s) on a side thread; embed_all picks it
        up.  Called by the CLI before phase 1 so that the build overlaps segmentation + coordinates."""
        if self._prefetched is not None or not self.extractor_names:
            return
        import concurrent.futures as futures
        name = self.extractor_names[0]
        device_index = torch.cuda.current_device() if (self.device.type == "cuda" and torch.cuda.is_available()) else None

        def build():
            if device_index is not None:
                torch.cuda.set_device(device_index)          # the HIP device is per thread
            # the checkpoint load runs unlocked; HipViT takes HIP_CAPTURE_LOCK itself, around its device uploads only, so a
            # SAM2 graph capture waits for the upload and not for torch.load / a hub download
            with stage("encoder_create"):
                return self.registry.create(name)

        pool = futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix="encoder")
        self._prefetched = (name, pool.submit(build), pool)

    def _create_extractor(self, name: str):
        if self._prefetched is not None and self._prefetched[0] == name:
            _, fut, pool = self._prefetched
            self._prefetched = None
            try:
                return fut.result()
            finally:
                pool.shutdown(wait=False)
        with stage("encoder_create"):
            return self.registry.create(name)

    # one ring batch: 2048 tiles quantise best onto the 256 persistent GEMM workgroups; tiles that are
    # read larger than the patch size (resized on the device) shrink the batch so that a pinned slot stays <= ~0.8 GB
    _SLOT_BYTES = 2048 * 256 * 256 * 3 * 2

    def _ring_batch(self, extractor, tile_hw) -> int:
        cap = min(2048, max(1, int(getattr(extractor, "max_batch", 1024))))
        by_bytes = max(64, self._SLOT_BYTES // (tile_hw[0] * tile_hw[1] * 3))
        return max(1, min(cap, by_bytes))

    def embed_matrix(self, result: ExtractionResult, wsi: IWSI, extractor: HipViTFeatureExtractor, *, slot=None) -> np.ndarray:
        """float32 [N, D] for one slide through the pinned tile ring (device pipeline).  Tiles cross the ring at
        their read size; ``cv2.resize`` to ``patch_size`` (feature_embedding.py:94-95) runs on the device."""
        from .tile_ring import TileRing
        coords = read_coords(result.h5_path)
        ps = int(self.cfg.patch_size)
        if coords.shape[0] == 0:
            return np.empty((0, extractor.embedding_dim), dtype=np.float32)
        rw, rh = int(coords[0, 2]), int(coords[0, 3])
        if np.any(coords[:, 2] != rw) or np.any(coords[:, 3] != rh):
            raise ValueError("coords rows of one slide must share one read size")
        tile_hw = (rh, rw)
        batch = self._ring_batch(extractor, tile_hw)
        pinned = self._result_buffer(slot, coords.shape[0], extractor.embedding_dim) if slot is not None else None
        feats = self._embed_device_source(coords, wsi, extractor, batch, pinned)
        if feats is not None:
            return feats
        # sized once per (patch size, read size): short slides run as partial batches, the pinned slots are kept
        ring = self._ring
        if ring is None or ring.batch != batch or ring.ps != ps or (ring.th, ring.tw) != tile_hw or \
                ring.device != extractor.device:
            if ring is not None:
                ring.close()
            with stage("ring_build"):
                self._ring = ring = TileRing(device=extractor.device, batch=batch, patch_size=ps, tile_hw=tile_hw,
                                             slots=3, workers=max(1, self.feature_cfg.num_workers))

        def read(x, y, rw_, rh_, lv):
            return wsi.extract((x, y), lv=lv, wh=(rw_, rh_), mode="array")

        with torch.cuda.device(extractor.device):
            return ring.run(coords, read,
                            lambda tiles, out: extractor.forward_device(_to_patch_size(tiles, ps), out),
                            extractor.embedding_dim, read_chunk=getattr(wsi, "read_tiles_into", None), out_host=pinned)

    def _embed_device_source(self, coords: np.ndarray, wsi: IWSI, extractor, batch: int, pinned=None):
        """Backends that can materialise tiles in HBM themselves (``extract_batch_device``, e.g. the synthetic
        slide) skip the host ring: tiles never cross PCIe.  ATLASPATCH_HOST_TILES=1 forces the ring.  Features leave
        the device batch by batch (asynchronous copies into pinned memory, hidden behind the next batch's forward)."""
        source = getattr(wsi, "extract_batch_device", None)
        if source is None or os.environ.get("ATLASPATCH_HOST_TILES"):
            return None
        n = int(coords.shape[0])
        ps = int(self.cfg.patch_size)
        dim = extractor.embedding_dim
        host = pinned if pinned is not None else torch.empty((n, dim), dtype=torch.float32, pin_memory=True)
        outs = [torch.empty((batch, dim), dtype=torch.float32, device=extractor.device) for _ in range(2)]
        with torch.cuda.device(extractor.device):
            for i, lo in enumerate(range(0, n, batch)):
                tiles = source(coords[lo:lo + batch], extractor.device, ps)
                if tiles is None:
                    torch.cuda.synchronize(extractor.device)
                    return None
                out = outs[i & 1][:tiles.shape[0]]           # stream order protects the buffer two batches back
                extractor.forward_device(_to_patch_size(tiles, ps), out)
                host[lo:lo + tiles.shape[0]].copy_(out, non_blocking=True)
            torch.cuda.synchronize(extractor.device)
        return host[:n].numpy()

    def embed_all(self, results: list[ExtractionResult], *, wsi_loader, progress=None) -> list[tuple]:
        failures: list[tuple] = []
        todo: dict[Path, set[str]] = {}
        already = 0
        for res in results:
            have = self._existing(res.h5_path, expected_total=res.num_patches)
            missing = [n for n in self.extractor_names if n not in have]
            if missing:
                todo[res.h5_path] = set(missing)
            else:
                self._stamp(res)
            already += len(self.extractor_names) - len(missing)
        if progress and already:
            progress.update(already)

        for name in self.extractor_names:
            try:
                extractor = self._create_extractor(name)
            except Exception as exc:  # noqa: BLE001
                for res in results:
                    if name in todo.get(res.h5_path, ()):
                        failures.append((res.slide, exc))
                        if progress:
                            progress.update(1)
                continue
            deferred: list = []
            try:
                for res in results:
                    if name not in todo.get(res.h5_path, ()):
                        continue
                    wsi = None
                    try:
                        if extractor.name.lower() not in self._existing(res.h5_path, expected_total=res.num_patches):
                            wsi = wsi_loader.open(res.slide)
                            self._embed_with_extractor(result=res, wsi=wsi, extractor=extractor, defer=deferred)
                        self._stamp(res)
                    except Exception as exc:  # noqa: BLE001
                        failures.append((res.slide, exc))
                    finally:
                        if wsi is not None:
                            try:
                                wsi.cleanup()
                            except Exception:  # noqa: BLE001
                                pass
                    if progress:
                        progress.update(1)
            finally:
                # the writer thread still holds this extractor's last matrices: every H5 of this extractor is complete
                # (and its lock released) before the next extractor -- or the caller -- looks at the files again
                for slide, fut in deferred:
                    try:
                        fut.result()
                    except Exception as exc:  # noqa: BLE001
                        failures.append((slide, exc))
                self._result_busy = [None, None]
                try:
                    extractor.cleanup()
                except Exception:  # noqa: BLE001
                    pass
        if self._ring is not None:
            self._ring.close()
            self._ring = None
        if self._h5_pool is not None:
            self._h5_pool.shutdown(wait=True)
            self._h5_pool = None
        return failures
