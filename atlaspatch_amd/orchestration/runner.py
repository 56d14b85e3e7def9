"""Phase-1 orchestration: discover -> (skip | reuse | lock) -> segment -> coords (reference:
orchestration/runner.py:39-306, orchestration/parallel.py).

The reference hides the slow Python grid scan behind a thread pool with an in-flight tracker;
here coordinate extraction is a few milliseconds of GPU work per slide, so slides are processed
in order within each segmentation batch (results therefore come back in slide order, which the
reference only guarantees per batch).  Everything observable is kept: per-slide lock files
(``O_CREAT | O_EXCL``), ``--skip-existing`` reuse of an H5 whose feature sets are incomplete,
per-slide failure isolation, ``(results, failures)`` return value.

MI355X addition: ``rank`` / ``world_size`` shard the slide list one-slide-per-rank
(``slides[rank::world_size]``) -- the device/rank dispatch of the north star.
"""
from __future__ import annotations

import logging
import os
import time
from pathlib import Path
from typing import Any, Iterable, Sequence

from tqdm import tqdm

from ..core.config import AppConfig
from ..core.models import ExtractionResult, Slide
from ..core.paths import find_existing_patch, patch_lock_path
from ..services.interfaces import (ExtractionService, MPPResolver, SegmentationService,
                                   VisualizationService, WSILoader)
from ..utils.features import missing_features
from ..utils.h5 import h5
from ..utils.params import get_wsi_files

from ..utils.stages import stage

logger = logging.getLogger("atlaspatch_amd.runner")


def _batches(items: Sequence[Slide], size: int) -> Iterable[Sequence[Slide]]:
    for start in range(0, len(items), size):
        yield items[start:start + size]


class ProcessingRunner:
    def __init__(self, config: AppConfig, segmentation: SegmentationService, extractor: ExtractionService,
                 visualizer: VisualizationService | None, mpp_resolver: MPPResolver, wsi_loader: WSILoader, *,
                 show_progress: bool = False, rank: int = 0, world_size: int = 1) -> None:
        self.config = config.validated()
        self.segmentation = segmentation
        self.extractor = extractor
        self.visualizer = visualizer
        self.mpp_resolver = mpp_resolver
        self.wsi_loader = wsi_loader
        self.show_progress = show_progress
        self.rank, self.world_size = int(rank), max(1, int(world_size))
        self._device_index = None

    # ------------------------------------------------------------------ discovery
    def discover_slides(self) -> list[Slide]:
        files = get_wsi_files(str(self.config.processing.input_path), recursive=self.config.processing.recursive)
        slides = [Slide(path=Path(f)) for f in files]
        return slides[self.rank::self.world_size]

    def _with_mpp(self, slides: list[Slide]) -> list[Slide]:
        return [Slide(path=s.path, mpp=self.mpp_resolver.resolve(s), backend=s.backend) for s in slides]

    # ------------------------------------------------------------------ skip / reuse
    def _existing_result(self, slide: Slide, h5_path: Path) -> ExtractionResult | None:
        try:
            with h5.File(h5_path, "r") as f:
                total = f.attrs.get("num_patches")
                if total is None and "coords" in f:
                    total = f["coords"].shape[0]
                ps0 = f.attrs.get("patch_size_level0")
        except Exception as exc:  # noqa: BLE001
            logger.warning("Failed to read existing output for %s; will reprocess. Error: %s", slide.path.name, exc)
            return None
        if total is None or int(total) <= 0:
            return None
        meta: dict[str, Any] = {}
        return ExtractionResult(slide=slide, h5_path=h5_path, num_patches=int(total),
                                patch_size_level0=int(ps0) if ps0 is not None else None, metadata=meta)

    def _handled_as_existing(self, slide: Slide, results: list[ExtractionResult], tick) -> bool:
        if not self.config.output.skip_existing:
            return False
        path = find_existing_patch(slide, self.config.output, self.config.extraction)
        if path is None:
            return False
        feats = self.config.features
        if feats is None or not feats.extractors:
            logger.info("Skipping %s (already processed).", slide.path.name)
            tick()
            return True
        prior = self._existing_result(slide, path)
        if prior is None:
            logger.info("Existing output invalid for %s; reprocessing.", slide.path.name)
            return False
        gaps = missing_features(path, feats.extractors, expected_total=prior.num_patches)
        if gaps:
            results.append(prior)
            logger.info("Reusing existing patches for %s; missing features: %s", slide.path.name, ", ".join(gaps))
        else:
            logger.info("Skipping %s (features complete).", slide.path.name)
        tick()
        return True

    # ------------------------------------------------------------------ locks
    def _acquire_lock(self, slide: Slide):
        path = patch_lock_path(slide, self.config.output, self.config.extraction)
        path.parent.mkdir(parents=True, exist_ok=True)
        try:
            fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_WRONLY)
        except FileExistsError:
            return None, path
        except Exception as exc:  # noqa: BLE001
            raise RuntimeError(f"Failed to create lock {path}: {exc}") from exc
        os.write(fd, f"pid={os.getpid()},time={int(time.time())},slide={slide.path}".encode())
        os.fsync(fd)
        return fd, path

    @staticmethod
    def _release_lock(fd, path: Path) -> None:
        if fd is not None:
            try:
                os.close(fd)
            except OSError:
                pass
        try:
            path.unlink()
        except OSError:
            pass

    # ------------------------------------------------------------------ main loop
    def run(self):
        slides = self._with_mpp(self.discover_slides())
        if not slides:
            logger.warning("No slides found to process.")
            return [], []
        results: list[ExtractionResult] = []
        failures: list[tuple] = []
        bar = tqdm(total=len(slides), disable=not self.show_progress, desc="Processing slides")
        tick = (lambda: bar.update(1)) if self.show_progress else (lambda: None)
        import concurrent.futures as futures
        import torch
        self._device_index = torch.cuda.current_device() if torch.cuda.is_available() else None
        workers = self.config.extraction.workers or min(8, os.cpu_count() or 1)
        open_cap = max(1, int(self.config.extraction.max_open_slides or 200))
        pool = futures.ThreadPoolExecutor(max_workers=max(1, int(workers)), thread_name_prefix="coords")
        inflight: list = []
        # cohorts: the per-slide coords files are written by helper processes (one libhdf5 lock per process instead of one for
        # all coordinate workers); started in the background now, used as soon as they answer (services/h5_writer_proc.py)
        if len(slides) >= 8 and hasattr(self.extractor, "h5_pool"):
            from ..services.h5_writer_proc import shared_pool
            h5_pool = shared_pool()
            if h5_pool is not None:
                h5_pool.prestart()
                self.extractor.h5_pool = h5_pool

        def drain(item):
            slide, fut = item
            try:
                results.append(fut.result())
            except Exception as exc:  # noqa: BLE001
                failures.append((slide, exc))
                logger.error("Extraction failed for %s: %s", slide.path.name, exc)
            tick()

        def open_group(group):
            opened = []
            for slide in group:
                if self._handled_as_existing(slide, results, tick):
                    continue
                fd, lock_path = self._acquire_lock(slide)
                if fd is None:
                    logger.info("Skipping %s (locked by another process).", slide.path.name)
                    tick()
                    continue
                try:
                    with stage("open_slide"):
                        opened.append((slide, self.wsi_loader.open(slide), fd, lock_path))
                except Exception as exc:  # noqa: BLE001
                    failures.append((slide, exc))
                    logger.error("Failed to open %s: %s", slide.path.name, exc)
                    self._release_lock(fd, lock_path)
                    tick()
            return opened

        def segmentation_failed(opened, exc):
            for slide, wsi, fd, lock_path in opened:
                failures.append((slide, exc))
                logger.error("Segmentation failed for %s: %s", slide.path.name, exc)
                self._close(wsi)
                self._release_lock(fd, lock_path)
                tick()

        def submit(opened, masks):
            # coordinates + H5 of a segmented slide run on a worker pool (the reference's PatchExtractionExecutor:
            # orchestration/parallel.py:105-160, `--patch-workers` threads, at most `--max-open-slides` slides open) while
            # this thread goes on to segment the next group; results keep the submission order
            for (slide, wsi, fd, lock_path), mask in zip(opened, masks):
                while len(inflight) >= open_cap:
                    drain(inflight.pop(0))
                inflight.append((slide, pool.submit(self._extract_one, slide, wsi, fd, lock_path, mask)))

        groups = _batches(slides, max(1, self.config.segmentation.batch_size))
        seg = self.segmentation
        if os.environ.get("ATLASPATCH_SEG_PIPELINE", "1") != "0" and hasattr(seg, "prepare_input") and hasattr(seg, "segment_prepared"):
            # Pipelined: a helper thread prepares group k + 1's network inputs (level read, cv2 / Pillow resizes) while
            # this thread runs group k's forwards.  Same calls on the same inputs in the same order per slide: same masks.
            prep = futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix="seg-input")
            read_workers = max(1, min(8, max(1, self.config.segmentation.batch_size), os.cpu_count() or 8))
            readers = futures.ThreadPoolExecutor(max_workers=read_workers, thread_name_prefix="thumb") if read_workers > 1 else None
            warmed = [not hasattr(seg, "warm_up")]

            def warm_up_once():
                # every hipGraph the run needs (full groups and the remainder group) is captured before the first group's
                # input is prepared: no coordinate worker and no input thread is issuing HIP calls of its own yet.  (Groups
                # that lose slides to skip-existing / lock files may still need another size: captured when met, under
                # capture_error_mode="thread_local" as before.)
                warmed[0] = True
                B = max(1, self.config.segmentation.batch_size)
                try:
                    seg.warm_up({min(B, len(slides)), len(slides) % B})
                except Exception as exc:  # noqa: BLE001  (no weights / no device: reported per slide by the forward itself)
                    logger.debug("segmentation warm-up skipped: %s", exc)

            def prepare_one(w):
                if self._device_index is not None:
                    torch.cuda.set_device(self._device_index)
                return seg.prepare_input(w)

            def prepare(opened):
                # the level reads of a group overlap like the reference's 8 thumbnail threads (segmentation.py:216-220)
                if self._device_index is not None:
                    torch.cuda.set_device(self._device_index)
                wsis = [w for _, w, _, _ in opened]
                if readers is None or len(wsis) == 1:
                    return [seg.prepare_input(w) for w in wsis]
                return list(readers.map(prepare_one, wsis))

            def finish(item):
                opened, fut = item
                try:
                    with stage("segmentation"):
                        thumbs = fut.result()
                        masks = (seg.segment_prepared_batch(thumbs) if len(thumbs) > 1 and hasattr(seg, "segment_prepared_batch")
                                 else [seg.segment_prepared(t) for t in thumbs])
                except Exception as exc:  # noqa: BLE001
                    segmentation_failed(opened, exc)
                    return
                submit(opened, masks)

            pending = None
            for group in groups:
                opened = open_group(group)
                if opened and not warmed[0]:
                    warm_up_once()
                nxt = (opened, prep.submit(prepare, opened)) if opened else None
                if pending is not None:
                    finish(pending)
                pending = nxt
            if pending is not None:
                finish(pending)
            prep.shutdown(wait=True)
            if readers is not None:
                readers.shutdown(wait=True)
        else:
            for group in groups:
                opened = open_group(group)
                if not opened:
                    continue
                try:
                    wsis = [w for _, w, _, _ in opened]
                    with stage("segmentation"):
                        masks = (seg.segment_batch(wsis) if len(wsis) > 1 else [seg.segment_thumbnail(wsis[0])])
                except Exception as exc:  # noqa: BLE001
                    segmentation_failed(opened, exc)
                    continue
                submit(opened, masks)
        for item in inflight:
            drain(item)
        pool.shutdown(wait=True)
        bar.close()
        return results, failures

    def _extract_one(self, slide, wsi, fd, lock_path, mask):
        try:
            if self._device_index is not None:       # the HIP device is per thread: follow the rank's device, not device 0
                import torch
                torch.cuda.set_device(self._device_index)
            with stage("coords_and_h5"):
                result = self.extractor.extract(wsi, mask.data, slide=slide)
            if self.visualizer is not None:
                with stage("visualize"):
                    self.visualizer.visualize(result, wsi=wsi, mask=mask.data)
            return result
        finally:
            self._close(wsi)
            self._release_lock(fd, lock_path)

    @staticmethod
    def _close(wsi) -> None:
        try:
            wsi.cleanup()
        except Exception:  # noqa: BLE001
            pass
