"""``atlaspatch`` command line: ``process`` and ``segment-and-get-coords`` with the reference's option
names and defaults (/root/reference/atlas_patch/cli.py:54-192,466-693), driving the MI355X path.

Quirks kept on purpose: ``--tissue-thresh`` defaults to 0.0 here although the dataclass default is
0.01; ``--feature-precision`` defaults to float16 although the dataclass default is float32;
``--skip-existing`` is on by default; the group reports version 0.2.0.

Differences (stated, not hidden): ``--no-fast-mode`` and ``--save-images`` run on the device path; the
``--visualize-*`` overlays are the reference's Pillow calls except that contour outlines are drawn with
``ImageDraw.line`` instead of ``cv2.polylines``; segmentation uses the analytic mask for ``.synth`` slides and
the HIP SAM2 path (``services/sam2_hip.py``) for real slides.  When launched under ``torch.distributed.run`` slides are sharded one per rank.
"""
from __future__ import annotations

import logging
import os
import sys
from pathlib import Path

import click
import torch
from tqdm import tqdm

from .core.config import (AppConfig, ExtractionConfig, FeatureExtractionConfig, OutputConfig,
                          ProcessingConfig, SegmentationConfig, VisualizationConfig)
from .encoders import build_default_registry, register_feature_extractors_from_module
from .orchestration.dispatch import env_rank_world
from .orchestration.runner import ProcessingRunner
from .services.extraction import PatchExtractionService
from .services.feature_embedding import PatchFeatureEmbeddingService, resolve_feature_dtype
from .services.mpp import CSVMPPResolver
from .services.segmentation import AnalyticSegmentationService, SAM2SegmentationService
from .services.visualization import DefaultVisualizationService
from .services.wsi_loader import DefaultWSILoader
from .utils.features import parse_feature_list
from .utils.stages import stage
from .utils.params import get_wsi_files

logging.basicConfig(level=logging.WARNING, format="%(asctime)s | %(levelname)s | %(name)s | %(message)s")
logger = logging.getLogger("atlaspatch_amd.cli")

FEATURE_EXTRACTOR_CHOICES = build_default_registry(device="cpu").available()

_COMMON = [
    click.argument("wsi_path", type=click.Path(exists=True)),
    click.option("--output", "-o", type=click.Path(), required=True, help="Output directory root for generated artifacts."),
    click.option("--patch-size", type=int, required=True, help="Patch size at target magnification."),
    click.option("--step-size", type=int, default=None, help="Stride between patches; defaults to patch size when omitted."),
    click.option("--target-mag", type=click.IntRange(1, 120), required=True, help="Target magnification (e.g., 20, 40)."),
    click.option("--device", type=str, default="cuda", show_default=True, help="Segmentation device (e.g., cuda, cuda:0, cpu)."),
    click.option("--tissue-thresh", type=float, default=0.0, show_default=True, help="Minimum tissue area fraction."),
    click.option("--white-thresh", type=int, default=15, show_default=True, help="Saturation threshold for white filtering."),
    click.option("--black-thresh", type=int, default=50, show_default=True, help="RGB threshold for black filtering."),
    click.option("--seg-batch-size", type=int, default=1, show_default=True, help="Segmentation batch."),
    click.option("--write-batch", type=int, default=8192, show_default=True, help="HDF5 write batch."),
    click.option("--patch-workers", type=int, default=None, show_default=True,
                 help="Parallel worker threads for per-slide patch extraction; defaults to CPU count."),
    click.option("--max-open-slides", type=int, default=200, show_default=True,
                 help="Upper bound on simultaneously open slides (segmentation + extraction)."),
    click.option("--fast-mode/--no-fast-mode", default=True, show_default=True,
                 help="fast-mode skips per-patch content filtering; use --no-fast-mode to enable filtering."),
    click.option("--save-images", is_flag=True, help="Export individual patch PNGs."),
    click.option("--visualize-grids", is_flag=True, help="Render patch grid overlay."),
    click.option("--visualize-mask", is_flag=True, help="Render predicted mask overlay."),
    click.option("--visualize-contours", is_flag=True, help="Render contour overlay."),
    click.option("--recursive", is_flag=True, help="Recursively search directories for WSIs."),
    click.option("--mpp-csv", type=click.Path(exists=True), default=None, help="CSV with custom MPP."),
    click.option("--skip-existing/--force", default=True, show_default=True, help="Skip existing H5."),
    click.option("--verbose", "-v", is_flag=True, help="Enable debug logging."),
]

_FEATURE = [
    click.option("--feature-device", type=str, default=None,
                 help="Device for feature extraction; e.g. cuda, cuda:0, cpu. Defaults to --device."),
    click.option("--feature-extractors", required=True, type=str,
                 help="Space/comma separated feature extractors to run (available: "
                      + ", ".join(FEATURE_EXTRACTOR_CHOICES) + "; add more via --feature-plugin)."),
    click.option("--feature-batch-size", type=int, default=32, show_default=True,
                 help="Batch size used when embedding patches."),
    click.option("--feature-num-workers", type=int, default=4, show_default=True,
                 help="DataLoader worker count for feature extraction."),
    click.option("--feature-precision", type=click.Choice(["float32", "float16", "bfloat16"], case_sensitive=False),
                 default="float16", show_default=True, help="Computation precision for feature extraction."),
    click.option("--feature-plugin", "feature_plugins", type=click.Path(exists=True), multiple=True,
                 help="Path(s) to Python modules that register custom feature extractors via "
                      "register_feature_extractors(registry, device, dtype, num_workers)."),
    # MI355X addition (not a reference option): under torch.distributed.run, slides are sharded one per rank and this adds the
    # ONE collective of the path -- an RCCL all-gather-v of the per-slide feature matrices -> <out>/features_all/<extractor>.npy
    click.option("--gather-features/--no-gather-features", "gather_features", default=None,
                 help="Multi-GPU runs only: reassemble the rank-sharded feature matrices on rank 0 with one RCCL all-gather "
                      "(<out>/features_all/<extractor>.npy + index.json). Default: the ATLASPATCH_GATHER_FEATURES environment variable."),
]


def _decorate(func, options):
    for option in reversed(options):
        func = option(func)
    return func


def _pick_segmenter(app_cfg: AppConfig):
    files = get_wsi_files(str(app_cfg.processing.input_path), recursive=app_cfg.processing.recursive)
    return _segmenter_for(files, app_cfg.segmentation)


def _run_pipeline(*, wsi_path, output, patch_size, step_size, target_mag, device, tissue_thresh, white_thresh,
                  black_thresh, seg_batch_size, write_batch, patch_workers, max_open_slides, fast_mode, save_images,
                  visualize_grids, visualize_mask, visualize_contours, recursive, mpp_csv, skip_existing, verbose,
                  feature_cfg=None, registry=None, gather_features=None):
    logging.getLogger().setLevel(logging.DEBUG if verbose else logging.WARNING)
    seg_yaml = Path(__file__).resolve().parent / "configs" / "sam2.1_hiera_t.yaml"
    app_cfg = AppConfig(
        processing=ProcessingConfig(input_path=Path(wsi_path), recursive=recursive,
                                    mpp_csv=Path(mpp_csv) if mpp_csv else None),
        segmentation=SegmentationConfig(checkpoint_path=None, config_path=seg_yaml, device=device.lower(),
                                        batch_size=seg_batch_size),
        extraction=ExtractionConfig(patch_size=patch_size, step_size=step_size, target_magnification=target_mag,
                                    tissue_threshold=tissue_thresh, white_threshold=white_thresh,
                                    black_threshold=black_thresh, fast_mode=fast_mode, write_batch=write_batch,
                                    workers=patch_workers, max_open_slides=max_open_slides),
        output=OutputConfig(output_root=Path(output), save_images=save_images, visualize_grids=visualize_grids,
                            visualize_mask=visualize_mask, visualize_contours=visualize_contours,
                            skip_existing=skip_existing),
        visualization=VisualizationConfig(), features=feature_cfg, device=device.lower()).validated()

    rank, world, local_rank = env_rank_world()
    if gather_features is None:
        from .utils.env import env_flag
        gather_features = env_flag("ATLASPATCH_GATHER_FEATURES")
    gather_features = bool(gather_features) and world > 1
    if world > 1 and torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    segmenter = _pick_segmenter(app_cfg)
    loader = DefaultWSILoader()
    runner = ProcessingRunner(config=app_cfg, segmentation=segmenter,
                              extractor=PatchExtractionService(app_cfg.extraction, app_cfg.output),
                              visualizer=DefaultVisualizationService(app_cfg.output, app_cfg.extraction, app_cfg.visualization),
                              mpp_resolver=CSVMPPResolver(app_cfg.processing.mpp_csv), wsi_loader=loader,
                              show_progress=not verbose, rank=rank, world_size=world)
    service = None
    if app_cfg.features is not None:
        service = PatchFeatureEmbeddingService(app_cfg.extraction, app_cfg.output, app_cfg.features, registry=registry,
                                               keep_feature_blocks=gather_features)
        service.prefetch_extractor()         # the first encoder is built on a side thread while phase 1 runs
    try:
        with stage("phase1_segment_and_coords"):
            results, failures = runner.run()
    finally:
        segmenter.close()
    click.echo("Segmentation and patch coordinate extraction complete.")

    if service is not None:
        units = len(results) * len(app_cfg.features.extractors)
        bar = tqdm(total=units, desc="Feature embedding", disable=verbose or units == 0)
        try:
            with stage("phase2_embed_all"):
                failures.extend(service.embed_all(results, wsi_loader=loader, progress=bar))
        finally:
            bar.close()
        if gather_features:
            # MI355X addition (north star): one RCCL all-gather-v reassembles the rank-sharded feature matrices
            from .orchestration.dispatch import gather_run_features, init_process_group
            dev = torch.device("cuda", local_rank) if torch.cuda.is_available() else None
            init_process_group(device=dev)
            # every slide assigned to this rank whose H5 is on disk -- also those a --skip-existing rerun found complete
            # (they never enter `results`); slides without a usable feature set are skipped inside, never raised on
            from .core.paths import patch_h5_path
            # ... except slides that FAILED in this run (segmentation, coords or embedding): an H5 left by an earlier run would
            # otherwise contribute stale features with nothing in index.json to show it
            failed = {Path(slide.path).resolve() for slide, _ in failures}
            mine = [p for s, p in ((s, patch_h5_path(s, app_cfg.output, app_cfg.extraction)) for s in runner.discover_slides())
                    if p.exists() and Path(s.path).resolve() not in failed]
            try:
                gather_run_features(mine, [e.lower() for e in app_cfg.features.extractors], app_cfg.output.output_root,
                                    device=dev, cache=service.feature_blocks)
            finally:
                service.feature_blocks.clear()          # the per-slide float32 matrices are not needed past the gather
    return results, failures


def _echo_results(results, failures, verbose, feature_cfg) -> None:
    click.echo(f"Completed {len(results)} slide(s), failures: {len(failures)}")
    if verbose:
        for res in results:
            click.echo(f"[OK] {res.slide.path.name} -> {res.h5_path} ({res.num_patches} patches)")
    for slide, err in failures:
        click.echo(f"[FAIL] {slide.path.name}: {err}", err=True)


@click.group()
@click.version_option(version="0.2.0")
def cli():
    """AtlasPatch CLI (MI355X-native hot path)."""


def _segment_and_get_coords(**kw):
    verbose = kw["verbose"]
    results, failures = _run_pipeline(feature_cfg=None, **kw)
    _echo_results(results, failures, verbose, None)


segment_and_get_coords = cli.command(name="segment-and-get-coords",
                                     help="Segment, patchify, and optionally visualize WSI files.")(
    _decorate(_segment_and_get_coords, _COMMON))


def _process(*, feature_device, feature_extractors, feature_batch_size, feature_num_workers, feature_precision,
             feature_plugins, gather_features=None, **kw):
    feat_device = (feature_device or kw["device"]).lower()
    torch_device = torch.device(feat_device)
    dtype = resolve_feature_dtype(torch_device, feature_precision.lower())
    registry = build_default_registry(device=torch_device, num_workers=feature_num_workers, dtype=dtype)
    for plugin in feature_plugins:
        register_feature_extractors_from_module(plugin, registry=registry, device=torch_device, dtype=dtype,
                                                num_workers=feature_num_workers)
    names = parse_feature_list(feature_extractors, choices=registry.available())
    feature_cfg = FeatureExtractionConfig(extractors=names, batch_size=feature_batch_size, device=feat_device,
                                          num_workers=feature_num_workers, precision=feature_precision.lower(),
                                          plugins=[Path(p) for p in feature_plugins])
    results, failures = _run_pipeline(feature_cfg=feature_cfg, registry=registry, gather_features=gather_features, **kw)
    _echo_results(results, failures, kw["verbose"], feature_cfg)


process = cli.command(name="process",
                      help="Run segmentation, patch extraction, and feature embedding into a single H5.")(
    _decorate(_decorate(_process, _COMMON), _FEATURE))


def _segmenter_for(files, seg_cfg: SegmentationConfig):
    """Synthetic slides carry their own tissue geometry (analytic mask); everything else goes through SAM2 like the
    reference.  ATLASPATCH_SEGMENTER=sam2 forces the SAM2 path for synthetic slides too (timing / plumbing runs)."""
    forced = os.environ.get("ATLASPATCH_SEGMENTER", "auto").lower()
    if forced != "sam2" and all(Path(f).suffix.lower() == ".synth" for f in files):
        return AnalyticSegmentationService(seg_cfg.thumbnail_max)
    return SAM2SegmentationService(seg_cfg)


@cli.command(name="detect-tissue")
@click.argument("wsi_path", type=click.Path(exists=True))
@click.option("--output", "-o", type=click.Path(), required=True, help="Output directory root for generated artifacts.")
@click.option("--device", type=str, default="cuda", show_default=True, help="Segmentation device (e.g., cuda, cuda:0, cpu).")
@click.option("--seg-batch-size", type=click.IntRange(1, None), default=1, show_default=True,
              help="Segmentation batch size for thumbnail inference.")
@click.option("--recursive", is_flag=True, help="Recursively search directories for WSIs.")
@click.option("--mpp-csv", type=click.Path(exists=True), default=None, help="CSV with custom MPP.")
@click.option("--verbose", "-v", is_flag=True, help="Enable debug logging.")
def detect_tissue(wsi_path, output, device, seg_batch_size, recursive, mpp_csv, verbose):
    """Run tissue segmentation only and export mask overlays (reference: cli.py:329-438, 531-578)."""
    from .core.models import Slide
    from .services.visualization import visualize_mask_on_thumbnail
    logging.getLogger().setLevel(logging.DEBUG if verbose else logging.WARNING)
    proc = ProcessingConfig(input_path=Path(wsi_path), recursive=recursive, mpp_csv=Path(mpp_csv) if mpp_csv else None).validated()
    seg_yaml = Path(__file__).resolve().parent / "configs" / "sam2.1_hiera_t.yaml"
    seg_cfg = SegmentationConfig(checkpoint_path=None, config_path=seg_yaml, device=device.lower(),
                                 batch_size=seg_batch_size).validated()
    vis_cfg = VisualizationConfig().validated()
    files = get_wsi_files(str(proc.input_path), recursive=proc.recursive)
    vis_dir = Path(output) / "visualization"
    Path(output).mkdir(parents=True, exist_ok=True)
    resolver, loader = CSVMPPResolver(proc.mpp_csv), DefaultWSILoader()
    segmenter = _segmenter_for(files, seg_cfg)
    results, failures = [], []
    bar = tqdm(total=len(files), disable=verbose, desc="Tissue detection")

    def run_batch(batch):
        if not batch:
            return
        wsis = [w for _, w in batch]
        try:
            masks = segmenter.segment_batch(wsis) if len(wsis) > 1 else [segmenter.segment_thumbnail(wsis[0])]
        except Exception as exc:  # noqa: BLE001
            for slide, wsi in batch:
                failures.append((slide, exc))
                ProcessingRunner._close(wsi)
                bar.update(1)
            return
        for (slide, wsi), mask in zip(batch, masks):
            try:
                results.append((slide, visualize_mask_on_thumbnail(mask=mask.data, wsi=wsi, output_dir=vis_dir,
                                                                   thumbnail_size=vis_cfg.thumbnail_size)))
            except Exception as exc:  # noqa: BLE001
                failures.append((slide, exc))
            finally:
                ProcessingRunner._close(wsi)
            bar.update(1)

    try:
        batch = []
        for f in files:
            base = Slide(path=Path(f))
            slide = Slide(path=base.path, mpp=resolver.resolve(base), backend=base.backend)
            try:
                batch.append((slide, loader.open(slide)))
            except Exception as exc:  # noqa: BLE001
                failures.append((slide, exc))
                bar.update(1)
                continue
            if len(batch) >= seg_cfg.batch_size:
                run_batch(batch)
                batch = []
        run_batch(batch)
    finally:
        try:
            segmenter.close()
        finally:
            bar.close()
    click.echo(f"Created {len(results)} mask overlay(s), failures: {len(failures)}")
    if verbose:
        for slide, path in results:
            click.echo(f"[OK] {slide.path.name} -> {path}")
        for slide, err in failures:
            click.echo(f"[FAIL] {slide.path.name}: {err}", err=True)


@cli.command()
def info():
    """Display supported formats and output structure."""
    click.echo("Supported WSI formats (OpenSlide): .svs, .tif, .tiff, .ndpi, .vms, .vmu, .scn, .mrxs, .bif, .dcm")
    click.echo("Image formats: .png, .jpg, .jpeg, .bmp, .webp, .gif")
    click.echo("Synthetic slides: .synth (JSON descriptor, rendered on the fly)")
    click.echo("Outputs: HDF5 per slide under patches/<stem>.h5; optional PNGs under images/<stem>; "
               "visualizations under visualization/.")


def main():
    try:
        cli()
    except click.ClickException as exc:
        click.echo(f"Error: {exc}", err=True)
        sys.exit(1)
    except KeyboardInterrupt:
        click.echo("\nInterrupted by user", err=True)
        sys.exit(130)
    except Exception as exc:  # noqa: BLE001
        click.echo(f"Unexpected error: {exc}", err=True)
        sys.exit(1)


if __name__ == "__main__":
    main()
