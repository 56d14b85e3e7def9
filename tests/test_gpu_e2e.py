"""GPU end-to-end: the CLI `process` command on a synthetic slide -> H5, checked against the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu


def _make_slide(folder, name, **kw):
    spec = {"width": 12000, "height": 9000, "seed": 7, "mag": 20, "mpp": 0.5, "downsamples": [1, 4, 16]}
    spec.update(kw)
    path = os.path.join(folder, name)
    json.dump(spec, open(path, "w"))
    return path, spec


def test_cli_process_synthetic_slide(tmp_path, monkeypatch):
    from click.testing import CliRunner
    from atlaspatch_amd.cli import cli
    from atlaspatch_amd.core.wsi.synth_pixels import SynthSpec, analytic_mask, render_region
    from atlaspatch_amd.encoders.vit import ARCHS, random_canonical_state_dict
    from atlaspatch_amd.utils.h5 import h5
    from oracle import coords_oracle, vit_oracle

    monkeypatch.setenv("ATLASPATCH_RANDOM_INIT", "0")
    slide, raw = _make_slide(str(tmp_path), "s1.synth")
    out = tmp_path / "out"
    args = ["process", slide, "-o", str(out), "--patch-size", "256", "--target-mag", "20",
            "--feature-extractors", "vit_b_16", "--feature-precision", "float32", "--feature-num-workers", "4"]
    res = CliRunner().invoke(cli, args, catch_exceptions=False)
    assert res.exit_code == 0, res.output
    assert "failures: 0" in res.output
    h5_path = out / "patches" / "s1.h5"
    assert h5_path.exists() and not (out / "patches" / "s1.lock").exists()

    spec = SynthSpec(width=raw["width"], height=raw["height"], seed=raw["seed"])
    want_coords, _ = coords_oracle.coords_from_mask(
        analytic_mask(spec), level0_wh=(spec.width, spec.height), downsamples=[1.0, 4.0, 16.0], src_mag=20,
        tgt_mag=20, patch_size=256, step_size=None, tissue_thresh=0.0)          # CLI default thresh 0.0
    with h5.File(h5_path, "r") as f:
        coords = f["coords"][:]
        feats = f["features"]["vit_b_16"][:]
        assert f.attrs["num_patches"] == coords.shape[0] and f.attrs["patch_size"] == 256
        assert f["passports"][0].decode().startswith("s1__x")
    assert np.array_equal(coords, want_coords)
    assert feats.shape == (coords.shape[0], 768) and feats.dtype == np.float32

    # features of a sample of rows vs the CPU oracle (same seeded weights, same tiles)
    rows = np.linspace(0, coords.shape[0] - 1, 12).astype(int)
    tiles = [render_region(spec, int(coords[r, 0]), int(coords[r, 1]), 256, 256, 0) for r in rows]
    sd = helpers.canonical_to_hf(random_canonical_state_dict(ARCHS["vit_b_16"], 0), 12)
    want = vit_oracle.extract_batch(sd, tiles, heads=12)
    rel = np.linalg.norm(feats[rows] - want) / np.linalg.norm(want)
    assert rel <= 1e-3, rel

    # second run: --skip-existing is the default -> nothing to do, file untouched
    before = os.path.getmtime(h5_path)
    res2 = CliRunner().invoke(cli, args, catch_exceptions=False)
    assert res2.exit_code == 0 and os.path.getmtime(h5_path) == before


def test_segment_and_get_coords_two_slides(tmp_path):
    from click.testing import CliRunner
    from atlaspatch_amd.cli import cli
    from atlaspatch_amd.utils.h5 import h5
    _make_slide(str(tmp_path), "a.synth", seed=1)
    _make_slide(str(tmp_path), "b.synth", seed=2, width=20000, height=20000)
    out = tmp_path / "out"
    res = CliRunner().invoke(cli, ["segment-and-get-coords", str(tmp_path), "-o", str(out), "--patch-size", "256",
                                   "--target-mag", "20", "--step-size", "128"], catch_exceptions=False)
    assert res.exit_code == 0 and "Completed 2 slide(s), failures: 0" in res.output
    for stem in ("a", "b"):
        with h5.File(out / "patches" / f"{stem}.h5", "r") as f:
            assert f.attrs["overlap"] == 128 and f["coords"].shape[1] == 5 and "features" not in f


def test_tile_ring_matches_direct_forward():
    from atlaspatch_amd.encoders.vit import build_hip_vit_extractor
    from atlaspatch_amd.services.tile_ring import TileRing
    dev = torch.device("cuda:0")
    ex = build_hip_vit_extractor(name="ring", arch="vit_b_16", depth=1, device=dev, dtype=torch.float16,
                                 random_init_seed=3)
    rng = np.random.default_rng(0)
    n = 70
    tiles = rng.integers(0, 256, (n, 256, 256, 3), dtype=np.uint8)
    coords = np.stack([np.arange(n), np.zeros(n), np.full(n, 256), np.full(n, 256), np.zeros(n)], 1).astype(np.int32)
    ring = TileRing(device=dev, batch=16, patch_size=256, slots=2, workers=3)
    got = ring.run(coords, lambda x, y, rw, rh, lv: tiles[x],
                   lambda t, o: ex.vit.forward_u8(t, ex.mean, ex.std, o), 768)
    ring.close()
    want = ex.extract_batch(list(tiles), batch_size=32)
    assert np.array_equal(got, want)
    ex.cleanup()


def test_tile_ring_survives_a_failing_tile_source():
    """A tile source that raises makes ``run`` raise (the slide is recorded as failed by the caller) after every decode
    task still filling pinned slots has finished; the same ring then serves the next slide correctly."""
    import os
    from atlaspatch_amd.encoders.vit import ARCHS, build_hip_vit_extractor, random_canonical_state_dict
    from atlaspatch_amd.services.tile_ring import TileRing
    arch = dict(ARCHS["vit_b_16"]); arch["depth"] = 1
    ex = build_hip_vit_extractor(name="t", arch=arch, state_dict=random_canonical_state_dict(arch, seed=1),
                                 source="canonical", device=torch.device("cuda:0"), dtype=torch.float16, expect_size=256)
    n = 300
    host = np.random.default_rng(0).integers(0, 256, (n, 256, 256, 3), dtype=np.uint8)
    coords = np.stack([np.arange(n), np.zeros(n), np.full(n, 256), np.full(n, 256), np.zeros(n)], 1).astype(np.int32)
    ring = TileRing(device=ex.device, batch=64, patch_size=256, slots=3, workers=4)

    def bad(x, y, rw, rh, lv):
        if x == 150:
            raise OSError("decode failed")
        return host[x]

    fwd = lambda t, o: ex.forward_device(t, o)
    with pytest.raises(OSError):
        ring.run(coords, bad, fwd, ex.embedding_dim)
    got = ring.run(coords, lambda x, y, rw, rh, lv: host[x], fwd, ex.embedding_dim)
    want = torch.empty((n, ex.embedding_dim), dtype=torch.float32, device=ex.device)
    ex.forward_device(torch.from_numpy(host).to(ex.device), want)
    assert np.array_equal(got, want.cpu().numpy())
    ring.close()
    ex.cleanup()


def test_cli_process_conch_and_uni_fp16(tmp_path, monkeypatch):
    """BASELINE configs 3 / 5 in miniature: `process` with uni_v1 (ViT-L/16 + LayerScale, host bicubic 224) and
    conch_v1 (448-px trunk + attentional pooler) in float16 on one small synthetic slide -> both feature sets in
    one H5, rows aligned with coords."""
    from click.testing import CliRunner
    from atlaspatch_amd.cli import cli
    from atlaspatch_amd.utils.h5 import h5

    monkeypatch.setenv("ATLASPATCH_RANDOM_INIT", "1")
    slide, raw = _make_slide(str(tmp_path), "s2.synth", width=6000, height=5000, seed=3)
    out = tmp_path / "out"
    args = ["process", slide, "-o", str(out), "--patch-size", "256", "--target-mag", "20",
            "--feature-extractors", "uni_v1,conch_v1", "--feature-precision", "float16"]
    res = CliRunner().invoke(cli, args, catch_exceptions=False)
    assert res.exit_code == 0 and "failures: 0" in res.output, res.output
    with h5.File(out / "patches" / "s2.h5", "r") as f:
        n = f["coords"].shape[0]
        uni, conch = f["features"]["uni_v1"][:], f["features"]["conch_v1"][:]
    assert n > 0 and uni.shape == (n, 1024) and conch.shape == (n, 512)
    assert np.isfinite(uni).all() and np.isfinite(conch).all()
    assert np.abs(conch).max() > 0.1 and np.unique(np.round(conch[:, 0], 3)).size > 1


def test_cli_process_with_the_widened_encoder_families(tmp_path, monkeypatch):
    """`process` with one encoder of every family added past the three starred files -- transformers DINOv2 (dinov2_small), DINOv3
    with the rotary embedding (dinov3_vits16), a CLIP tower with its projection head (plip, 512-d) and class | mean patch token
    pooling (h0_mini, 1536-d) -- all in one H5, rows aligned with coords, and the CLI's --feature-extractors help lists them."""
    from click.testing import CliRunner
    from atlaspatch_amd.cli import cli
    from atlaspatch_amd.utils.h5 import h5

    monkeypatch.setenv("ATLASPATCH_RANDOM_INIT", "1")
    slide, raw = _make_slide(str(tmp_path), "s5.synth", width=5000, height=4000, seed=5)
    out = tmp_path / "out"
    names = {"dinov2_small": 384, "dinov3_vits16": 384, "plip": 512, "h0_mini": 1536}
    args = ["process", slide, "-o", str(out), "--patch-size", "256", "--target-mag", "20",
            "--feature-extractors", ",".join(names), "--feature-precision", "float16"]
    res = CliRunner().invoke(cli, args, catch_exceptions=False)
    assert res.exit_code == 0 and "failures: 0" in res.output, res.output
    with h5.File(out / "patches" / "s5.h5", "r") as f:
        n = f["coords"].shape[0]
        feats = {k: f["features"][k][:] for k in names}
    assert n > 0
    for k, d in names.items():
        assert feats[k].shape == (n, d) and np.isfinite(feats[k]).all() and np.abs(feats[k]).max() > 0.05, k
    # (random-init LayerScale is 1e-5, so the LayerScale encoders' features barely depend on the tile; the CLIP tower has none)
    assert np.unique(np.round(feats["plip"][:, 0], 3)).size > 1
    helptext = CliRunner().invoke(cli, ["process", "--help"]).output
    assert "dinov3_vits16" in helptext and "clip_vit_l_14" in helptext


def test_cli_no_fast_mode_and_save_images(tmp_path):
    """segment-and-get-coords --no-fast-mode --save-images: rows = oracle coords minus tiles the cv2-restated
    is_black / is_white reject (reference extraction.py:105-116), one PNG per kept row with the tile's pixels."""
    from click.testing import CliRunner
    from PIL import Image
    from atlaspatch_amd.cli import cli
    from atlaspatch_amd.core.wsi.synth_pixels import SynthSpec, analytic_mask, render_region
    from atlaspatch_amd.utils.h5 import h5
    from oracle import coords_oracle, cv2_restated as cv2r

    slide, raw = _make_slide(str(tmp_path), "s3.synth", width=9000, height=7000, seed=11)
    out = tmp_path / "out"
    res = CliRunner().invoke(cli, ["segment-and-get-coords", slide, "-o", str(out), "--patch-size", "256",
                                   "--target-mag", "20", "--no-fast-mode", "--save-images"], catch_exceptions=False)
    assert res.exit_code == 0 and "failures: 0" in res.output, res.output
    spec = SynthSpec(width=raw["width"], height=raw["height"], seed=raw["seed"])
    cand, _ = coords_oracle.coords_from_mask(analytic_mask(spec), level0_wh=(spec.width, spec.height),
                                             downsamples=[1.0, 4.0, 16.0], src_mag=20, tgt_mag=20, patch_size=256,
                                             step_size=None, tissue_thresh=0.0)
    keep = []
    for row in cand:
        tile = render_region(spec, int(row[0]), int(row[1]), 256, 256, 0)
        if cv2r.is_black_patch(tile, rgb_thresh=50) or cv2r.is_white_patch(tile, sat_thresh=15):
            continue
        keep.append(row)
    want = np.asarray(keep, np.int32).reshape(-1, 5)
    with h5.File(out / "patches" / "s3.h5", "r") as f:
        got = f["coords"][:]
        assert f.attrs["num_patches"] == got.shape[0]
    assert 0 < want.shape[0] < cand.shape[0], "the slide should have both kept and rejected candidate tiles"
    assert np.array_equal(got, want)
    pngs = sorted((out / "images" / "s3").glob("*.png"))
    assert len(pngs) == want.shape[0]
    x, y = int(want[3, 0]), int(want[3, 1])
    img = np.asarray(Image.open(out / "images" / "s3" / f"s3_x{x}_y{y}.png"))
    assert np.array_equal(img, render_region(spec, x, y, 256, 256, 0))


def test_device_tile_source_equals_host_ring(tmp_path, monkeypatch):
    """Synthetic slides serve tiles straight from HBM (extract_batch_device -> ap_synth_tiles); the features must
    be bit-identical to the host path (render_region -> pinned ring -> H2D)."""
    from click.testing import CliRunner
    from atlaspatch_amd.cli import cli
    from atlaspatch_amd.utils.h5 import h5

    monkeypatch.setenv("ATLASPATCH_RANDOM_INIT", "2")
    slide, _ = _make_slide(str(tmp_path), "s4.synth", width=8000, height=6000, seed=5)
    feats = {}
    for mode in ("device", "host"):
        if mode == "host":
            monkeypatch.setenv("ATLASPATCH_HOST_TILES", "1")
        out = tmp_path / mode
        res = CliRunner().invoke(cli, ["process", slide, "-o", str(out), "--patch-size", "256", "--target-mag", "20",
                                       "--feature-extractors", "vit_b_16", "--feature-precision", "float16",
                                       "--feature-num-workers", "4"], catch_exceptions=False)
        assert res.exit_code == 0 and "failures: 0" in res.output, res.output
        with h5.File(out / "patches" / "s4.h5", "r") as f:
            feats[mode] = f["features"]["vit_b_16"][:]
    d, h = feats["device"], feats["host"]
    assert d.shape[0] > 0 and d.shape == h.shape
    if not np.array_equal(d, h):                      # say what differs: a race shows as a few rows, a layout bug as all
        bad = np.flatnonzero((d != h).any(axis=1))
        raise AssertionError(f"device-source and host-ring features differ in {bad.size} of {d.shape[0]} rows "
                             f"(first rows {bad[:8].tolist()}), max |diff| {np.nanmax(np.abs(d - h)):.3e}, "
                             f"NaNs device {int(np.isnan(d).sum())} host {int(np.isnan(h).sum())}")


def test_compressed_tile_store_through_the_native_decode_hook(tmp_path, monkeypatch):
    """A slide whose tiles sit in a (lossless) deflate tile store: the ring's workers fill the pinned slots through
    ``IWSI.read_tiles_into`` -> ``ap_host_inflate_tiles``; the features must equal those of the same slide served from
    the device tile source bit for bit."""
    import zlib
    from click.testing import CliRunner
    from atlaspatch_amd.cli import cli
    from atlaspatch_amd.core.wsi.synth_pixels import SynthSpec, render_region
    from atlaspatch_amd.utils.h5 import h5

    monkeypatch.setenv("ATLASPATCH_RANDOM_INIT", "2")
    spec_kw = dict(width=7000, height=5000, seed=8)
    args = ["--patch-size", "256", "--target-mag", "20", "--feature-extractors", "vit_b_16", "--feature-precision", "float16",
            "--feature-num-workers", "4"]
    plain = tmp_path / "plain"; plain.mkdir()
    (plain / "s5.synth").write_text(json.dumps({**spec_kw, "mag": 20, "mpp": 0.5, "downsamples": [1, 4, 16]}))
    res = CliRunner().invoke(cli, ["process", str(plain / "s5.synth"), "-o", str(tmp_path / "o1"), *args], catch_exceptions=False)
    assert res.exit_code == 0 and "failures: 0" in res.output, res.output
    with h5.File(tmp_path / "o1" / "patches" / "s5.h5", "r") as f:
        coords, want = f["coords"][:], f["features"]["vit_b_16"][:]
    stored = tmp_path / "stored"; (stored / "tiles").mkdir(parents=True)
    spec = SynthSpec(**spec_kw)
    for x, y in coords[:, :2].tolist():
        (stored / "tiles" / f"{x}_{y}_256.z").write_bytes(zlib.compress(render_region(spec, x, y, 256, 256, 0).tobytes(), 1))
    (stored / "s5.synth").write_text(json.dumps({**spec_kw, "mag": 20, "mpp": 0.5, "downsamples": [1, 4, 16], "jpeg_tiles": "tiles"}))
    res = CliRunner().invoke(cli, ["process", str(stored / "s5.synth"), "-o", str(tmp_path / "o2"), *args], catch_exceptions=False)
    assert res.exit_code == 0 and "failures: 0" in res.output, res.output
    with h5.File(tmp_path / "o2" / "patches" / "s5.h5", "r") as f:
        got = f["features"]["vit_b_16"][:]
    assert got.shape[0] > 0 and np.array_equal(got, want)


def test_segment_and_get_coords_with_sam2_on_an_image_slide(tmp_path, monkeypatch):
    """Real (non-synthetic) slide path: a PNG through the Pillow backend, SAM2 Hiera-T segmenter on the HIP
    operator set (seeded random weights: the mask is arbitrary, the plumbing is what is checked), device coords,
    H5 output; the coords must equal the oracle's for the mask the segmenter produced."""
    from click.testing import CliRunner
    from PIL import Image
    from atlaspatch_amd.cli import cli
    from atlaspatch_amd.core.config import SegmentationConfig
    from atlaspatch_amd.core.wsi.image_wsi import ImageWSI
    from atlaspatch_amd.services.segmentation import SAM2SegmentationService
    from atlaspatch_amd.utils.h5 import h5
    from oracle import coords_oracle

    monkeypatch.setenv("ATLASPATCH_RANDOM_INIT", "4")
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (3000, 4000, 3), dtype=np.uint8)
    path = tmp_path / "slide.png"
    Image.fromarray(img).save(path)
    out = tmp_path / "out"
    res = CliRunner().invoke(cli, ["segment-and-get-coords", str(path), "-o", str(out), "--patch-size", "256",
                                   "--target-mag", "20", "--mpp-csv", str(_mpp_csv(tmp_path, "slide.png", 0.5))],
                             catch_exceptions=False)
    assert res.exit_code == 0 and "failures: 0" in res.output, res.output
    with h5.File(out / "patches" / "slide.h5", "r") as f:
        got = f["coords"][:]
    seg = SAM2SegmentationService(SegmentationConfig(checkpoint_path=None, config_path=path, device="cuda"))
    wsi = ImageWSI(path=str(path), mpp=0.5)
    wsi._ensure_loaded()
    mask = seg.segment_thumbnail(wsi).data
    seg.close()
    assert mask.shape == (188, 250) and set(np.unique(mask)) <= {0.0, 1.0}      # 4000 x 3000 at 20x -> 1.25x
    want, _ = coords_oracle.coords_from_mask(mask, level0_wh=(4000, 3000), downsamples=[1.0], src_mag=wsi.mag,
                                             tgt_mag=20, patch_size=256, step_size=None, tissue_thresh=0.0)
    assert np.array_equal(got, want)


def test_process_on_a_slide_behind_a_stub_openslide(tmp_path, monkeypatch):
    """The OpenSlide backend end to end (BASELINE config 1's code path; openslide-python itself is absent): a stub
    ``openslide`` module with the library's interface (dimensions, level_count, level_downsamples, level_dimensions,
    properties, read_region -> RGBA, get_thumbnail, close) serves a synthetic slide with CMU-1-like level
    downsamples behind a ``.svs`` path.  ``process`` runs SAM2 segmentation (seeded random weights: the mask is
    arbitrary), device coords, the tile ring through ``OpenSlideWSI.extract`` (read_region(...).convert("RGB")) and
    the encoder.  Coords must equal the oracle's for the mask the segmenter produced; features must equal the
    encoder's on the same tiles rendered directly."""
    import types
    from click.testing import CliRunner
    from PIL import Image
    from atlaspatch_amd.cli import cli
    from atlaspatch_amd.core.config import SegmentationConfig
    from atlaspatch_amd.core.wsi import openslide_wsi
    from atlaspatch_amd.core.wsi.synth_pixels import SynthSpec, render_region
    from atlaspatch_amd.encoders import build_default_registry
    from atlaspatch_amd.services.segmentation import SAM2SegmentationService
    from atlaspatch_amd.utils.h5 import h5
    from oracle import coords_oracle

    spec = SynthSpec(width=11000, height=8200, seed=77)
    ds = (1.0, 4.0001, 16.00097)
    calls = {"read_region": 0, "closed": 0}

    class FakeOpenSlide:
        def __init__(self, path):
            assert str(path).endswith(".svs")
            self.dimensions = (spec.width, spec.height)
            self.level_count = len(ds)
            self.level_downsamples = ds
            self.level_dimensions = tuple((int(spec.width / d), int(spec.height / d)) for d in ds)
            self.properties = {"openslide.mpp-x": "0.4990", "openslide.mpp-y": "0.4990", "aperio.AppMag": "20",
                               "openslide.objective-power": "20", "openslide.vendor": "aperio"}

        def read_region(self, location, level, size):
            calls["read_region"] += 1
            rgb = render_region(spec, int(location[0]), int(location[1]), int(size[0]), int(size[1]), int(level))
            rgba = np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 255, np.uint8)], axis=2)
            return Image.fromarray(rgba, "RGBA")

        def get_thumbnail(self, size):
            lv = self.level_count - 1
            w, h = self.level_dimensions[lv]
            im = self.read_region((0, 0), lv, (w, h))
            im.thumbnail(size)
            return im

        def close(self):
            calls["closed"] += 1

    monkeypatch.setattr(openslide_wsi, "openslide", types.SimpleNamespace(OpenSlide=FakeOpenSlide))
    monkeypatch.setenv("ATLASPATCH_RANDOM_INIT", "4")
    path = tmp_path / "CMU-like.svs"
    path.write_bytes(b"stub")
    out = tmp_path / "out"
    res = CliRunner().invoke(cli, ["process", str(path), "-o", str(out), "--patch-size", "256", "--target-mag", "20",
                                   "--feature-extractors", "vit_b_16", "--feature-precision", "float32",
                                   "--feature-num-workers", "4"], catch_exceptions=False)
    assert res.exit_code == 0 and "failures: 0" in res.output, res.output
    with h5.File(out / "patches" / "CMU-like.h5", "r") as f:
        coords = f["coords"][:]
        feats = f["features"]["vit_b_16"][:]
        assert float(f.attrs["level0_magnification"]) == 20 and int(f.attrs["patch_size_level0"]) == 256
    assert calls["read_region"] >= coords.shape[0] > 0 and calls["closed"] >= 1
    # coords: the oracle on the mask the segmenter produces for this slide
    seg = SAM2SegmentationService(SegmentationConfig(checkpoint_path=None, config_path=path, device="cuda"))
    wsi = openslide_wsi.OpenSlideWSI(path=str(path))
    wsi._ensure_loaded()
    assert wsi.mpp == 0.499 and wsi.mag == 20 and wsi.ds == list(ds)
    mask = seg.segment_thumbnail(wsi).data
    seg.close()
    want, _ = coords_oracle.coords_from_mask(mask, level0_wh=(spec.width, spec.height), downsamples=list(ds), src_mag=20,
                                             tgt_mag=20, patch_size=256, step_size=None, tissue_thresh=0.0)
    assert np.array_equal(coords, want)
    assert (coords[:, 2] == 256).all() and (coords[:, 4] == 0).all()
    # features: the same tiles rendered directly, through the encoder's boundary call
    rows = np.linspace(0, coords.shape[0] - 1, 12).astype(int)
    tiles = [render_region(spec, int(coords[r, 0]), int(coords[r, 1]), 256, 256, 0) for r in rows]
    ex = build_default_registry(device="cuda", dtype=torch.float32).create("vit_b_16")      # same seed as the CLI run
    direct = ex.extract_batch(tiles)
    ex.cleanup()
    assert np.allclose(feats[rows], direct, rtol=1e-5, atol=1e-6)


def _mpp_csv(folder, name, mpp):
    p = folder / "mpp.csv"
    p.write_text(f"wsi,mpp\n{name},{mpp}\n")
    return p


@pytest.mark.parametrize("host_tiles", [False, True])
def test_cli_process_mag40_slide_resizes_tiles_on_device(tmp_path, monkeypatch, host_tiles):
    """A 40x slide at --target-mag 20 with levels 1/4/16 (the common scanner case): every tile is READ 512 x 512 at
    level 0 and the reference shrinks it with cv2.resize(patch, (256, 256)) (feature_embedding.py:94-95).  Here the
    tiles cross the ring (or come from the device tile source) at their read size and ap_cv2_resize_u8 shrinks them on
    the device; features equal the oracle chain render -> oracle cv2.resize -> fp32 ViT."""
    from click.testing import CliRunner
    from atlaspatch_amd.cli import cli
    from atlaspatch_amd.core.wsi.synth_pixels import SynthSpec, analytic_mask, render_region
    from atlaspatch_amd.encoders.vit import ARCHS, random_canonical_state_dict
    from atlaspatch_amd.utils.h5 import h5
    from oracle import coords_oracle, cv2_resize, vit_oracle

    monkeypatch.setenv("ATLASPATCH_RANDOM_INIT", "0")
    if host_tiles:
        monkeypatch.setenv("ATLASPATCH_HOST_TILES", "1")
    slide, raw = _make_slide(str(tmp_path), "m40.synth", mag=40, mpp=0.25, width=14000, height=10000)
    out = tmp_path / "out"
    res = CliRunner().invoke(cli, ["process", slide, "-o", str(out), "--patch-size", "256", "--target-mag", "20",
                                   "--feature-extractors", "vit_b_16", "--feature-precision", "float32",
                                   "--feature-num-workers", "4"], catch_exceptions=False)
    assert res.exit_code == 0 and "failures: 0" in res.output, res.output
    spec = SynthSpec(width=raw["width"], height=raw["height"], seed=raw["seed"], mag=40, mpp=0.25)
    want_coords, _ = coords_oracle.coords_from_mask(
        analytic_mask(spec), level0_wh=(spec.width, spec.height), downsamples=[1.0, 4.0, 16.0], src_mag=40,
        tgt_mag=20, patch_size=256, step_size=None, tissue_thresh=0.0)
    with h5.File(out / "patches" / "m40.h5", "r") as f:
        coords = f["coords"][:]
        feats = f["features"]["vit_b_16"][:]
        assert f.attrs["patch_size_level0"] == 512
    assert np.array_equal(coords, want_coords) and coords.shape[0] > 20
    assert (coords[:, 2] == 512).all() and (coords[:, 3] == 512).all() and (coords[:, 4] == 0).all()
    rows = np.linspace(0, coords.shape[0] - 1, 10).astype(int)
    tiles = [cv2_resize.resize(render_region(spec, int(coords[r, 0]), int(coords[r, 1]), 512, 512, 0), (256, 256))
             for r in rows]
    sd = helpers.canonical_to_hf(random_canonical_state_dict(ARCHS["vit_b_16"], 0), 12)
    want = vit_oracle.extract_batch(sd, tiles, heads=12)
    rel = np.linalg.norm(feats[rows] - want) / np.linalg.norm(want)
    assert rel <= 1e-3, rel


def test_no_fast_mode_and_save_images_on_a_mag40_slide(tmp_path):
    """--no-fast-mode / --save-images on a slide whose tiles need cv2.resize (extraction.py:105-128): saved PNGs are the
    oracle-resized tiles, and the content filters see the resized pixels."""
    from click.testing import CliRunner
    from PIL import Image
    from atlaspatch_amd.cli import cli
    from atlaspatch_amd.core.wsi.synth_pixels import SynthSpec, render_region
    from atlaspatch_amd.utils.h5 import h5
    from oracle import cv2_resize
    slide, raw = _make_slide(str(tmp_path), "f40.synth", mag=40, mpp=0.25, width=9000, height=7000)
    out = tmp_path / "out"
    res = CliRunner().invoke(cli, ["segment-and-get-coords", slide, "-o", str(out), "--patch-size", "256", "--target-mag",
                                   "20", "--no-fast-mode", "--save-images"], catch_exceptions=False)
    assert res.exit_code == 0 and "failures: 0" in res.output, res.output
    with h5.File(out / "patches" / "f40.h5", "r") as f:
        coords = f["coords"][:]
    assert coords.shape[0] > 5
    spec = SynthSpec(width=raw["width"], height=raw["height"], seed=raw["seed"], mag=40, mpp=0.25)
    for r in (0, coords.shape[0] // 2, coords.shape[0] - 1):
        x, y = int(coords[r, 0]), int(coords[r, 1])
        png = np.asarray(Image.open(out / "images" / "f40" / f"f40_x{x}_y{y}.png"))
        assert np.array_equal(png, cv2_resize.resize(render_region(spec, x, y, 512, 512, 0), (256, 256)))


def test_segment_and_get_coords_on_a_cmu1_shaped_slide(tmp_path):
    """BASELINE config 1's stand-in (CMU-1.svs, OpenSlide and the SAM2 checkpoint are absent): a synthetic slide with
    CMU-1's geometry -- 46000 x 32914 at 20x, mpp 0.499, level downsamples (1, 4.0001, 16.00097) -- through
    `segment-and-get-coords`.  The 1.25x thumbnail level is picked by the |d - t| < 0.01 exact-match rule
    (iwsi.py:325-358), the mask is 733 x 1024, sx = 46000 / 1024; coords equal the CPU oracle row for row and the H5
    carries the reference's attrs."""
    from click.testing import CliRunner
    from atlaspatch_amd.cli import cli
    from atlaspatch_amd.core.wsi.synth_pixels import SynthSpec, analytic_mask
    from atlaspatch_amd.utils.h5 import h5
    from oracle import coords_oracle
    ds = [1.0, 4.000121536217793, 16.000972053462940]
    slide, raw = _make_slide(str(tmp_path), "cmu1.synth", width=46000, height=32914, mag=20, mpp=0.499, seed=11,
                             downsamples=ds)
    out = tmp_path / "out"
    res = CliRunner().invoke(cli, ["segment-and-get-coords", slide, "-o", str(out), "--patch-size", "256",
                                   "--target-mag", "20"], catch_exceptions=False)
    assert res.exit_code == 0 and "failures: 0" in res.output, res.output
    spec = SynthSpec(width=46000, height=32914, seed=11, mag=20, mpp=0.499, downsamples=tuple(ds))
    mask = analytic_mask(spec)
    assert mask.shape == (733, 1024)
    want, _ = coords_oracle.coords_from_mask(mask, level0_wh=(46000, 32914), downsamples=ds, src_mag=20, tgt_mag=20,
                                             patch_size=256, step_size=None, tissue_thresh=0.0)
    with h5.File(out / "patches" / "cmu1.h5", "r") as f:
        coords = f["coords"][:]
        assert f.attrs["level0_width"] == 46000 and f.attrs["level0_height"] == 32914
        assert f.attrs["patch_size_level0"] == 256 and f.attrs["level0_magnification"] == 20
        assert f["passports"].shape[0] == coords.shape[0]
    assert coords.shape[0] > 1000 and np.array_equal(coords, want)
    assert (coords[:, 2] == 256).all() and (coords[:, 4] == 0).all()


def test_process_40k_slide_vit_b16(tmp_path, monkeypatch):
    """BASELINE config 2 at full size: `process` on one synthetic 40 000 x 40 000 slide, ViT-B/16 (torchvision layout),
    float16 (the CLI's default feature precision, cli.py:175-181), 1 x MI355X.  Coordinates equal the oracle's row for row;
    every feature row is finite; sampled rows are bit-equal to a direct `extract_batch` of the same tiles (the tile source,
    the device batches of 2048 and the H5 writer add no arithmetic) and within 1e-3 norm-wise of the fp32 CPU oracle.
    The twin of the config-3 test below."""
    from click.testing import CliRunner
    from atlaspatch_amd.cli import cli
    from atlaspatch_amd.core.wsi.synth_pixels import SynthSpec, analytic_mask, render_region
    from atlaspatch_amd.encoders import build_default_registry
    from atlaspatch_amd.encoders.vit import ARCHS, random_canonical_state_dict
    from atlaspatch_amd.utils.h5 import h5
    from oracle import coords_oracle, vit_oracle
    monkeypatch.setenv("ATLASPATCH_RANDOM_INIT", "0")
    slide, raw = _make_slide(str(tmp_path), "c2.synth", width=40000, height=40000, seed=4040)
    out = tmp_path / "out"
    res = CliRunner().invoke(cli, ["process", slide, "-o", str(out), "--patch-size", "256", "--target-mag", "20",
                                   "--feature-extractors", "vit_b_16", "--feature-precision", "float16"], catch_exceptions=False)
    assert res.exit_code == 0 and "failures: 0" in res.output, res.output
    spec = SynthSpec(width=40000, height=40000, seed=4040)
    want_coords, _ = coords_oracle.coords_from_mask(
        analytic_mask(spec), level0_wh=(40000, 40000), downsamples=[1.0, 4.0, 16.0], src_mag=20, tgt_mag=20,
        patch_size=256, step_size=None, tissue_thresh=0.0)
    with h5.File(out / "patches" / "c2.h5", "r") as f:
        coords = f["coords"][:]
        feats = f["features"]["vit_b_16"][:]
        assert f.attrs["level0_width"] == 40000 and f.attrs["patch_size_level0"] == 256
    n = coords.shape[0]
    assert n == want_coords.shape[0] and n > 5000 and np.array_equal(coords, want_coords)
    assert feats.shape == (n, 768) and feats.dtype == np.float32 and np.isfinite(feats).all()
    rows = np.unique(np.array([0, 1, 2047, 2048, n // 2, n - 2, n - 1]).clip(0, n - 1))
    tiles = [render_region(spec, int(coords[r, 0]), int(coords[r, 1]), 256, 256, 0) for r in rows]
    ex = build_default_registry(device="cuda", dtype=torch.float16).create("vit_b_16")
    direct = ex.extract_batch(tiles, batch_size=32)
    ex.cleanup()
    assert np.array_equal(feats[rows], direct)
    sd = helpers.canonical_to_hf(random_canonical_state_dict(ARCHS["vit_b_16"], 0), 12)
    want = vit_oracle.extract_batch(sd, tiles, heads=12)
    rel = np.linalg.norm(feats[rows] - want) / np.linalg.norm(want)
    print(f"PARITY config 2 (40 000^2, vit_b_16 float16): {n} rows, sampled rows vs fp32 oracle {rel:.3e}")
    assert rel <= 1e-3, rel


def test_process_100k_slide_through_the_host_ring_uni_v1(tmp_path, monkeypatch):
    """BASELINE config 3 at full size: `process` on the 100 000 x 100 000 synthetic slide with uni_v1 (ViT-L/16 +
    LayerScale, float16), tiles produced on HOST threads (the native renderer standing in for a slide decoder), crossing
    the pinned ring -> H2D -> device resize (256 -> 224 bicubic) -> encoder.  58 938 rows; coords equal the oracle;
    sampled feature rows equal a direct forward of the same tiles (bit for bit: the ring adds no arithmetic) and the
    fp32 oracle within the float16 tolerance."""
    from click.testing import CliRunner
    from PIL import Image
    from atlaspatch_amd.cli import cli
    from atlaspatch_amd.core.wsi.synth_pixels import SynthSpec, analytic_mask, render_region
    from atlaspatch_amd.encoders import build_default_registry
    from atlaspatch_amd.encoders.vit import ARCHS, random_canonical_state_dict
    from atlaspatch_amd.utils.h5 import h5
    from oracle import coords_oracle, vit_oracle
    monkeypatch.setenv("ATLASPATCH_RANDOM_INIT", "0")
    monkeypatch.setenv("ATLASPATCH_HOST_TILES", "1")
    slide, raw = _make_slide(str(tmp_path), "big.synth", width=100000, height=100000, seed=1234)
    out = tmp_path / "out"
    res = CliRunner().invoke(cli, ["process", slide, "-o", str(out), "--patch-size", "256", "--target-mag", "20",
                                   "--feature-extractors", "uni_v1", "--feature-precision", "float16",
                                   "--feature-num-workers", "32"], catch_exceptions=False)
    assert res.exit_code == 0 and "failures: 0" in res.output, res.output
    spec = SynthSpec(width=100000, height=100000, seed=1234)
    want_coords, _ = coords_oracle.coords_from_mask(
        analytic_mask(spec), level0_wh=(100000, 100000), downsamples=[1.0, 4.0, 16.0], src_mag=20, tgt_mag=20,
        patch_size=256, step_size=None, tissue_thresh=0.0)
    with h5.File(out / "patches" / "big.h5", "r") as f:
        coords = f["coords"][:]
        feats = f["features"]["uni_v1"][:]
    assert coords.shape[0] == 58938 and np.array_equal(coords, want_coords)
    assert feats.shape == (58938, 1024) and np.isfinite(feats).all()
    rows = np.array([0, 1, 2047, 2048, 30000, 58937])
    tiles = [render_region(spec, int(coords[r, 0]), int(coords[r, 1]), 256, 256, 0) for r in rows]
    ex = build_default_registry(device="cuda", dtype=torch.float16).create("uni_v1")
    direct = ex.extract_batch(tiles, batch_size=32)
    ex.cleanup()
    assert np.array_equal(feats[rows], direct)
    sd = random_canonical_state_dict(ARCHS["uni_v1"], 0)
    pre = np.stack([np.asarray(Image.fromarray(t).resize((224, 224), Image.Resampling.BICUBIC)) for t in tiles], 0)
    want = vit_oracle.vit_tokens_canonical(sd, vit_oracle.preprocess_center_crop(pre, crop=224), heads=16, depth=24)[:, 0].numpy()
    rel = np.linalg.norm(feats[rows] - want) / np.linalg.norm(want)
    assert rel <= 1.5e-3, rel


def test_process_jpeg_tile_store_through_the_native_decoder(tmp_path, monkeypatch):
    """A slide whose tiles are JPEG files (the stand-in for a real slide's compressed tiles): the ring's pinned host
    threads decode chunks through ap_host_decode_jpeg_tiles (libjpeg-turbo, outside the interpreter lock).  Features equal
    a direct forward of the Pillow-decoded tiles bit for bit (same pixels in, same kernels)."""
    from click.testing import CliRunner
    from PIL import Image
    from atlaspatch_amd.cli import cli
    from atlaspatch_amd.core.wsi.synth_pixels import SynthSpec, analytic_mask, render_region
    from atlaspatch_amd.encoders import build_default_registry
    from atlaspatch_amd.services.extraction import coords_from_mask
    from atlaspatch_amd.utils.h5 import h5
    monkeypatch.setenv("ATLASPATCH_RANDOM_INIT", "0")
    store = tmp_path / "tiles"
    store.mkdir()
    slide, raw = _make_slide(str(tmp_path), "j.synth", width=14000, height=10000, jpeg_tiles="tiles")
    spec = SynthSpec(width=14000, height=10000, seed=raw["seed"])
    coords, _ = coords_from_mask(analytic_mask(spec), level0_wh=(14000, 10000), downsamples=[1.0, 4.0, 16.0], src_mag=20,
                                 tgt_mag=20, patch_size=256, step_size=None, tissue_thresh=0.0)
    for x, y in coords[:, :2]:
        Image.fromarray(render_region(spec, int(x), int(y), 256, 256, 0)).save(str(store / f"{x}_{y}_256.jpg"), quality=80)
    out = tmp_path / "out"
    res = CliRunner().invoke(cli, ["process", slide, "-o", str(out), "--patch-size", "256", "--target-mag", "20",
                                   "--feature-extractors", "vit_b_16", "--feature-precision", "float16",
                                   "--feature-num-workers", "8"], catch_exceptions=False)
    assert res.exit_code == 0 and "failures: 0" in res.output, res.output
    with h5.File(out / "patches" / "j.h5", "r") as f:
        got_coords, feats = f["coords"][:], f["features"]["vit_b_16"][:]
    assert np.array_equal(got_coords, coords) and coords.shape[0] > 100
    rows = np.linspace(0, coords.shape[0] - 1, 16).astype(int)
    tiles = [np.asarray(Image.open(str(store / f"{coords[r, 0]}_{coords[r, 1]}_256.jpg")).convert("RGB")) for r in rows]
    ex = build_default_registry(device="cuda", dtype=torch.float16).create("vit_b_16")
    want = ex.extract_batch(tiles, batch_size=32)
    ex.cleanup()
    assert np.array_equal(feats[rows], want)


def test_visualize_flags_write_the_three_overlays(tmp_path):
    """--visualize-grids / --visualize-mask / --visualize-contours (services/visualization.py:36-102): PNGs under
    <out>/visualization with the reference's file names; the contour overlay carries red tissue outlines."""
    from click.testing import CliRunner
    from PIL import Image
    from atlaspatch_amd.cli import cli
    slide, _ = _make_slide(str(tmp_path), "viz.synth", width=20000, height=14000)
    out = tmp_path / "out"
    res = CliRunner().invoke(cli, ["segment-and-get-coords", slide, "-o", str(out), "--patch-size", "256", "--target-mag", "20",
                                   "--visualize-grids", "--visualize-mask", "--visualize-contours"], catch_exceptions=False)
    assert res.exit_code == 0 and "failures: 0" in res.output, res.output
    vis = out / "visualization"
    for name in ("viz.png", "viz_mask.png", "viz_mask_bw.png", "viz_contours.png"):
        assert (vis / name).exists(), name
    cont = np.asarray(Image.open(vis / "viz_contours.png"))
    red = (cont[..., 0] == 255) & (cont[..., 1] == 0) & (cont[..., 2] == 0)
    assert red.sum() > 200
