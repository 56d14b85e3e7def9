"""CPU tests of the host side: config / registry / plugin API / H5 output / C-ABI symbols."""
import ctypes
import json
import os
import re
import subprocess
import textwrap

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ----------------------------------------------------------------------------- config (G7)
def _cases(golden_dir):
    return json.load(open(os.path.join(golden_dir, "config_cases.json")))


def test_device_strings(golden_dir):
    from atlaspatch_amd.core.config import normalise_device
    for raw, want in _cases(golden_dir)["device"]:
        if isinstance(want, dict):
            with pytest.raises(ValueError) as err:
                normalise_device(raw)
            assert str(err.value) == want["error"]
        else:
            assert normalise_device(raw) == want


def test_extraction_and_feature_config(golden_dir):
    from atlaspatch_amd.core.config import ExtractionConfig, FeatureExtractionConfig
    for kw, want in _cases(golden_dir)["extraction"]:
        if "error" in want:
            with pytest.raises(ValueError) as err:
                ExtractionConfig(**kw).validated()
            assert str(err.value) == want["error"]
        else:
            cfg = ExtractionConfig(**kw).validated()
            assert {"step_size": cfg.step_size, "max_open_slides": cfg.max_open_slides} == want
    for kw, want in _cases(golden_dir)["features"]:
        if "error" in want:
            with pytest.raises(ValueError) as err:
                FeatureExtractionConfig(**kw).validated()
            assert str(err.value) == want["error"]
        else:
            cfg = FeatureExtractionConfig(**kw).validated()
            assert {"precision": cfg.precision, "device": cfg.device} == want


def test_resolve_feature_dtype_and_registry(golden_dir):
    import click
    from atlaspatch_amd.encoders.registry import PatchFeatureExtractorRegistry
    from atlaspatch_amd.services.feature_embedding import resolve_feature_dtype
    from atlaspatch_amd.utils.features import parse_feature_list
    cases = _cases(golden_dir)
    for dev, prec, want in cases["dtype"]:
        assert str(resolve_feature_dtype(torch.device(dev), prec)) == want
    reg = PatchFeatureExtractorRegistry()
    reg.register("Foo", lambda: "foo-built")
    reg.register("bar", lambda: "bar-built")
    ev = cases["registry"]
    assert reg.available() == ev["available"] and reg.create("FOO") == ev["create_FOO"]
    with pytest.raises(ValueError) as e1:
        reg.register("foo", lambda: 1)
    assert str(e1.value) == ev["dup"]
    with pytest.raises(KeyError) as e2:
        reg.create("nope")
    assert str(e2.value) == ev["unknown"]
    for raw, want in cases["parse_feature_list"]:
        if isinstance(want, dict):
            with pytest.raises(click.BadParameter) as e3:
                parse_feature_list(raw, choices=["vit_b_16", "uni_v1"])
            assert e3.value.message == want["error"]
        else:
            assert parse_feature_list(raw, choices=["vit_b_16", "uni_v1"]) == want


def test_product_geometry_matches_reference(golden_dir):
    from atlaspatch_amd.services.extraction import _StaticLevels
    from atlaspatch_amd.services.geometry import prepare_geometry
    table = json.load(open(os.path.join(golden_dir, "geometry.json")))
    for row in table["geometry"]:
        wsi = _StaticLevels(row["ds"], row["mag"])
        try:
            g = prepare_geometry(wsi, patch_size=row["ps"], step_size=row["step"], target_magnification=row["tgt"])
            got = [g.level, g.read_wh[0], g.read_wh[1], g.patch_size_src, g.step_src, g.patch_size_level0]
        except ValueError as exc:
            got = {"error": str(exc)}
        assert got == row["out"], row
    for row in table["levels"]:
        wsi = _StaticLevels(row["ds"], 20)
        if "error" in row:
            with pytest.raises(ValueError):
                wsi.optimal_level(row["target"])
        else:
            assert wsi.optimal_level(row["target"]) == (row["level"], row["extra"])


# ----------------------------------------------------------------------------- plugin API on CPU
def test_plugin_hook_and_generic_extractor_match_reference_golden(tmp_path, golden_dir):
    """A plugin file written for the reference's API runs unchanged (generic torch path, CPU) and
    reproduces the reference's own extract_batch outputs (golden G1)."""
    from atlaspatch_amd.encoders import PatchFeatureExtractorRegistry, register_feature_extractors_from_module
    from tests import helpers
    plugin = tmp_path / "my_plugin.py"
    plugin.write_text(textwrap.dedent('''
        import numpy as np, torch
        from atlaspatch_amd.encoders import CustomEncoderComponents, register_custom_encoder
        from oracle import vit_oracle
        MEAN = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
        STD = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
        def preprocess(pil):
            arr = np.asarray(pil, dtype=np.uint8)[16:240, 16:240, :]
            x = torch.from_numpy(arr.copy()).permute(2, 0, 1).to(torch.float32).div(255)
            return x.sub(MEAN).div(STD)
        def register_feature_extractors(registry, device, dtype, num_workers):
            def loader(dev, dt):
                m = vit_oracle.make_hf_vit(layers=2)
                return CustomEncoderComponents(model=m, preprocess=preprocess,
                                               forward_fn=lambda x: m(pixel_values=x).last_hidden_state[:, 0])
            register_custom_encoder(registry=registry, name="HFvit_L2", embedding_dim=768, loader=loader,
                                    device=device, dtype=dtype, num_workers=num_workers)
    '''))
    reg = PatchFeatureExtractorRegistry()
    register_feature_extractors_from_module(plugin, reg, device=torch.device("cpu"), dtype=torch.float32, num_workers=0)
    assert reg.available() == ["hfvit_l2"]
    ex = reg.create("hfvit_L2")
    g = np.load(os.path.join(golden_dir, "extract_batch.npz"))
    ns = (0, 1, 5, 32, 33)
    patches = helpers.golden_patches(ns)
    for n in ns:
        out = ex.extract_batch(patches[n], batch_size=32)
        assert out.dtype == np.float32 and out.shape == (n, 768) and out.flags.c_contiguous
        if n:
            assert np.linalg.norm(out - g[f"L2_n{n}_out"]) / np.linalg.norm(g[f"L2_n{n}_out"]) < 2e-5
    ex.cleanup()
    bad = tmp_path / "no_hook.py"
    bad.write_text("x = 1\n")
    with pytest.raises(AttributeError):
        register_feature_extractors_from_module(bad, reg, device=torch.device("cpu"), dtype=torch.float32)


def test_state_dict_adapters_agree():
    """torchvision / timm / HF key layouts map to the same canonical parameters."""
    from atlaspatch_amd.encoders.vit import ARCHS, canonical_state_dict, random_canonical_state_dict
    arch = dict(ARCHS["uni_v1"]); arch["depth"] = 2
    canon = random_canonical_state_dict(arch, seed=3)
    d = arch["dim"]
    tv = {"conv_proj.weight": canon["patch_embed.weight"], "conv_proj.bias": canon["patch_embed.bias"],
          "class_token": canon["cls_token"].view(1, 1, d), "encoder.pos_embedding": canon["pos_embed"][None],
          "encoder.ln.weight": canon["norm.weight"], "encoder.ln.bias": canon["norm.bias"]}
    timm = {"patch_embed.proj.weight": canon["patch_embed.weight"], "patch_embed.proj.bias": canon["patch_embed.bias"],
            "cls_token": canon["cls_token"].view(1, 1, d), "pos_embed": canon["pos_embed"][None],
            "norm.weight": canon["norm.weight"], "norm.bias": canon["norm.bias"]}
    for i in range(2):
        b = f"blocks.{i}."
        p = f"encoder.layers.encoder_layer_{i}."
        tv.update({p + "ln_1.weight": canon[b + "ln1.weight"], p + "ln_1.bias": canon[b + "ln1.bias"],
                   p + "self_attention.in_proj_weight": canon[b + "qkv.weight"],
                   p + "self_attention.in_proj_bias": canon[b + "qkv.bias"],
                   p + "self_attention.out_proj.weight": canon[b + "proj.weight"],
                   p + "self_attention.out_proj.bias": canon[b + "proj.bias"],
                   p + "ln_2.weight": canon[b + "ln2.weight"], p + "ln_2.bias": canon[b + "ln2.bias"],
                   p + "mlp.0.weight": canon[b + "fc1.weight"], p + "mlp.0.bias": canon[b + "fc1.bias"],
                   p + "mlp.3.weight": canon[b + "fc2.weight"], p + "mlp.3.bias": canon[b + "fc2.bias"]})
        timm.update({b + "norm1.weight": canon[b + "ln1.weight"], b + "norm1.bias": canon[b + "ln1.bias"],
                     b + "attn.qkv.weight": canon[b + "qkv.weight"], b + "attn.qkv.bias": canon[b + "qkv.bias"],
                     b + "attn.proj.weight": canon[b + "proj.weight"], b + "attn.proj.bias": canon[b + "proj.bias"],
                     b + "ls1.gamma": canon[b + "ls1"], b + "ls2.gamma": canon[b + "ls2"],
                     b + "norm2.weight": canon[b + "ln2.weight"], b + "norm2.bias": canon[b + "ln2.bias"],
                     b + "mlp.fc1.weight": canon[b + "fc1.weight"], b + "mlp.fc1.bias": canon[b + "fc1.bias"],
                     b + "mlp.fc2.weight": canon[b + "fc2.weight"], b + "mlp.fc2.bias": canon[b + "fc2.bias"]})
    a = canonical_state_dict(timm, depth=2, layer_scale=True)
    for k, v in canon.items():
        assert torch.equal(a[k], v), k
    t = canonical_state_dict(tv, depth=2, layer_scale=False)
    for k, v in canon.items():
        if ".ls" not in k:
            assert torch.equal(t[k], v), k


# ----------------------------------------------------------------------------- H5 output (G5)
def test_h5_layout_matches_reference(tmp_path, golden_dir):
    from atlaspatch_amd.services.storage import H5PatchWriter, read_coords
    from atlaspatch_amd.utils.features import get_existing_features, missing_features
    from atlaspatch_amd.utils.h5 import h5
    want = json.load(open(os.path.join(golden_dir, "features_h5.json")))
    arrays = np.load(os.path.join(golden_dir, "features_h5.npz"))
    path = tmp_path / "patches" / "tiny.h5"
    path.parent.mkdir()
    writer = H5PatchWriter(chunk_rows=8192, patch_size=256, patch_size_level0=256, level0_mag=20, target_mag=20,
                           level0_wh=(8192, 8192), overlap=0, slide_stem="tiny", wsi_path="x",
                           extra_file_attrs={"filename": "tiny.synth", "mpp": 0.5, "magnification": 20,
                                             "vendor": "synthetic"})
    n = writer.write_coords_array(path, arrays["coords"])
    assert n == want["num_patches"] and not list(path.parent.glob(".tiny.h5.tmp.*"))
    assert missing_features(path, ["tiny12"], expected_total=n) == ["tiny12"]
    # iterator form with the reference's signature, batch 7 like the golden run
    feats = arrays["feats"]
    calls = []

    def fn(buf):
        calls.append(len(buf))
        start = sum(calls[:-1])
        return feats[start:start + len(buf)]

    entries = ((int(r[0]), int(r[1]), 256, 256, 0, np.zeros((2, 2, 3), np.uint8)) for r in arrays["coords"])
    writer.append_features(output_path=path, entries=entries, feature_name="tiny12", feature_fn=fn,
                           feature_attrs={"name": "tiny12", "embedding_dim": 12}, feature_batch=7, expected_total=n)
    assert calls == [7] * (n // 7) + ([n % 7] if n % 7 else [])
    assert get_existing_features(path, expected_total=n) == {"tiny12"}
    assert get_existing_features(path, expected_total=n + 1) == set()
    with h5.File(path, "r") as f:
        for name, spec in want["layout"]["datasets"].items():
            ds = f[name.split("/")[0]] if "/" not in name else f[name.split("/")[0]][name.split("/")[1]]
            assert ds.dtype.str == spec["dtype"] and list(ds.shape) == spec["shape"]
            assert list(ds.chunks) == spec["chunks"]
            assert [None if m is None else int(m) for m in ds.maxshape] == spec["maxshape"]
        mine = {k: f.attrs[k] for k in f.attrs.keys()}
        for k, v in want["layout"]["file_attr_values"].items():
            assert mine[k] == v, k
        assert set(mine) - set(want["layout"]["file_attr_values"]) == {"creation_date", "wsi_path"}
        assert np.array_equal(f["features"]["tiny12"][:], feats)
    assert np.array_equal(read_coords(path), arrays["coords"])
    with pytest.raises(ValueError):        # duplicate dataset guard (storage.py:269-272)
        writer.append_feature_matrix(output_path=path, feature_name="tiny12", features=feats,
                                     feature_attrs={"embedding_dim": 12}, feature_batch=7, expected_total=n)
    with pytest.raises(ValueError):        # row-count check (storage.py:323-326); partial dataset removed
        writer.append_feature_matrix(output_path=path, feature_name="other", features=feats[:5],
                                     feature_attrs={"embedding_dim": 12}, feature_batch=7, expected_total=n)
    assert get_existing_features(path) == {"tiny12"}
    dump = "/opt/conda/bin/h5dump"
    if os.path.exists(dump):
        out = subprocess.run([dump, "-H", str(path)], capture_output=True, text=True)
        assert out.returncode == 0 and 'DATASET "coords"' in out.stdout and "H5T_STD_I32LE" in out.stdout


def test_thumbnail_size_matches_pillow():
    from PIL import Image
    from atlaspatch_amd.core.wsi.synth_pixels import thumbnail_size
    for w, h in [(6250, 6250), (2875, 2057), (1024, 733), (500, 300), (2500, 1875), (1025, 1024), (3000, 1001), (7, 9000)]:
        im = Image.new("RGB", (w, h))
        im.thumbnail((1024, 1024))
        assert im.size == thumbnail_size(w, h, 1024), (w, h)


# ----------------------------------------------------------------------------- C ABI
def test_c_abi_library_loads_and_exports_every_declared_symbol():
    from atlaspatch_amd import _lib
    header = open(os.path.join(ROOT, "include", "atlaspatch_hip.h")).read()
    declared = set(re.findall(r"\b(ap_[a-z0-9_]+)\s*\(", header))
    declared -= {"ap_vit", "ap_contours"}
    assert declared, "no prototypes found"
    lib = ctypes.CDLL(_lib.library_path())
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/atlaspatch_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.load().ap_abi_version() == _lib.ABI_VERSION == 20


def test_product_has_no_cpu_fallback():
    from atlaspatch_amd import _lib
    from atlaspatch_amd.encoders.vit import ARCHS, HipViT, random_canonical_state_dict
    arch = dict(ARCHS["vit_b_16"]); arch["depth"] = 1
    with pytest.raises(_lib.HipLibraryError):
        HipViT(arch, random_canonical_state_dict(arch), device=torch.device("cpu"), dtype=torch.float32)
    if not torch.cuda.is_available():
        from atlaspatch_amd.utils.contours import mask_to_contours
        with pytest.raises(_lib.HipLibraryError):
            mask_to_contours(np.ones((8, 8), np.float32))


def test_product_never_imports_the_oracle():
    import glob
    for path in glob.glob(os.path.join(ROOT, "atlaspatch_amd", "**", "*.py"), recursive=True):
        text = open(path).read()
        assert "import oracle" not in text and "from oracle" not in text, path


def test_pillow_resample_tables_reproduce_pil():
    """The host-built coefficient tables (product code feeding ap_resample_u8) applied in numpy equal
    PIL.Image.resize bit for bit -- Pillow is the reference's resampler (timm / open_clip Resize on PIL images)."""
    from PIL import Image
    from atlaspatch_amd.utils.resample import pillow_resample_tables

    def apply(img, oh, ow, f):
        h, w, c = img.shape
        bx, kx, _ = pillow_resample_tables(w, ow, f)
        by, ky, _ = pillow_resample_tables(h, oh, f)
        a = img.astype(np.int64)
        tmp = np.zeros((h, ow, c), np.uint8)
        for xx in range(ow):
            s = np.full((h, c), 1 << 21, np.int64)
            for x in range(bx[xx, 1]):
                s += a[:, bx[xx, 0] + x, :] * int(kx[xx, x])
            tmp[:, xx] = np.clip(s >> 22, 0, 255)
        t = tmp.astype(np.int64)
        out = np.zeros((oh, ow, c), np.uint8)
        for yy in range(oh):
            s = np.full((ow, c), 1 << 21, np.int64)
            for y in range(by[yy, 1]):
                s += t[by[yy, 0] + y] * int(ky[yy, y])
            out[yy] = np.clip(s >> 22, 0, 255)
        return out

    rng = np.random.default_rng(1)
    for (h, w), (oh, ow) in (((256, 256), (224, 224)), ((256, 256), (448, 448)), ((131, 97), (302, 224))):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        for f, pf in (("bicubic", Image.Resampling.BICUBIC), ("bilinear", Image.Resampling.BILINEAR)):
            assert np.array_equal(apply(img, oh, ow, f), np.asarray(Image.fromarray(img).resize((ow, oh), pf))), (h, w, f)


def test_host_gather_tiles_copies_in_order():
    """ap_host_gather_tiles (host-only entry of the C ABI): n scattered tiles -> consecutive slots."""
    import ctypes as C
    from atlaspatch_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(0)
    tiles = [rng.integers(0, 256, (8, 8, 3), dtype=np.uint8) for _ in range(5)]
    dst = np.zeros((5, 8, 8, 3), np.uint8)
    ptrs = (C.c_void_p * 5)(*[t.ctypes.data for t in tiles])
    assert lib.ap_host_gather_tiles(dst.ctypes.data, ptrs, 5, 8 * 8 * 3) == 0
    assert all(np.array_equal(dst[i], tiles[i]) for i in range(5))
    assert lib.ap_host_gather_tiles(dst.ctypes.data, ptrs, 0, 8 * 8 * 3) == 0


def test_synth_slide_with_jpeg_tile_store(tmp_path):
    """``"jpeg_tiles"`` in a .synth descriptor: level-0 tiles present in the store are decoded with Pillow (others are
    rendered), and such a slide never offers a device tile source (its tiles cross the ring)."""
    import json
    from PIL import Image
    from atlaspatch_amd.core.wsi.synth_pixels import SynthSpec, render_region
    from atlaspatch_amd.core.wsi.wsi_factory import WSIFactory
    store = tmp_path / "tiles"
    store.mkdir()
    spec = SynthSpec(width=4096, height=4096, seed=3)
    marked = np.full((256, 256, 3), 77, np.uint8)
    Image.fromarray(marked).save(store / "512_768_256.jpg", quality=95)
    path = tmp_path / "s.synth"
    path.write_text(json.dumps({"width": 4096, "height": 4096, "seed": 3, "jpeg_tiles": "tiles"}))
    wsi = WSIFactory.load(str(path))
    got = wsi.extract((512, 768), 0, (256, 256))
    assert got.shape == (256, 256, 3) and abs(int(got.mean()) - 77) <= 1            # decoded, not rendered
    assert np.array_equal(wsi.extract((0, 0), 0, (256, 256)), render_region(spec, 0, 0, 256, 256, 0))   # not in the store
    assert wsi.extract_batch_device(np.array([[0, 0, 256, 256, 0]]), "cpu", 256) is None


def test_native_batched_decode_of_the_deflate_tile_store(tmp_path):
    """ap_host_inflate_tiles through SynthWSI.read_tiles_into: the chunk lands in consecutive slots, bit-equal to the
    stored pixels; a missing tile or a wrong size makes the capability decline (False) instead of guessing."""
    import json, zlib
    from atlaspatch_amd.core.wsi.wsi_factory import WSIFactory
    store = tmp_path / "tiles"
    store.mkdir()
    rng = np.random.default_rng(0)
    tiles = {(256 * i, 512): rng.integers(0, 256, (256, 256, 3), dtype=np.uint8) for i in range(4)}
    for (x, y), t in tiles.items():
        (store / f"{x}_{y}_256.z").write_bytes(zlib.compress(t.tobytes(), 1))
    path = tmp_path / "s.synth"
    path.write_text(json.dumps({"width": 4096, "height": 4096, "seed": 3, "jpeg_tiles": "tiles"}))
    wsi = WSIFactory.load(str(path))
    rows = [[x, y, 256, 256, 0] for (x, y) in tiles]
    dst = np.zeros((4, 256, 256, 3), np.uint8)
    assert wsi.read_tiles_into(rows, dst.ctypes.data, 256) is True
    for i, key in enumerate(tiles):
        assert np.array_equal(dst[i], tiles[key])
        assert np.array_equal(wsi.extract(key, 0, (256, 256)), tiles[key])          # the per-tile path reads the same store
    assert wsi.read_tiles_into(rows + [[0, 0, 256, 256, 0]], dst.ctypes.data, 256) is False
    assert wsi.read_tiles_into([[0, 512, 512, 512, 0]], dst.ctypes.data, 256) is False


# ----------------------------------------------------------------------------- OpenSlide property lookup (golden G8)
def test_openslide_mpp_and_mag_lookup_matches_reference_golden(golden_dir):
    """mpp_from_properties / mag_from_properties == the reference's OpenSlideWSI._extract_mpp / _extract_mag
    (core/wsi/openslide_wsi.py:71-147) on 28 fake property dicts (fixture produced by running the reference's methods,
    tests/golden/gen_golden.py::gen_openslide_props), and the key tables keep the reference's order."""
    import json
    from atlaspatch_amd.core.wsi import openslide_wsi as o
    from atlaspatch_amd.core.wsi.iwsi import IWSI
    with open(os.path.join(golden_dir, "openslide_props.json")) as fh:
        g = json.load(fh)
    assert list(o._MPP_KEYS) == g["keys"]["mpp"] and list(o._MAG_KEYS) == g["keys"]["mag"]
    assert list(o._MPP_TEXT_KEYS) == g["keys"]["text"]
    for case in g["cases"]:
        mpp = o.mpp_from_properties(case["meta"])
        mag = o.mag_from_properties(case["meta"], mpp, lambda m: IWSI._infer_mag(None, m))
        assert mpp == case["mpp"] and mag == case["mag"], case


def test_transform_resize_table():
    """vit_l_16 resolves torchvision's ViT_L_16_Weights.IMAGENET1K_V1 = ImageClassification(crop 224, resize 242):
    256-px tiles are resampled; vit_b_16 keeps resize 256 (pure centre crop)."""
    from atlaspatch_amd.encoders.vit import TRANSFORM_RESIZE
    assert TRANSFORM_RESIZE["vit_b_16"] == (256, "bilinear") and TRANSFORM_RESIZE["vit_l_16"] == (242, "bilinear")
    assert TRANSFORM_RESIZE["uni_v1"] == (224, "bicubic") and TRANSFORM_RESIZE["conch_v1"] == (448, "bicubic")


# ----------------------------------------------------------------------------- native host tile sources of the ring
def test_host_synth_tiles_equal_the_numpy_renderer():
    """ap_host_synth_tiles (the synthetic slide's native 'decoder' behind read_tiles_into) == render_region bit for bit,
    for level 0, a 512-px read and a downsampled level, including out-of-bounds pixels."""
    import ctypes as C
    from atlaspatch_amd import _lib
    from atlaspatch_amd.core.wsi.synth_pixels import SynthSpec, render_region
    lib = _lib.load()
    spec = SynthSpec(width=40000, height=40000)
    xy = np.array([[0, 0], [12800, 20224], [39900, 39900], [-100, 500], [20000, 20000]], np.int32)
    ell = np.ascontiguousarray(spec.ellipses())
    for side, ds, lv in ((256, 1, 0), (512, 1, 0), (64, 4, 1)):
        out = np.empty((len(xy), side, side, 3), np.uint8)
        _lib.check(lib.ap_host_synth_tiles(out.ctypes.data, xy.ctypes.data, len(xy), side, ds, lv, spec.width, spec.height,
                                           spec.seed, ell.ctypes.data, ell.shape[0]))
        for i, (x, y) in enumerate(xy):
            assert np.array_equal(out[i], render_region(spec, int(x), int(y), side, side, lv)), (side, i)


def test_native_jpeg_tile_decode_equals_pillow(tmp_path):
    """ap_host_decode_jpeg_tiles (system libjpeg-turbo behind restated declarations, validated by the library's own
    struct-size check and a load-time self-check) == PIL's Image.open(...).convert("RGB") bit for bit: 4:2:0 / 4:4:4
    chroma, two qualities, noise, a greyscale stream; a tile of the wrong size is an error, not garbage."""
    import ctypes as C
    from PIL import Image
    from atlaspatch_amd import _lib
    from atlaspatch_amd.core.wsi.synth_pixels import SynthSpec, render_region
    lib = _lib.load()
    spec = SynthSpec(width=40000, height=40000)
    rng = np.random.default_rng(0)
    paths, want = [], []
    for i, (x, y) in enumerate([(12800, 20224), (20000, 20000), (0, 0), (15000, 9000)]):
        tile = render_region(spec, x, y, 256, 256, 0) if i != 3 else rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)
        p = str(tmp_path / f"{i}.jpg")
        Image.fromarray(tile).save(p, quality=95 if i == 2 else 80, subsampling=0 if i == 1 else 2)
        paths.append(p)
    p = str(tmp_path / "grey.jpg")
    Image.fromarray(render_region(spec, 100, 100, 256, 256, 0)).convert("L").save(p, quality=80)
    paths.append(p)
    want = [np.asarray(Image.open(p).convert("RGB")) for p in paths]
    out = np.zeros((len(paths), 256, 256, 3), np.uint8)
    arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
    rc = lib.ap_host_decode_jpeg_tiles(out.ctypes.data, arr, len(paths), 256)
    if rc == _lib.AP_ERR_UNSUPPORTED:
        pytest.skip("no usable libjpeg.so.8 on this host: " + lib.ap_last_error().decode())
    _lib.check(rc, "ap_host_decode_jpeg_tiles")
    for i in range(len(paths)):
        assert np.array_equal(out[i], want[i]), i
    small = str(tmp_path / "small.jpg")
    Image.fromarray(want[0][:128, :128]).save(small)
    bad = (C.c_char_p * 1)(small.encode())
    assert lib.ap_host_decode_jpeg_tiles(out.ctypes.data, bad, 1, 256) == -1 and b"expected 256" in lib.ap_last_error()


# ----------------------------------------------------------------------------- visualisation overlays (f4)
def test_visualization_overlays_are_the_references_pillow_calls(tmp_path):
    """--visualize-mask / --visualize-grids restated from utils/visualization/{mask,patches}.py: same Pillow operations in
    the same order (checked here against an independent spelling of those operations); --visualize-contours draws the same
    scaled vertices (outline rasterisation differs from cv2.polylines: stated in the module)."""
    import json
    from PIL import Image, ImageDraw
    from atlaspatch_amd.core.config import ExtractionConfig, OutputConfig
    from atlaspatch_amd.core.models import ExtractionResult, Slide
    from atlaspatch_amd.core.wsi.wsi_factory import WSIFactory
    from atlaspatch_amd.services.visualization import DefaultVisualizationService
    path = tmp_path / "v.synth"
    path.write_text(json.dumps({"width": 20000, "height": 14000, "seed": 5}))
    wsi = WSIFactory.load(str(path))
    mask = wsi.tissue_mask(1024)
    coords = np.array([[0, 0, 256, 256, 0], [5120, 2560, 256, 256, 0], [19900, 13900, 256, 256, 0]], np.int32)
    out_cfg = OutputConfig(output_root=tmp_path / "out", visualize_grids=True, visualize_mask=True)   # contours: GPU suite
    ext_cfg = ExtractionConfig(patch_size=256, target_magnification=20)
    res = ExtractionResult(slide=Slide(path=path), h5_path=tmp_path / "none.h5", num_patches=3, coords=coords,
                           patch_size_level0=256)
    DefaultVisualizationService(out_cfg, ext_cfg).visualize(res, wsi=wsi, mask=mask)
    assert set(res.visualizations) == {"grids", "mask"}
    vis = res.visualizations["mask"].parent
    assert vis.name == "visualization" and (vis / "v_mask_bw.png").exists()
    thumb = wsi.get_thumb((1024, 1024)).convert("RGB")
    # mask overlay: alpha-composite of a green layer whose alpha is 80 where the (nearest-resized) mask is set
    m = Image.fromarray(((mask > 0.5) * 255).astype(np.uint8), mode="L").resize(thumb.size, Image.Resampling.NEAREST)
    layer = Image.new("RGBA", thumb.size, (0, 255, 0, 0))
    layer.putalpha(Image.fromarray((np.asarray(m, np.float32) / 255.0 * 80).astype(np.uint8), mode="L"))
    want = Image.alpha_composite(thumb.convert("RGBA"), layer).convert("RGB")
    assert np.array_equal(np.asarray(Image.open(res.visualizations["mask"])), np.asarray(want))
    assert np.array_equal(np.asarray(Image.open(vis / "v_mask_bw.png")), np.asarray(m))
    # grid overlay: rectangles at int(coords / ratio) .. int((coords + ps0) / ratio), away from the info box
    grid = np.asarray(Image.open(res.visualizations["grids"]))
    rx, ry = 20000 / thumb.width, 14000 / thumb.height
    x0, y0 = int(np.float32(5120) / rx), int(np.float32(2560) / ry)
    assert tuple(grid[y0, x0]) == (0, 0, 0) and tuple(grid[y0, int((5120 + 256) / rx)]) == (0, 0, 0)


def test_sam2_checkpoint_loader_accepts_the_reference_layout_and_names_what_is_missing(tmp_path):
    """load_sam2_state_dict: the reference's {"model": state_dict} layout (segmentation.py:66-67), a DataParallel prefix, and a
    clear KeyError when a tensor the image path reads is absent (names: the sam2 package's state dict, restated in the oracle)."""
    from atlaspatch_amd.services.sam2_hip import load_sam2_state_dict, required_sam2_keys
    from oracle import sam2_oracle as so
    sd = so.random_state_dict(0)
    req = required_sam2_keys()
    assert len(req) == 269 and all(k in sd for k in req)
    p = tmp_path / "model.pth"
    torch.save({"model": {"module." + k: v for k, v in sd.items()}}, p)
    assert set(load_sam2_state_dict(p)) == set(sd)
    broken = dict(sd)
    broken.pop("sam_mask_decoder.iou_token.weight")
    torch.save({"model": broken}, p)
    with pytest.raises(KeyError, match="iou_token"):
        load_sam2_state_dict(p)


def test_host_scale_contours_matches_reference_golden(golden_dir):
    """utils.contours.scale_contours (host helper of the visualisation overlay) == the reference's scale_contours on G3."""
    from atlaspatch_amd.utils.contours import scale_contours
    g = np.load(os.path.join(golden_dir, "scale_contours.npz"))
    k = 0
    while f"c{k}_in" in g:
        W, H, mw, mh = (int(v) for v in g[f"c{k}_dims"])
        out = scale_contours([g[f"c{k}_in"].reshape(-1, 1, 2)], W / float(mw), H / float(mh))[0]
        assert out.dtype == np.int32 and np.array_equal(out.reshape(-1, 2), g[f"c{k}_out"].reshape(-1, 2)), k
        k += 1
    assert k >= 4


def test_detect_tissue_command_writes_mask_overlays(tmp_path):
    """`detect-tissue` (reference cli.py:329-438, 531-578): segmentation only + mask overlay PNGs under <out>/visualization,
    the reference's summary line.  Synthetic slides use the analytic mask, so this runs without a device."""
    import json
    from click.testing import CliRunner
    from atlaspatch_amd.cli import cli
    for name, seed in (("a.synth", 3), ("b.synth", 4)):
        (tmp_path / name).write_text(json.dumps({"width": 12000, "height": 9000, "seed": seed}))
    out = tmp_path / "o"
    res = CliRunner().invoke(cli, ["detect-tissue", str(tmp_path), "-o", str(out), "--seg-batch-size", "2"], catch_exceptions=False)
    assert res.exit_code == 0 and "Created 2 mask overlay(s), failures: 0" in res.output
    assert sorted(p.name for p in (out / "visualization").iterdir()) == ["a_mask.png", "a_mask_bw.png", "b_mask.png", "b_mask_bw.png"]

def test_process_exposes_the_feature_gather_as_a_flag():
    """The one collective of the multi-GPU path is a CLI flag (MI355X addition, next to the ATLASPATCH_GATHER_FEATURES
    environment variable), named in `process --help`; `segment-and-get-coords` does not carry it; with one rank it is a no-op
    that needs no process group."""
    import inspect
    from click.testing import CliRunner
    from atlaspatch_amd import cli as cli_mod
    res = CliRunner().invoke(cli_mod.cli, ["process", "--help"])
    assert res.exit_code == 0 and "--gather-features / --no-gather-features" in res.output
    res = CliRunner().invoke(cli_mod.cli, ["segment-and-get-coords", "--help"])
    assert res.exit_code == 0 and "--gather-features" not in res.output
    assert "gather_features" in inspect.signature(cli_mod._run_pipeline).parameters
    from atlaspatch_amd.services.feature_embedding import PatchFeatureEmbeddingService
    assert "keep_feature_blocks" in inspect.signature(PatchFeatureEmbeddingService.__init__).parameters


def test_h5_file_equals_the_file_the_references_writer_makes_with_real_h5py(tmp_path, golden_dir):
    """G5b: tests/golden/reference_real.h5 was written by the reference's own H5PatchWriter (write_coords + append_features,
    unmodified) under the REAL h5py of the image's conda interpreter (gen_golden_h5_real.py).  The build's writer, given the
    same inputs, must produce the same file: every dataset's values read back equal, and -- through the real h5py when
    that interpreter is present (it is in this image, here and on the GPU box) -- the same names, dtypes, shapes, chunk
    shapes, max shapes, fill values and attribute names / types / values."""
    import json
    import os
    import subprocess
    from atlaspatch_amd.services.storage import H5PatchWriter
    from atlaspatch_amd.utils.h5 import h5
    inp = np.load(os.path.join(golden_dir, "reference_real_h5_inputs.npz"))
    coords, feats = inp["coords"], inp["feats"]
    n = coords.shape[0]
    out = tmp_path / "ours.h5"
    w = H5PatchWriter(chunk_rows=8, patch_size=256, patch_size_level0=512, level0_mag=40, target_mag=20,
                      level0_wh=(100000, 90000), overlap=0, slide_stem="slide_A", wsi_path="/data/slide_A.svs",
                      total_patches=None, extra_file_attrs={"mpp": 0.2528})
    entries = [tuple(int(v) for v in row) + (None,) for row in coords]
    total, _ = w.write_coords(out, entries, batch=10)
    assert total == n
    patches = [np.full((2, 2, 3), i, dtype=np.uint8) for i in range(n)]
    wrote = w.append_features(output_path=out, entries=[e[:5] + (p,) for e, p in zip(entries, patches)], feature_name="tiny12",
                              feature_fn=lambda batch: feats[[int(p[0, 0, 0]) for p in batch]],
                              feature_attrs={"embedding_dim": 12, "source": "unit"}, feature_batch=7, expected_total=n)
    assert wrote == n
    ref_path = os.path.join(golden_dir, "reference_real.h5")
    with h5.File(ref_path, "r") as fr, h5.File(str(out), "r") as fo:
        assert sorted(fr.keys()) == sorted(fo.keys()) == ["coords", "features", "passports"]
        assert np.array_equal(fr["coords"][:], fo["coords"][:]) and np.array_equal(fo["coords"][:], coords)
        assert np.array_equal(fr["passports"][:], fo["passports"][:])
        assert np.array_equal(fr["features"]["tiny12"][:], fo["features"]["tiny12"][:])
        assert fo["features"]["tiny12"].dtype == np.float32 and fo["passports"].dtype == np.dtype("S160")
    conda_py = "/opt/conda/bin/python3.9"
    if not os.path.exists(conda_py):
        pytest.skip("no interpreter with the real h5py here: values compared, structural dump skipped")
    got = json.loads(subprocess.run([conda_py, os.path.join(golden_dir, "describe_h5_real.py"), str(out)], check=True,
                                    capture_output=True, text=True).stdout)
    with open(os.path.join(golden_dir, "reference_real_h5.json")) as fh:
        want = json.load(fh)
    got["file_attrs"].pop("creation_date", None)
    assert got["datasets"] == want["datasets"]
    assert got["file_attrs"] == want["file_attrs"]


def _fake_sysfs(root, *, nodes, smt, gpu_nodes):
    """nodes: {node: [cpus]}; smt: {cpu: sibling}; gpu_nodes: {bdf: node}."""
    for node, cpus in nodes.items():
        d = root / "devices/system/node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(",".join(str(c) for c in cpus) + "\n")
    for cpu, sib in smt.items():
        d = root / "devices/system/cpu" / f"cpu{cpu}" / "topology"
        d.mkdir(parents=True)
        (d / "thread_siblings_list").write_text(f"{min(cpu, sib)},{max(cpu, sib)}\n")
    for bdf, node in gpu_nodes.items():
        d = root / "bus/pci/devices" / bdf
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{node}\n")


def test_pin_order_is_numa_local_first_and_disjoint_across_local_ranks(tmp_path):
    """The ring's decode threads (north star: 'tile decode on host cores pinned'): the GPU's NUMA node first, physical
    cores before SMT siblings, and the ranks of one node take DISJOINT cores (every rank used to pin to the
    same cores because the PCI address lookup raised and the counter started at 0 in every rank)."""
    from atlaspatch_amd.services.tile_ring import _pin_order
    # 2 sockets x 8 cores x 2 threads: cpus 0-7 / 8-15 are the cores, 16-23 / 24-31 their SMT siblings
    nodes = {0: list(range(0, 8)) + list(range(16, 24)), 1: list(range(8, 16)) + list(range(24, 32))}
    smt = {c: c + 16 for c in range(16)}
    smt.update({c + 16: c for c in range(16)})
    gpus = {"0000:05:00.0": 0, "0000:85:00.0": 1}
    _fake_sysfs(tmp_path, nodes=nodes, smt=smt, gpu_nodes=gpus)
    allowed = set(range(32))
    kw = dict(sysfs=str(tmp_path), allowed=allowed)
    one = _pin_order(None, bdf="0000:05:00.0", local_rank=0, local_world=1, **kw)
    assert one == list(range(0, 8)) + list(range(8, 16)) + list(range(16, 24)) + list(range(24, 32))
    other = _pin_order(None, bdf="0000:85:00.0", local_rank=0, local_world=1, **kw)
    assert other[:8] == list(range(8, 16)) and other[8:16] == list(range(0, 8))
    # four ranks, two per socket: the first cores each rank would take (1 + workers) never collide
    orders = [_pin_order(None, bdf=b, local_rank=r, local_world=4, **kw)
              for r, b in enumerate(["0000:05:00.0", "0000:05:00.0", "0000:85:00.0", "0000:85:00.0"])]
    heads = [tuple(o[:2]) for o in orders]
    assert heads == [(0, 4), (1, 5), (10, 14), (11, 15)]
    flat = [c for h in heads for c in h]
    assert len(set(flat)) == len(flat)
    for o, node in zip(orders, (0, 0, 1, 1)):
        assert set(o[:2]) <= set(nodes[node]) and len(set(o)) == len(o) == 8
    assert not (set(orders[0]) & set(orders[1])) and not (set(orders[2]) & set(orders[3]))
    # a restricted affinity mask is honoured; an unknown device falls back to "every allowed CPU is local"
    assert _pin_order(None, bdf="0000:05:00.0", local_rank=0, local_world=1, sysfs=str(tmp_path), allowed={2, 3, 18, 9}) == [2, 3, 9, 18]
    assert _pin_order(None, bdf="ffff:ff:1f.0", local_rank=0, local_world=1, **kw)[:16] == list(range(16))
    # LOCAL_RANK / LOCAL_WORLD_SIZE from the launcher's environment
    import os
    old = {k: os.environ.get(k) for k in ("LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    os.environ.update(LOCAL_RANK="1", LOCAL_WORLD_SIZE="4")
    try:
        assert _pin_order(None, bdf="0000:05:00.0", **kw) == orders[1]
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)


def test_native_passports_equal_the_python_format_string():
    """H5PatchWriter._passports (ap_host_format_passports) == the per-row f-string of services/storage.py:387-392 for
    every sign / width of the five integers, long stems (S160 truncation), mag / target-mag 0 -> "na"."""
    from atlaspatch_amd.services.storage import H5PatchWriter
    rng = np.random.default_rng(0)
    for stem, mag, tmag in [("s000", 20, 20), ("a very long slide name " * 8, 40, 0), ("x", 0, 20), ("CMU-1", 20, 40)]:
        w = H5PatchWriter(chunk_rows=8192, patch_size=256, patch_size_level0=256, level0_mag=mag, target_mag=tmag,
                          level0_wh=(100000, 100000), overlap=0, slide_stem=stem, wsi_path="/x")
        w.total_patches = 58938
        block = np.concatenate([rng.integers(-5000, 2 ** 31 - 1, (2000, 5)), rng.integers(0, 100000, (2000, 5)),
                                np.array([[0, 0, 0, 0, 0], [-2 ** 31, 2 ** 31 - 1, -1, 10, 9]])]).astype(np.int32)
        got = w._passports(block)
        want = np.asarray([w._passport(*r) for r in block.tolist()], dtype="S160")
        assert got.dtype == want.dtype and np.array_equal(got, want), stem
    w = H5PatchWriter(chunk_rows=8, patch_size=256, patch_size_level0=256, level0_mag=20, target_mag=20, level0_wh=(10, 10),
                      overlap=0, slide_stem="s", wsi_path="/x")
    with pytest.raises(RuntimeError):
        w._passports(np.zeros((1, 5), np.int32))          # total_patches not set, like _passport


def test_h5_backend_survives_concurrent_writers_and_readers(tmp_path):
    """The coordinate path writes one H5 per slide from worker threads (orchestration/runner.py) while other threads read:
    the ctypes binding over a non-thread-safe libhdf5 serialises every library call.  16 threads write and re-read their
    own files at once; every file must come back intact."""
    import concurrent.futures as futures
    from atlaspatch_amd.services.storage import H5PatchWriter, read_coords
    from atlaspatch_amd.utils.h5 import h5

    def work(i):
        rng = np.random.default_rng(i)
        n = int(rng.integers(1000, 20000))
        coords = rng.integers(0, 100000, (n, 5)).astype(np.int32)
        path = tmp_path / f"s{i}.h5"
        w = H5PatchWriter(chunk_rows=2048, patch_size=256, patch_size_level0=256, level0_mag=20, target_mag=20,
                          level0_wh=(100000, 100000), overlap=0, slide_stem=f"s{i}", wsi_path=f"/x/s{i}")
        assert w.write_coords_array(path, coords) == n
        feats = rng.standard_normal((n, 64)).astype(np.float32)
        w.append_feature_matrix(output_path=path, feature_name="enc", features=feats, feature_attrs={"name": "enc", "embedding_dim": 64},
                                feature_batch=512, expected_total=n)
        assert np.array_equal(read_coords(path), coords)
        with h5.File(path, "r") as f:
            assert np.array_equal(f["features"]["enc"][:], feats) and int(f.attrs["num_patches"]) == n
            first = f["passports"][0]
        assert bytes(first).rstrip(b"\\0").decode().startswith(f"s{i}__x{coords[0, 0]}_y{coords[0, 1]}")
        return n

    with futures.ThreadPoolExecutor(16) as pool:
        assert all(n > 0 for n in pool.map(work, range(32)))


def test_h5_writer_processes_write_the_same_file_as_the_in_process_writer(tmp_path):
    """services/h5_writer_proc.py: cohort runs hand each slide's coords file to a helper process (libhdf5 is one lock per
    process).  Same H5PatchWriter code in the child -> same datasets, chunking and attributes; a broken pool means None (the
    caller writes in-process)."""
    import time
    from atlaspatch_amd.services.h5_writer_proc import H5WriterPool
    from atlaspatch_amd.services.storage import H5PatchWriter
    from atlaspatch_amd.utils.h5 import h5
    rng = np.random.default_rng(5)
    n = 20001
    coords = np.stack([rng.integers(0, 99000, n), rng.integers(0, 99000, n), np.full(n, 256), np.full(n, 256), np.zeros(n, int)], 1).astype(np.int32)

    def mk():
        return H5PatchWriter(chunk_rows=8192, patch_size=256, patch_size_level0=256, level0_mag=20, target_mag=20,
                             level0_wh=(100000, 99000), overlap=0, slide_stem="slide_a", wsi_path="/data/slide_a.svs",
                             extra_file_attrs={"filename": "slide_a.svs", "mpp": 0.5, "vendor": "synthetic"})
    pool = H5WriterPool(2)
    assert pool.write({}, "x", coords, np.zeros(n, "S160")) is None and not pool.ready()      # nothing started yet: in-process
    pool.prestart()
    t0 = time.time()
    while not pool.ready() and time.time() - t0 < 60:
        time.sleep(0.02)
    assert pool.ready()
    w = mk()
    assert pool.write(w.to_kwargs(), str(tmp_path / "proc.h5"), coords, w.passports_array(coords)) == n
    assert pool.write(mk().to_kwargs(), str(tmp_path / "empty.h5"), coords[:0], mk().passports_array(coords[:0])) == 0
    # a write the child reports as failed: None like every other failure mode (the caller repeats it in-process, where a real
    # error raises with its own traceback), counted, and the child lives on
    assert pool.write(mk().to_kwargs(), str(tmp_path / "no_such_dir" / "x.h5"), coords, mk().passports_array(coords)) is None
    assert pool.child_errors == 1 and pool.restarts == 0
    assert pool.write(mk().to_kwargs(), str(tmp_path / "again.h5"), coords[:7], mk().passports_array(coords[:7])) == 7
    pool.close()
    assert mk().write_coords_array(tmp_path / "here.h5", coords) == n
    with h5.File(tmp_path / "proc.h5", "r") as a, h5.File(tmp_path / "here.h5", "r") as b:
        for name in ("coords", "passports"):
            assert a[name].shape == b[name].shape and a[name].dtype == b[name].dtype and a[name].chunks == b[name].chunks
            assert a[name].maxshape == b[name].maxshape and np.array_equal(a[name][:], b[name][:])
        assert sorted(a.attrs.keys()) == sorted(b.attrs.keys())
        for k in a.attrs.keys():
            if k != "creation_date":
                assert a.attrs[k] == b.attrs[k], k
    with h5.File(tmp_path / "empty.h5", "r") as e:
        assert e["coords"].shape == (0, 5) and e.attrs["num_patches"] == 0
    assert not [p for p in tmp_path.iterdir() if ".tmp." in p.name]


def test_h5_writer_pool_survives_a_hung_and_a_dead_child(tmp_path, monkeypatch):
    """A stopped child (SIGSTOP: what an NFS stall looks like from outside) is killed by the watchdog after
    ATLASPATCH_H5_PROC_TIMEOUT seconds, the call returns None (in-process write), and a replacement is started: `broken` is not
    sticky, the next call is served by a process again.  Noise a C library prints on fd 1 of a child cannot reach the pipe."""
    import signal
    import time
    from atlaspatch_amd.services import h5_writer_proc as hp
    from atlaspatch_amd.services.storage import H5PatchWriter
    monkeypatch.setenv("ATLASPATCH_H5_PROC_TIMEOUT", "1.5")
    coords = np.array([[0, 0, 256, 256, 0], [256, 0, 256, 256, 0]], dtype=np.int32)

    def mk():
        return H5PatchWriter(chunk_rows=8192, patch_size=256, patch_size_level0=256, level0_mag=20, target_mag=20,
                             level0_wh=(1000, 1000), overlap=0, slide_stem="s", wsi_path="/data/s.svs")

    def wait_ready(pool, want=1):
        t0 = time.time()
        while time.time() - t0 < 60:
            with pool._lock:
                if pool._ready >= want:
                    return True
            time.sleep(0.02)
        return False

    pool = hp.H5WriterPool(1)
    pool.prestart()
    assert wait_ready(pool)
    victim = pool._idle.get()
    pool._idle.put(victim)
    os.kill(victim.proc.pid, signal.SIGSTOP)
    t0 = time.time()
    assert pool.write(mk().to_kwargs(), str(tmp_path / "hung.h5"), coords, mk().passports_array(coords)) is None
    assert 1.0 < time.time() - t0 < 20 and pool.restarts == 1
    assert victim.proc.poll() is not None                                  # reaped, not left stopped
    assert not list(tmp_path.glob(".hung.h5.tmp.*"))                       # nothing of the killed child's is left beside the target
    # (the hello of the replacement has its own limit -- ATLASPATCH_H5_PROC_START_TIMEOUT, >= 60 s -- not the 1.5-s job limit)
    assert hp._start_timeout() >= 60 and abs(hp._job_timeout(20000) - 3.5) < 1e-9
    (tmp_path / ".x.h5.tmp.deadbeef").write_bytes(b"half a file")
    (tmp_path / ".x.h5.tmp.cafe").write_bytes(b"another")
    (tmp_path / ".y.h5.tmp.keep").write_bytes(b"someone else's")
    hp._remove_leftovers(str(tmp_path / "x.h5"))
    assert sorted(p.name for p in tmp_path.glob(".*.tmp.*")) == [".y.h5.tmp.keep"]
    assert wait_ready(pool) and not pool.broken                            # the replacement said hello
    assert pool.write(mk().to_kwargs(), str(tmp_path / "after.h5"), coords, mk().passports_array(coords)) == 2
    # a child that dies between jobs
    victim = pool._idle.get()
    pool._idle.put(victim)
    victim.proc.kill()
    victim.proc.wait()
    assert pool.write(mk().to_kwargs(), str(tmp_path / "dead.h5"), coords, mk().passports_array(coords)) is None
    assert wait_ready(pool)
    assert pool.write(mk().to_kwargs(), str(tmp_path / "after2.h5"), coords, mk().passports_array(coords)) == 2
    pool.close()
    # fd 1 of a child is not the protocol pipe: a child that writes to it C-side before every frame still talks cleanly
    import subprocess
    import sys as _sys
    code = ("import os, sys; from atlaspatch_amd.services import h5_writer_proc as hp; "
            "real = hp._send\n"
            "def noisy(stream, obj, payload=b''):\n"
            "    os.write(1, b'libhdf5 says hello on stdout\\n'); real(stream, obj, payload)\n"
            "hp._send = noisy; hp._serve()")
    env = dict(os.environ, PYTHONPATH=ROOT)
    proc = subprocess.Popen([_sys.executable, "-c", code], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert hp._recv(proc.stdout) == {"ready": True}
    hp._send(proc.stdin, {"quit": True})
    proc.stdin.close()
    assert proc.wait(timeout=30) == 0 and b"libhdf5 says hello" in proc.stderr.read()


def test_env_flag_zero_never_switches_a_feature_on(monkeypatch):
    """ATLASPATCH_GATHER_FEATURES=0 (or false / no / off) must not enable the collective: one parser, both call sites."""
    from atlaspatch_amd.utils.env import env_flag
    for raw, want in (("1", True), ("yes", True), ("true", True), ("0", False), ("false", False), ("No", False), ("OFF", False), (" ", False),
                      ("", False)):
        monkeypatch.setenv("ATLASPATCH_GATHER_FEATURES", raw)
        assert env_flag("ATLASPATCH_GATHER_FEATURES") is want, raw
    monkeypatch.delenv("ATLASPATCH_GATHER_FEATURES")
    assert env_flag("ATLASPATCH_GATHER_FEATURES") is False and env_flag("ATLASPATCH_GATHER_FEATURES", True) is True
    import inspect
    from atlaspatch_amd import cli
    from atlaspatch_amd.services import feature_embedding
    for mod in (cli, feature_embedding):
        src = inspect.getsource(mod)
        assert 'env_flag("ATLASPATCH_GATHER_FEATURES")' in src and 'os.environ.get("ATLASPATCH_GATHER_FEATURES")' not in src

