"""Generate the golden fixtures in tests/golden/ by RUNNING THE REFERENCE ITSELF.

Run once, in the build container (the only place /root/reference exists):

    python tests/golden/gen_golden.py

It imports the reference's own modules through ``_ref_harness.install()`` (stub
packages + shim cv2 from oracle/cv2_restated.py + fake h5py) and records inputs
and outputs only -- no reference source is copied.  Fixtures written:

  geometry.json        G2  optimal_level / _prepare_geometry table
  scale_contours.npz   G3  scale_contours f32-truncation vectors
  coords_cases.npz     G4  masks (bit-packed) -> coords int32 [N,5] for >=10 cases, produced by the
                           reference's PatchExtractionService.extract + H5PatchWriter.write_coords
  coords_cases.json    G4/G5  per-case config, passports (first/last), file attrs, H5 layout dump
  contours_cases.npz   reference mask_to_contours outputs (point lists, hole grouping) per case
  extract_batch.npz    G1  seeded patches -> reference PatchFeatureExtractor.extract_batch outputs
                           (HF ViTModel ViT-B/16-shaped, reduced depth to keep the fixture small,
                           and the full 12-layer model for n=5)
  features_h5.json     G5  layout of features/<name> after the reference's embed_all
  config_cases.json    G7  validated() accept/reject table, resolve_feature_dtype table,
                           registry behaviour, parse_feature_list errors
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent.parent))

import _ref_harness  # noqa: E402

ref = _ref_harness.install()

from atlaspatch_amd.core.wsi.synth_pixels import (SynthSpec, analytic_mask, inside_any,  # noqa: E402
                                                  render_region)


# ----------------------------------------------------------------------------- synthetic IWSI
class RefSynthWSI(ref.iwsi.IWSI):
    """Synthetic slide behind the REFERENCE's IWSI base class."""

    def __init__(self, path, *, width, height, mag, downsamples, seed=1234, n_ellipses=12, mpp=0.5):
        super().__init__(path=path, mpp=None)
        self._spec = SynthSpec(width=width, height=height, seed=seed, n_ellipses=n_ellipses,
                               mag=mag, mpp=mpp, downsamples=tuple(downsamples))

    def _setup(self):
        s = self._spec
        self.w, self.h = s.width, s.height
        self.ds = [float(d) for d in s.downsamples]
        self.nlvl = len(self.ds)
        self.dims = [(int(round(s.width / d)), int(round(s.height / d))) for d in self.ds]
        self.meta = {"openslide.vendor": "synthetic", "synth.seed": str(s.seed)}
        self.mpp = s.mpp
        self.mag = s.mag

    def _extract_mpp(self):
        return self._spec.mpp

    def _extract_mag(self):
        return self._spec.mag

    def extract(self, xy, lv, wh, *, mode="array"):
        self._ensure_loaded()
        return render_region(self._spec, int(xy[0]), int(xy[1]), int(wh[0]), int(wh[1]), int(lv))

    def get_size(self, lv=0):
        self._ensure_loaded()
        return self.dims[lv]

    def get_thumb(self, max_hw):
        raise NotImplementedError

    def cleanup(self):
        self._loaded = False


# ----------------------------------------------------------------------------- masks
def ellipse_mask(h, w, items):
    """items: list of (cx, cy, a, b, value) in mask pixels, painted in order."""
    yy, xx = np.mgrid[0:h, 0:w]
    m = np.zeros((h, w), dtype=np.float32)
    for cx, cy, a, b, v in items:
        inside = ((xx - cx) * b) ** 2 + ((yy - cy) * a) ** 2 <= (a * b) ** 2
        m[inside] = v
    return m


def build_cases():
    rng = np.random.default_rng(20240917)
    cases = []

    def add(name, mask, **kw):
        cfg = dict(width=40000, height=40000, mag=20, downsamples=[1.0, 4.0, 16.0],
                   patch_size=256, step_size=None, target_mag=20, tissue_thresh=0.0)
        cfg.update(kw)
        cases.append((name, mask.astype(np.float32), cfg))

    add("single_blob", ellipse_mask(1024, 1024, [(500, 520, 300, 210, 1.0)]))
    add("blob_with_holes", ellipse_mask(1024, 1024, [
        (512, 512, 400, 330, 1.0), (400, 420, 80, 60, 0.0), (650, 600, 40, 55, 0.0),
        (520, 300, 2, 2, 0.0), (300, 640, 30, 12, 0.0)]))
    add("island_in_hole", ellipse_mask(1024, 1024, [
        (512, 512, 420, 400, 1.0), (512, 512, 250, 220, 0.0), (512, 512, 120, 90, 1.0),
        (512, 512, 40, 30, 0.0), (512, 512, 10, 8, 1.0)]))
    # many fragments + more than ten holes -> exercises the global top-10 hole rule
    items = []
    for _ in range(40):
        cx, cy = rng.integers(40, 984, 2)
        a, b = rng.integers(4, 60, 2)
        items.append((int(cx), int(cy), int(a), int(b), 1.0))
    for _ in range(30):
        cx, cy = rng.integers(100, 900, 2)
        a, b = rng.integers(3, 14, 2)
        items.append((int(cx), int(cy), int(a), int(b), 0.0))
    add("fragments", ellipse_mask(1024, 1024, items))
    noise = (rng.random((1024, 1024)) > 0.5).astype(np.float32)
    smooth = noise.copy()
    for _ in range(3):   # cheap blur -> blobby random field
        smooth = (smooth + np.roll(smooth, 1, 0) + np.roll(smooth, -1, 0)
                  + np.roll(smooth, 1, 1) + np.roll(smooth, -1, 1)) / 5.0
    add("noise_field", (smooth > 0.52).astype(np.float32), tissue_thresh=0.0005)
    add("touching_border", ellipse_mask(1024, 1024, [
        (0, 0, 300, 260, 1.0), (1023, 600, 200, 330, 1.0), (500, 1023, 260, 120, 1.0),
        (80, 90, 30, 30, 0.0)]))
    add("empty", np.zeros((1024, 1024), dtype=np.float32))
    add("full", np.ones((256, 256), dtype=np.float32), width=10000, height=10000)
    add("cmu1_like", ellipse_mask(733, 1024, [(480, 360, 330, 250, 1.0), (430, 330, 70, 50, 0.0),
                                              (900, 80, 60, 50, 1.0)]),
        width=46000, height=32914, downsamples=[1.0, 4.000121536217793, 16.00097], tissue_thresh=0.0)
    add("w99999", ellipse_mask(1024, 1024, [(600, 480, 350, 380, 1.0), (640, 500, 90, 110, 0.0)]),
        width=99999, height=99999)
    add("mag40_to_20", ellipse_mask(512, 512, [(256, 256, 180, 140, 1.0), (300, 260, 40, 30, 0.0)]),
        width=60000, height=60000, mag=40, downsamples=[1.0, 2.0, 4.0, 16.0], target_mag=20)
    add("overlap_step128", ellipse_mask(1024, 1024, [(400, 600, 150, 220, 1.0)]),
        step_size=128, width=30000, height=30000)
    add("thresh_1pct", ellipse_mask(1024, 1024, [(300, 300, 150, 150, 1.0), (800, 800, 40, 40, 1.0),
                                                 (820, 200, 70, 50, 1.0)]), tissue_thresh=0.01)
    add("patch224", ellipse_mask(1024, 1024, [(512, 400, 260, 200, 1.0), (500, 390, 60, 50, 0.0)]),
        patch_size=224, width=50000, height=45000)
    # analytic synthetic-slide mask (the one the bench uses), 40k
    spec = SynthSpec(width=40000, height=40000)
    add("synth40k", analytic_mask(spec), width=40000, height=40000)
    holes14 = [(512, 300, 330, 230, 1.0), (512, 780, 300, 180, 1.0)]
    for i in range(9):
        holes14.append((260 + 60 * i, 300 + (i % 3) * 40 - 40, 9 + (i % 4) * 3, 8 + (i % 3) * 4, 0.0))
    for i in range(6):
        holes14.append((300 + 80 * i, 780, 10, 10, 0.0))      # six equal-area holes: tie handling
    add("many_holes", ellipse_mask(1024, 1024, holes14), width=70000, height=70000)
    thin = np.zeros((300, 400), dtype=np.float32)
    thin[50, 20:380] = 1
    thin[50:250, 200] = 1
    thin[100:104, 100:300] = 1
    thin[150, 150] = 1
    thin[200:203, 50:53] = 1
    thin[201, 51] = 0
    add("thin_structures", thin, width=20000, height=15000)
    return cases


def run_coords_case(name, mask, cfg, tmp):
    wsi = RefSynthWSI(os.path.join(tmp, f"{name}.synth"), width=cfg["width"], height=cfg["height"],
                      mag=cfg["mag"], downsamples=cfg["downsamples"])
    ecfg = ref.config.ExtractionConfig(patch_size=cfg["patch_size"], step_size=cfg["step_size"],
                                       target_magnification=cfg["target_mag"],
                                       tissue_threshold=cfg["tissue_thresh"])
    ocfg = ref.config.OutputConfig(output_root=Path(tmp) / f"out_{name}")
    svc = ref.extraction.PatchExtractionService(ecfg, ocfg)
    slide = ref.models.Slide(path=Path(wsi.path))
    result = svc.extract(wsi, mask, slide=slide)
    f = _ref_harness.read_h5(result.h5_path)
    coords = np.array(f["coords"][:], dtype=np.int32)
    passports = f["passports"][:]
    layout = _ref_harness.describe_h5(result.h5_path)
    tissue, holes = ref.contours.mask_to_contours(mask, tissue_area_thresh=cfg["tissue_thresh"])
    geom = svc._prepare_geometry(wsi)
    info = {
        "config": cfg, "num_patches": int(result.num_patches),
        "patch_size_level0": int(result.patch_size_level0),
        "geometry": [int(geom[0]), [int(geom[1][0]), int(geom[1][1])], int(geom[2]), int(geom[3]),
                     int(geom[4])],
        "passport_first": passports[0].decode() if len(passports) else None,
        "passport_last": passports[-1].decode() if len(passports) else None,
        "n_tissue": len(tissue), "n_holes": [len(h) for h in holes],
        "layout": layout,
    }
    return coords, info, tissue, holes, result, wsi, svc


def gen_coords(out_dir):
    arrays, meta, cont_arrays = {}, {}, {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, mask, cfg in build_cases():
            coords, info, tissue, holes, *_ = run_coords_case(name, mask, cfg, tmp)
            info["layout"]["file_attr_values"].pop("creation_date", None)
            info["layout"]["file_attr_values"].pop("wsi_path", None)
            arrays[f"{name}__mask_bits"] = np.packbits(mask > 0.5)
            arrays[f"{name}__mask_shape"] = np.array(mask.shape, dtype=np.int32)
            arrays[f"{name}__coords"] = coords
            meta[name] = info
            # contour point lists (unscaled, mask space): lengths + concatenated points
            all_c = list(tissue) + [h for hs in holes for h in hs]
            cont_arrays[f"{name}__lens"] = np.array([c.shape[0] for c in all_c], dtype=np.int32)
            cont_arrays[f"{name}__pts"] = (np.concatenate([c.reshape(-1, 2) for c in all_c], 0)
                                           if all_c else np.zeros((0, 2), np.int32)).astype(np.int32)
            cont_arrays[f"{name}__nholes"] = np.array([len(h) for h in holes], dtype=np.int32)
            print(f"  coords case {name}: N={coords.shape[0]} tissue={len(tissue)} "
                  f"holes={[len(h) for h in holes][:8]}")
    np.savez_compressed(out_dir / "coords_cases.npz", **arrays)
    np.savez_compressed(out_dir / "contours_cases.npz", **cont_arrays)
    (out_dir / "coords_cases.json").write_text(json.dumps(meta, indent=1, sort_keys=True))


def gen_geometry(out_dir):
    class _G(ref.iwsi.IWSI):
        def __init__(self, ds, mag):
            super().__init__(path="x")
            self.ds, self.mag, self._loaded = ds, mag, True

        _setup = _extract_mpp = _extract_mag = extract = get_size = get_thumb = cleanup = (
            lambda self, *a, **k: None)

    rows = []
    ds_lists = [[1.0], [1.0, 4.0, 16.0], [1.0, 2.0, 4.0], [1.0, 4.000121536217793, 16.00097],
                [1.0, 4.0, 16.0, 32.0], [1.0, 2.0, 4.0, 8.0, 16.0, 32.0], [1.0, 3.9999, 16.02],
                [2.0, 8.0]]
    for ds in ds_lists:
        for mag in (20, 40, 10, 80, None):
            for tgt in (5, 10, 20, 40):
                for ps, step in ((256, None), (224, None), (512, 256), (256, 128), (1, None), (255, 17)):
                    g = _G(list(ds), mag)
                    svc = ref.extraction.PatchExtractionService(
                        ref.config.ExtractionConfig(patch_size=ps, step_size=step,
                                                    target_magnification=tgt),
                        ref.config.OutputConfig(output_root=Path(tempfile.gettempdir()) / "ap_g"))
                    try:
                        lv, (rw, rh), pss, ss, p0 = svc._prepare_geometry(g)
                        out = [int(lv), int(rw), int(rh), int(pss), int(ss), int(p0)]
                    except ValueError as e:
                        out = {"error": str(e)}
                    rows.append({"ds": ds, "mag": mag, "tgt": tgt, "ps": ps, "step": step, "out": out})
    levels = []
    for ds in ds_lists:
        g = _G(list(ds), 20)
        for t in (0.5, 1.0, 1.005, 1.5, 2.0, 3.99, 4.0, 4.009, 4.011, 15.99, 16.0, 16.5, 31.0, 64.0):
            try:
                lv, extra = g.optimal_level(t)
                levels.append({"ds": ds, "target": t, "level": int(lv), "extra": float(extra)})
            except ValueError as e:
                levels.append({"ds": ds, "target": t, "error": str(e)})
    (out_dir / "geometry.json").write_text(json.dumps({"geometry": rows, "levels": levels}))
    print(f"  geometry rows {len(rows)}, level rows {len(levels)}")


def gen_scale(out_dir):
    rng = np.random.default_rng(7)
    arrays = {}
    k = 0
    for (W, H) in ((40000, 40000), (46000, 32914), (99999, 99999), (100000, 100000), (123457, 98765)):
        for (mh, mw) in ((1024, 1024), (733, 1024), (1024, 700)):
            pts = np.stack([rng.integers(0, mw, 2000), rng.integers(0, mh, 2000)], -1)
            pts = np.concatenate([pts, [[0, 0], [mw - 1, mh - 1], [mw - 1, 0], [0, mh - 1]]], 0)
            c = pts.astype(np.int32).reshape(-1, 1, 2)
            sx, sy = W / float(mw), H / float(mh)
            out = ref.contours.scale_contours([c], sx, sy)[0]
            arrays[f"c{k}_in"] = c.reshape(-1, 2)
            arrays[f"c{k}_out"] = out.reshape(-1, 2)
            arrays[f"c{k}_dims"] = np.array([W, H, mw, mh], dtype=np.int64)
            k += 1
    # full sweep of x = 0..1023 for the sizes quoted in SURVEY (f32 truncation differs from f64)
    for W in (99999, 100000, 40000, 46000):
        c = np.stack([np.arange(1024), np.arange(1024)], -1).astype(np.int32).reshape(-1, 1, 2)
        out = ref.contours.scale_contours([c], W / 1024.0, W / 1024.0)[0]
        arrays[f"sweep{W}"] = out.reshape(-1, 2)
    np.savez_compressed(out_dir / "scale_contours.npz", **arrays)
    print(f"  scale_contours vectors: {k} random sets + 4 sweeps")


def gen_config(out_dir):
    import torch
    cfgmod = ref.config
    table = {"device": [], "extraction": [], "features": [], "dtype": [], "registry": [],
             "parse_feature_list": []}
    for dev in ["cpu", "cuda", "CUDA:0", " cuda:12 ", "cuda:", "cuda:x", "gpu", "hip", "cuda:-1", ""]:
        try:
            table["device"].append([dev, cfgmod._validate_device(dev)])
        except ValueError as e:
            table["device"].append([dev, {"error": str(e)}])
    for kw in [dict(patch_size=256, target_magnification=20),
               dict(patch_size=0, target_magnification=20),
               dict(patch_size=256, target_magnification=0),
               dict(patch_size=256, target_magnification=20, step_size=0),
               dict(patch_size=256, target_magnification=20, tissue_threshold=1.5),
               dict(patch_size=256, target_magnification=20, tissue_threshold=-0.1),
               dict(patch_size=256, target_magnification=20, white_threshold=0),
               dict(patch_size=256, target_magnification=20, black_threshold=-1),
               dict(patch_size=256, target_magnification=20, write_batch=0),
               dict(patch_size=256, target_magnification=20, workers=0),
               dict(patch_size=256, target_magnification=20, max_open_slides=0),
               dict(patch_size=256, target_magnification=20, step_size=64, workers=3)]:
        try:
            c = cfgmod.ExtractionConfig(**kw).validated()
            table["extraction"].append([kw, {"step_size": c.step_size,
                                             "max_open_slides": c.max_open_slides}])
        except ValueError as e:
            table["extraction"].append([kw, {"error": str(e)}])
    for kw in [dict(extractors=["vit_b_16"]), dict(extractors=[]),
               dict(extractors=["a"], batch_size=0), dict(extractors=["a"], num_workers=-1),
               dict(extractors=["a"], precision="FLOAT16"), dict(extractors=["a"], precision="fp8"),
               dict(extractors=["a"], device="tpu")]:
        try:
            c = cfgmod.FeatureExtractionConfig(**kw).validated()
            table["features"].append([kw, {"precision": c.precision, "device": c.device}])
        except ValueError as e:
            table["features"].append([kw, {"error": str(e)}])
    for dev in ("cpu", "cuda"):
        for prec in ("float32", "float16", "bfloat16", "weird"):
            dt = ref.feature_embedding.resolve_feature_dtype(torch.device(dev), prec)
            table["dtype"].append([dev, prec, str(dt)])
    reg = ref.registry.PatchFeatureExtractorRegistry()
    reg.register("Foo", lambda: "foo-built")
    reg.register("bar", lambda: "bar-built")
    ev = {"available": reg.available(), "create_FOO": reg.create("FOO")}
    try:
        reg.register("foo", lambda: 1)
    except ValueError as e:
        ev["dup"] = str(e)
    try:
        reg.create("nope")
    except KeyError as e:
        ev["unknown"] = str(e)
    table["registry"] = ev
    import click
    for raw in ["vit_b_16", "vit_b_16, uni_v1", "VIT_B_16  uni_v1", "", " , ", "nope", "vit_b_16 vit_b_16"]:
        try:
            table["parse_feature_list"].append(
                [raw, ref.features.parse_feature_list(raw, choices=["vit_b_16", "uni_v1"])])
        except click.BadParameter as e:
            table["parse_feature_list"].append([raw, {"error": e.message}])
    (out_dir / "config_cases.json").write_text(json.dumps(table, indent=1))
    print("  config cases written")


def gen_extract_batch(out_dir):
    """G1: the reference's PatchFeatureExtractor.extract_batch driving a seeded HF ViTModel."""
    import torch
    from transformers import ViTConfig, ViTModel

    torch.set_num_threads(8)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)

    def preprocess(pil):
        # torchvision ImageClassification(crop=224, resize=256) on a 256x256 PIL image [3P semantics,
        # SURVEY 9.2]: resize no-op, centre crop 16, /255 (true division), (x-mean)/std.
        arr = np.asarray(pil, dtype=np.uint8)[16:240, 16:240, :]
        x = torch.from_numpy(arr.copy()).permute(2, 0, 1).to(torch.float32).div(255)
        return x.sub(mean).div(std)

    arrays = {}
    for tag, layers, ns in (("L2", 2, (0, 1, 5, 32, 33)), ("L12", 12, (5,))):
        torch.manual_seed(0)
        cfg = ViTConfig(hidden_size=768, num_hidden_layers=layers, num_attention_heads=12,
                        intermediate_size=3072, image_size=224, patch_size=16,
                        layer_norm_eps=1e-6, hidden_act="gelu")
        model = ViTModel(cfg, add_pooling_layer=False).eval()
        # make LN / bias / pos-embed non-trivial so that every parameter is exercised
        g = torch.Generator().manual_seed(1)
        with torch.no_grad():
            for n_, p in model.named_parameters():
                if "layernorm" in n_ and n_.endswith("weight"):
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
                elif n_.endswith("bias") or "position_embeddings" in n_ or "cls_token" in n_:
                    p.copy_(0.02 * torch.randn(p.shape, generator=g))

        def loader(device, dtype, _m=model):
            return ref.custom.CustomEncoderComponents(
                model=_m, preprocess=preprocess,
                forward_fn=lambda x, _mm=_m: _mm(pixel_values=x).last_hidden_state[:, 0])

        reg = ref.registry.PatchFeatureExtractorRegistry()
        ref.custom.register_custom_encoder(registry=reg, name=f"hfvit_{tag}", embedding_dim=768,
                                           loader=loader, device=torch.device("cpu"),
                                           dtype=torch.float32, num_workers=0)
        ex = reg.create(f"hfvit_{tag}")
        rng = np.random.default_rng(0)
        for n in ns:
            patches = [rng.integers(0, 256, (256, 256, 3), dtype=np.uint8) for _ in range(n)]
            feats = ex.extract_batch(patches, batch_size=32)
            assert feats.dtype == np.float32 and feats.shape == (n, 768), (feats.dtype, feats.shape)
            arrays[f"{tag}_n{n}_out"] = feats
            arrays[f"{tag}_n{n}_seed"] = np.array([0, n], dtype=np.int64)
            print(f"  extract_batch {tag} n={n}: {feats.shape} |f|={np.abs(feats).mean():.4f}")
        # the first two patches' preprocessed tensors pin the preprocess op order (bit-exact)
        if tag == "L2":
            rng = np.random.default_rng(0)
            p0 = rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)
            from PIL import Image
            arrays["preproc_in"] = p0
            arrays["preproc_out"] = preprocess(Image.fromarray(p0)).numpy()
    np.savez_compressed(out_dir / "extract_batch.npz", **arrays)


def gen_extract_batch_lowp(out_dir):
    """G1b: the reference's own float16 / bfloat16 feature path -- PatchFeatureExtractor converts the whole module with
    ``model.to(dtype)`` (models/patch/base.py:66) and feeds it ``batch.to(dtype)`` (:96-99) -- on the L12 model and the
    five patches of G1.  These outputs are the envelope the build's 16-bit modes are held to: its error against the
    fp32 path must not exceed the reference's own."""
    import copy
    import torch
    from transformers import ViTConfig, ViTModel

    torch.set_num_threads(8)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)

    def preprocess(pil):
        arr = np.asarray(pil, dtype=np.uint8)[16:240, 16:240, :]
        x = torch.from_numpy(arr.copy()).permute(2, 0, 1).to(torch.float32).div(255)
        return x.sub(mean).div(std)

    torch.manual_seed(0)
    cfg = ViTConfig(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                    image_size=224, patch_size=16, layer_norm_eps=1e-6, hidden_act="gelu")
    base = ViTModel(cfg, add_pooling_layer=False).eval()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n_, p in base.named_parameters():
            if "layernorm" in n_ and n_.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif n_.endswith("bias") or "position_embeddings" in n_ or "cls_token" in n_:
                p.copy_(0.02 * torch.randn(p.shape, generator=g))
    arrays = {}
    for tag, dt in (("f16", torch.float16), ("bf16", torch.bfloat16)):
        model = copy.deepcopy(base)

        def loader(device, dtype, _m=model):
            return ref.custom.CustomEncoderComponents(
                model=_m, preprocess=preprocess,
                forward_fn=lambda x, _mm=_m: _mm(pixel_values=x).last_hidden_state[:, 0])

        reg = ref.registry.PatchFeatureExtractorRegistry()
        ref.custom.register_custom_encoder(registry=reg, name=f"hfvit_L12_{tag}", embedding_dim=768, loader=loader,
                                           device=torch.device("cpu"), dtype=dt, num_workers=0)
        ex = reg.create(f"hfvit_L12_{tag}")
        assert next(ex.model.parameters()).dtype == dt
        rng = np.random.default_rng(0)
        patches = [rng.integers(0, 256, (256, 256, 3), dtype=np.uint8) for _ in range(5)]
        feats = ex.extract_batch(patches, batch_size=32)
        assert feats.dtype == np.float32 and feats.shape == (5, 768)
        arrays[f"L12_n5_out_{tag}"] = feats
        print(f"  reference extract_batch L12 n=5 in {tag}: |f|={np.abs(feats).mean():.4f}")
    np.savez_compressed(out_dir / "extract_batch_lowp.npz", **arrays)


STRESS_BLOCKS = (2, 5, 8)
STRESS_CHANNELS = (7, 300, 511)
STRESS_GAIN = 50.0


def gen_extract_batch_stress(out_dir):
    """G1c: trained-like stress for the 16-bit modes, with the envelope taken from the REFERENCE's own 16-bit runs.

    (a) "massive": the seeded depth-12 HF ViT of G1 with the fc2 rows of channels 7 / 300 / 511 scaled x50 in blocks 2, 5
        and 8 -- those channels of the residual stream then carry activations two orders of magnitude above the rest
        from block 2 on (the "massive activations" of trained ViTs), which is where a 16-bit residual stream and a
        LayerNorm folded into the next GEMM lose most.
    (b) "layerscale": a 24-block ViT-L/16 with LayerScale at UNI's init value 1e-5 (models/patch/uni.py:35), as a plain
        torch module with timm's forward order (norm1 -> qkv -> sdpa -> proj -> ls1 -> + ; norm2 -> fc1 -> GELU -> fc2 ->
        ls2 -> +), weights = atlaspatch_amd.encoders.vit.random_canonical_state_dict(ARCHS["uni_v1"], seed=31).
    Both run through the reference's PatchFeatureExtractor.extract_batch in float32, float16 and bfloat16
    (model.to(dtype), models/patch/base.py:66)."""
    import copy
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    from transformers import ViTConfig, ViTModel
    from PIL import Image
    from atlaspatch_amd.encoders.vit import ARCHS, random_canonical_state_dict

    torch.set_num_threads(8)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)

    def crop_preprocess(pil):
        arr = np.asarray(pil, dtype=np.uint8)[16:240, 16:240, :]
        x = torch.from_numpy(arr.copy()).permute(2, 0, 1).to(torch.float32).div(255)
        return x.sub(mean).div(std)

    def uni_preprocess(pil):                     # timm: Resize(224, bicubic) + CenterCrop(224) + ToTensor + Normalize
        arr = np.asarray(pil.resize((224, 224), Image.Resampling.BICUBIC), dtype=np.uint8)
        x = torch.from_numpy(arr.copy()).permute(2, 0, 1).to(torch.float32).div(255)
        return x.sub(mean).div(std)

    def run(tag, base, preprocess, forward, dim, n_patches, arrays):
        for dtag, dt in (("f32", torch.float32), ("f16", torch.float16), ("bf16", torch.bfloat16)):
            model = copy.deepcopy(base)

            def loader(device, dtype, _m=model):
                return ref.custom.CustomEncoderComponents(model=_m, preprocess=preprocess, forward_fn=lambda x, _mm=_m: forward(_mm, x))

            reg = ref.registry.PatchFeatureExtractorRegistry()
            ref.custom.register_custom_encoder(registry=reg, name=f"{tag}_{dtag}", embedding_dim=dim, loader=loader,
                                               device=torch.device("cpu"), dtype=dt, num_workers=0)
            ex = reg.create(f"{tag}_{dtag}")
            assert next(ex.model.parameters()).dtype == dt
            rng = np.random.default_rng(77)
            patches = [rng.integers(0, 256, (256, 256, 3), dtype=np.uint8) for _ in range(n_patches)]
            feats = ex.extract_batch(patches, batch_size=32)
            assert feats.dtype == np.float32 and feats.shape == (n_patches, dim) and np.isfinite(feats).all()
            arrays[f"{tag}_{dtag}"] = feats
            if dtag != "f32":
                err = np.linalg.norm(feats - arrays[f"{tag}_f32"]) / np.linalg.norm(arrays[f"{tag}_f32"])
                print(f"  reference {tag} in {dtag}: own error vs its f32 run {err:.3e}")

    arrays = {}
    # ---- (a)
    torch.manual_seed(0)
    cfg = ViTConfig(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                    image_size=224, patch_size=16, layer_norm_eps=1e-6, hidden_act="gelu")
    base = ViTModel(cfg, add_pooling_layer=False).eval()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n_, p in base.named_parameters():
            if "layernorm" in n_ and n_.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif n_.endswith("bias") or "position_embeddings" in n_ or "cls_token" in n_:
                p.copy_(0.02 * torch.randn(p.shape, generator=g))
        sd = base.state_dict()
        for blk in STRESS_BLOCKS:
            key = next(k for k in sd if k.endswith("weight") and (f"layer.{blk}.output.dense" in k or f"layers.{blk}.mlp.fc2" in k))
            sd[key][list(STRESS_CHANNELS), :] *= STRESS_GAIN
    run("massive", base, crop_preprocess, lambda m, x: m(pixel_values=x).last_hidden_state[:, 0], 768, 5, arrays)

    # ---- (b)
    arch = ARCHS["uni_v1"]
    csd = random_canonical_state_dict(arch, seed=31)

    class Block(nn.Module):
        def __init__(self, d, heads, mlp):
            super().__init__()
            self.heads = heads
            self.norm1, self.norm2 = nn.LayerNorm(d, eps=1e-6), nn.LayerNorm(d, eps=1e-6)
            self.qkv, self.proj = nn.Linear(d, 3 * d), nn.Linear(d, d)
            self.fc1, self.fc2 = nn.Linear(d, mlp), nn.Linear(mlp, d)
            self.ls1, self.ls2 = nn.Parameter(torch.ones(d)), nn.Parameter(torch.ones(d))

        def forward(self, x):
            n, t, d = x.shape
            q, k, v = self.qkv(self.norm1(x)).reshape(n, t, 3, self.heads, d // self.heads).permute(2, 0, 3, 1, 4).unbind(0)
            a = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(n, t, d)
            x = x + self.proj(a) * self.ls1
            return x + self.fc2(F.gelu(self.fc1(self.norm2(x)))) * self.ls2

    class TimmLikeViT(nn.Module):
        def __init__(self):
            super().__init__()
            d = arch["dim"]
            self.patch = nn.Conv2d(3, d, 16, 16)
            self.cls, self.pos = nn.Parameter(torch.zeros(1, 1, d)), nn.Parameter(torch.zeros(1, 197, d))
            self.blocks = nn.ModuleList([Block(d, arch["heads"], arch["mlp_dim"]) for _ in range(arch["depth"])])
            self.norm = nn.LayerNorm(d, eps=1e-6)

        def forward(self, x):
            x = self.patch(x).flatten(2).transpose(1, 2)
            x = torch.cat([self.cls.expand(x.shape[0], -1, -1), x], 1) + self.pos
            for b in self.blocks:
                x = b(x)
            return self.norm(x)[:, 0]

    vit = TimmLikeViT().eval()
    with torch.no_grad():
        vit.patch.weight.copy_(csd["patch_embed.weight"]); vit.patch.bias.copy_(csd["patch_embed.bias"])
        vit.cls.copy_(csd["cls_token"].view(1, 1, -1)); vit.pos.copy_(csd["pos_embed"][None])
        vit.norm.weight.copy_(csd["norm.weight"]); vit.norm.bias.copy_(csd["norm.bias"])
        for i, b in enumerate(vit.blocks):
            p = f"blocks.{i}."
            for mod, name in ((b.norm1, "ln1"), (b.norm2, "ln2"), (b.qkv, "qkv"), (b.proj, "proj"), (b.fc1, "fc1"), (b.fc2, "fc2")):
                mod.weight.copy_(csd[p + name + ".weight"]); mod.bias.copy_(csd[p + name + ".bias"])
            b.ls1.copy_(csd[p + "ls1"]); b.ls2.copy_(csd[p + "ls2"])
    run("layerscale", vit, uni_preprocess, lambda m, x: m(x), 1024, 3, arrays)
    np.savez_compressed(out_dir / "extract_batch_stress.npz", **arrays)


def gen_features_h5(out_dir):
    """G5: features/<name> layout after the reference's own embed_all on a tiny synthetic slide."""
    import torch

    class _Tiny(torch.nn.Module):
        def forward(self, x):
            return x.mean(dim=(2, 3)).repeat(1, 4)     # [B, 12]

    with tempfile.TemporaryDirectory() as tmp:
        mask = ellipse_mask(256, 256, [(128, 128, 60, 50, 1.0)])
        cfg = dict(width=8192, height=8192, mag=20, downsamples=[1.0, 4.0, 16.0], patch_size=256,
                   step_size=None, target_mag=20, tissue_thresh=0.0)
        coords, info, tissue, holes, result, wsi, svc = run_coords_case("tiny", mask, cfg, tmp)

        def preprocess(pil):
            return torch.from_numpy(np.asarray(pil, dtype=np.uint8).copy()).permute(2, 0, 1).float()

        reg = ref.registry.PatchFeatureExtractorRegistry()
        ref.custom.register_custom_encoder(
            registry=reg, name="tiny12", embedding_dim=12,
            loader=lambda d, t: ref.custom.CustomEncoderComponents(model=_Tiny(), preprocess=preprocess),
            device=torch.device("cpu"), dtype=torch.float32)
        fcfg = ref.config.FeatureExtractionConfig(extractors=["tiny12"], batch_size=7, device="cpu",
                                                  num_workers=0)
        fsvc = ref.feature_embedding.PatchFeatureEmbeddingService(svc.cfg, svc.output_cfg, fcfg,
                                                                  registry=reg)

        class _Loader:
            def open(self, slide):
                return wsi

        failures = fsvc.embed_all([result], wsi_loader=_Loader())
        assert not failures, failures
        layout = _ref_harness.describe_h5(result.h5_path)
        f = _ref_harness.read_h5(result.h5_path)
        feats = np.array(f["features"]["tiny12"][:])
        layout["file_attr_values"].pop("creation_date", None)
        layout["file_attr_values"].pop("wsi_path", None)
        out = {"layout": layout, "num_patches": int(result.num_patches),
               "lock_exists_after": os.path.exists(
                   ref.paths.patch_lock_path(result.slide, svc.output_cfg, svc.cfg)),
               "feature_sets": result.metadata.get("feature_sets")}
        (out_dir / "features_h5.json").write_text(json.dumps(out, indent=1, sort_keys=True))
        np.savez_compressed(out_dir / "features_h5.npz", coords=coords, feats=feats,
                            mask_bits=np.packbits(mask > 0.5))
        print(f"  features_h5: N={coords.shape[0]} feats {feats.shape}")


def gen_openslide_props(out_dir):
    """G8: MPP / magnification lookup of the reference's OpenSlide backend (core/wsi/openslide_wsi.py:71-147) on
    fake property dicts.  ``openslide`` is not installed: a stub module supplies the two property-name constants and
    an exception type, and ``_extract_mpp`` / ``_extract_mag`` run unmodified on an object whose ``meta`` is the dict."""
    import importlib
    import types
    stub = types.ModuleType("openslide")
    stub.PROPERTY_NAME_MPP_X = "openslide.mpp-x"
    stub.PROPERTY_NAME_OBJECTIVE_POWER = "openslide.objective-power"
    stub.OpenSlide = object
    stub.OpenSlideError = type("OpenSlideError", (Exception,), {})
    sys.modules["openslide"] = stub
    mod = importlib.import_module("atlas_patch.core.wsi.openslide_wsi")
    cases = [
        {},
        {"openslide.mpp-x": "0.2528"},
        {"openslide.mpp-x": "0.499", "openslide.objective-power": "20"},
        {"openslide.mpp-y": "0.2611", "aperio.MPP": "0.2500"},
        {"openslide.mirax.MPP": "0.23", "aperio.MPP": "0.5"},
        {"aperio.MPP": "0.25210", "aperio.AppMag": "40"},
        {"hamamatsu.XResolution": "0.44"},
        {"openslide.mpp-x": "garbage", "aperio.MPP": "0.3456789"},
        {"openslide.mpp-x": "garbage"},
        {"openslide.comment": "Aperio Image Library v12 |AppMag = 20|MPP = 0.4990|Left = 25"},
        {"tiff.ImageDescription": "scanner X, 0.2456 microns per pixel approx"},
        {"openslide.comment": "microns per pixel: 0.5021; mpp: 0.25"},
        {"openslide.comment": "mpp=.", "tiff.ImageDescription": "Micron per pixel = 1.0"},
        {"tiff.XResolution": "40000", "tiff.ResolutionUnit": "centimeter"},
        {"tiff.XResolution": "101600.5", "tiff.ResolutionUnit": "Inch"},
        {"tiff.XResolution": "abc", "tiff.ResolutionUnit": "inch", "aperio.AppMag": "20"},
        {"tiff.XResolution": "40000", "tiff.ResolutionUnit": "furlong", "openslide.objective-power": "40"},
        {"aperio.AppMag": "40"},
        {"aperio.AppMag": "0", "openslide.objective-power": "20"},
        {"openslide.objective-power": "20.0"},
        {"hamamatsu.SourceLens": "40"},
        {"aperio.AppMag": "x", "hamamatsu.SourceLens": "20"},
        {"openslide.objective-power": "n/a", "openslide.mpp-x": "0.25"},
        {"openslide.objective-power": "", "aperio.MPP": "0.17"},
        {"openslide.mpp-x": "0.12"},
        {"openslide.mpp-x": "1.9"},
        {"openslide.mpp-x": "3.0"},
        {"mirax.DICOM.PIXEL_SPACING": "0.23"},
    ]
    rows = []
    for meta in cases:
        obj = mod.OpenSlideWSI.__new__(mod.OpenSlideWSI)
        obj._oslide = object()
        obj.meta = dict(meta)
        mpp = obj._extract_mpp()
        obj.mpp = mpp
        rows.append({"meta": meta, "mpp": mpp, "mag": obj._extract_mag()})
    with open(out_dir / "openslide_props.json", "w") as fh:
        json.dump({"keys": {"mpp": list(mod.OpenSlideWSI._MPP_KEYS), "text": list(mod.OpenSlideWSI._MPP_TEXT_KEYS),
                            "mag": list(mod.OpenSlideWSI._MAG_KEYS)}, "cases": rows}, fh, indent=1)
    print(f"  openslide_props: {len(rows)} cases")


if __name__ == "__main__":
    out_dir = HERE
    which = set(sys.argv[1:]) or {"geometry", "scale", "coords", "config", "extract", "features", "openslide"}
    if "openslide" in which:
        gen_openslide_props(out_dir)
    if "geometry" in which:
        gen_geometry(out_dir)
    if "scale" in which:
        gen_scale(out_dir)
    if "config" in which:
        gen_config(out_dir)
    if "coords" in which:
        gen_coords(out_dir)
    if "features" in which:
        gen_features_h5(out_dir)
    if "extract" in which:
        gen_extract_batch(out_dir)
    if "lowp" in which:
        gen_extract_batch_lowp(out_dir)
    if "stress" in which:
        gen_extract_batch_stress(out_dir)
    print("done")
