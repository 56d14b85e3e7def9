"""Pin kit for the THIRD-PARTY arithmetic of the path ("parity unpinned").

The reference's hot path takes several results from packages that are absent from this image (no network):

  opencv-python  cv2.findContours / contourArea / boundingRect / pointPolygonTest (utils/contours.py:37,59,91,104,
                 services/extraction.py:79,94), cv2.resize (core/wsi/iwsi.py:305-321, services/feature_embedding.py:94-95),
                 cv2.cvtColor (utils/image.py:14,33)
  torchvision    vit_b_16 / vit_l_16 + weights.transforms()            (models/patch/vit.py:9-15, base.py:126-180)
  timm           the UNI ViT-L/16 module + its resolved transform       (models/patch/uni.py:32-48)
  sam2           SAM2.1 Hiera-T image predictor                         (services/segmentation.py:62-69,120-140)
  conch          conch_ViT-B-16 visual tower, encode_image              (models/patch/conch.py:37-52)

The build restates all of them (oracle/cv2_restated.py, oracle/cv2_resize.py, oracle/vit_oracle.py, oracle/sam2_oracle.py)
and is bit-exact / within tolerance against those restatements -- but a restatement nobody could run against the real
package is a reading, not a pin.  This script turns "unpinned" into fixtures with ONE command on any machine where the
packages import:

    python tests/golden/gen_golden_3p.py                 # every section whose package imports; the rest are reported
    python tests/golden/gen_golden_3p.py cv2 torchvision # chosen sections

It writes tests/golden/third_party/<section>.npz + <section>.json (inputs are regenerated from seeds by the tests, so
the files hold outputs, key lists and package versions only -- no package source).  tests/test_third_party_pins.py
consumes every fixture that is present (restatement AND, with -m gpu, the device path) and skips, with the reason, the
ones that are not.

`--out DIR` writes elsewhere; `--shim` runs the cv2 section with the oracle's own restatement standing in for cv2 -- a
self-test of the kit's round trip (used by the CPU suite; its output is NOT a pin and is never committed).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))

OUT_DEFAULT = HERE / "third_party"


# ============================================================================= shared, seed-defined inputs
def coords_case_masks():
    """The G4 masks (tests/golden/coords_cases.npz) as uint8 {0, 255} images, in file order."""
    arrays = np.load(HERE / "coords_cases.npz")
    with open(HERE / "coords_cases.json") as fh:
        meta = json.load(fh)
    out = {}
    for name in meta:
        shape = tuple(int(v) for v in arrays[f"{name}__mask_shape"])
        bits = np.unpackbits(arrays[f"{name}__mask_bits"])[: shape[0] * shape[1]]
        out[name] = (bits.reshape(shape).astype(np.uint8) * 255)
    return out


def random_masks(count: int = 16):
    """Seeded blob / noise masks (NumPy only, so every machine draws the same ones)."""
    out = {}
    for seed in range(count):
        rng = np.random.default_rng(7000 + seed)
        h, w = int(rng.integers(40, 200)), int(rng.integers(40, 200))
        if seed % 3 == 0:                                  # salt-and-pepper: many one-pixel contours and holes
            m = rng.random((h, w)) < rng.uniform(0.3, 0.7)
        else:                                              # blocky blobs with holes: coarse noise, up-sampled, eroded by a second layer
            cell = int(rng.integers(3, 12))
            coarse = rng.random((h // cell + 2, w // cell + 2)) < rng.uniform(0.35, 0.65)
            m = np.kron(coarse, np.ones((cell, cell), bool))[:h, :w]
            m &= ~(rng.random((h, w)) < 0.02)
            if seed % 3 == 2:
                m[[0, -1], :] |= rng.random((2, w)) < 0.5          # tissue touching the border
        out[f"rand{seed:02d}"] = m.astype(np.uint8) * 255
    return out


def pip_probe_points(contour: np.ndarray, seed: int, n_random: int = 48) -> np.ndarray:
    """Integer probe points for pointPolygonTest on one contour [n, 1, 2]: its first vertices (on-vertex case), midpoints
    of its first edges (on-edge / near-edge), and seeded points in the bounding box grown by 2."""
    pts = contour.reshape(-1, 2).astype(np.int64)
    rng = np.random.default_rng(seed)
    x0, y0 = pts.min(0) - 2
    x1, y1 = pts.max(0) + 3
    rand = np.stack([rng.integers(x0, x1, n_random), rng.integers(y0, y1, n_random)], 1)
    head = pts[:12]
    mid = (pts[:12] + np.roll(pts, -1, 0)[:12]) // 2
    return np.concatenate([head, mid, rand], 0).astype(np.int32)


RESIZE_CASES = [
    # (h, w) -> (oh, ow), cv2 interpolation constant -- the shapes tests/test_gpu_ops.py::_CV2_CASES runs on the device
    ((512, 512), (256, 256), 1), ((1024, 1024), (256, 256), 1), ((300, 300), (256, 256), 1), ((180, 200), (256, 256), 1),
    ((511, 513), (256, 256), 1), ((256, 256), (256, 256), 1), ((768, 768), (256, 256), 3), ((96, 64), (32, 16), 3),
    ((411, 300), (100, 128), 3), ((733, 1024), (699, 500), 3), ((100, 100), (77, 33), 3), ((100, 80), (200, 256), 3),
    ((40, 60), (80, 30), 3), ((100, 80), (200, 256), 2), ((300, 300), (256, 256), 2), ((60, 75), (163, 201), 2),
    ((1562, 1562), (781, 781), 3),
]


def resize_input(case_index: int) -> np.ndarray:
    (h, w), _, _ = RESIZE_CASES[case_index]
    return np.random.default_rng(8000 + case_index).integers(0, 256, (h, w, 3), dtype=np.uint8)


def color_input() -> np.ndarray:
    """A 64 x 64 tile that covers grey ramps, saturated primaries and random colours (utils/image.py filters)."""
    rng = np.random.default_rng(8100)
    t = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    t[:8] = np.arange(64, dtype=np.uint8)[None, :, None] * 4
    t[8:12] = rng.integers(195, 256, (4, 64, 1), dtype=np.uint8)             # near-white greys
    t[12:14, :, :] = [[255, 0, 0]]
    t[14:16, :, :] = [[0, 255, 1]]
    return t


def seeded_tiles(n: int, seed: int = 8200, side: int = 256):
    rng = np.random.default_rng(seed)
    return [rng.integers(0, 256, (side, side, 3), dtype=np.uint8) for _ in range(n)]


# ============================================================================= cv2
def section_cv2(out: Path, cv2) -> dict:
    arrays, meta = {}, {"package": "opencv-python", "version": getattr(cv2, "__version__", "?"), "cases": {}}
    masks = {**coords_case_masks(), **random_masks()}
    for name, img in masks.items():
        res = cv2.findContours(img.copy(), cv2.RETR_CCOMP, cv2.CHAIN_APPROX_NONE)
        contours, hierarchy = res[-2], res[-1]                      # OpenCV 3 returns (image, contours, hierarchy)
        contours = [np.asarray(c, np.int32).reshape(-1, 1, 2) for c in contours]
        hier = np.zeros((0, 4), np.int32) if hierarchy is None else np.asarray(hierarchy, np.int32).reshape(-1, 4)
        arrays[f"{name}__lens"] = np.array([len(c) for c in contours], np.int64)
        arrays[f"{name}__pts"] = (np.concatenate([c.reshape(-1, 2) for c in contours], 0) if contours
                                  else np.zeros((0, 2), np.int32)).astype(np.int32)
        arrays[f"{name}__hier"] = hier
        arrays[f"{name}__area"] = np.array([cv2.contourArea(c) for c in contours], np.float64)
        arrays[f"{name}__rect"] = np.array([cv2.boundingRect(c) for c in contours], np.int64).reshape(-1, 4)
        pip = []
        for k, c in enumerate(contours[:40]):
            pts = pip_probe_points(c, seed=9000 + k)
            pip.append(np.array([cv2.pointPolygonTest(c, (int(x), int(y)), False) for x, y in pts], np.int8))
        arrays[f"{name}__pip"] = np.concatenate(pip) if pip else np.zeros((0,), np.int8)
        meta["cases"][name] = {"shape": list(img.shape), "contours": len(contours)}
    np.savez_compressed(out / "cv2_primitives.npz", **arrays)

    rz, rmeta = {}, []
    for i, ((h, w), (oh, ow), interp) in enumerate(RESIZE_CASES):
        rz[f"case{i:02d}"] = cv2.resize(resize_input(i), (ow, oh), interpolation=interp)
        rmeta.append({"in_hw": [h, w], "out_hw": [oh, ow], "interpolation": interp})
    tile = color_input()
    rz["gray"] = cv2.cvtColor(tile, cv2.COLOR_RGB2GRAY)
    rz["hsv"] = cv2.cvtColor(tile, cv2.COLOR_RGB2HSV)
    np.savez_compressed(out / "cv2_resize.npz", **rz)
    meta["resize_cases"] = rmeta
    with open(out / "cv2.json", "w") as fh:
        json.dump(meta, fh, indent=1)
    return {"files": ["cv2_primitives.npz", "cv2_resize.npz", "cv2.json"], "masks": len(masks), "resize_cases": len(RESIZE_CASES)}


def oracle_as_cv2():
    """The oracle's own restatement behind cv2's names (kit self-test only; NOT a pin)."""
    from oracle import cv2_resize as R
    from oracle import cv2_restated as P
    m = types.ModuleType("cv2")
    m.__version__ = "oracle-shim (not OpenCV)"
    for k in ("RETR_CCOMP", "CHAIN_APPROX_NONE", "findContours", "contourArea", "boundingRect", "pointPolygonTest"):
        setattr(m, k, getattr(P, k))
    m.COLOR_RGB2GRAY, m.COLOR_RGB2HSV = 7, 41

    def cvt(img, code):
        if code == m.COLOR_RGB2GRAY:
            return P.cvtColor_RGB2GRAY(img)
        s, v = P.cvtColor_RGB2HSV_sv(img)
        return np.stack([np.zeros_like(s), s, v], -1).astype(np.uint8)       # the filters read S and V only
    m.cvtColor = cvt
    m.resize = lambda src, dsize, interpolation=1: R.resize(src, dsize, interpolation)
    return m


# ============================================================================= torchvision
def canonical_to_torchvision(sd: dict, depth: int) -> dict:
    out = {"conv_proj.weight": sd["patch_embed.weight"], "conv_proj.bias": sd["patch_embed.bias"],
           "class_token": sd["cls_token"].reshape(1, 1, -1), "encoder.pos_embedding": sd["pos_embed"][None],
           "encoder.ln.weight": sd["norm.weight"], "encoder.ln.bias": sd["norm.bias"]}
    for i in range(depth):
        p, b = f"encoder.layers.encoder_layer_{i}.", f"blocks.{i}."
        out.update({p + "ln_1.weight": sd[b + "ln1.weight"], p + "ln_1.bias": sd[b + "ln1.bias"],
                    p + "self_attention.in_proj_weight": sd[b + "qkv.weight"], p + "self_attention.in_proj_bias": sd[b + "qkv.bias"],
                    p + "self_attention.out_proj.weight": sd[b + "proj.weight"], p + "self_attention.out_proj.bias": sd[b + "proj.bias"],
                    p + "ln_2.weight": sd[b + "ln2.weight"], p + "ln_2.bias": sd[b + "ln2.bias"],
                    p + "mlp.0.weight": sd[b + "fc1.weight"], p + "mlp.0.bias": sd[b + "fc1.bias"],
                    p + "mlp.3.weight": sd[b + "fc2.weight"], p + "mlp.3.bias": sd[b + "fc2.bias"]})
    return out


def section_torchvision(out: Path, torchvision) -> dict:
    import torch
    from PIL import Image
    from torchvision.models import ViT_B_16_Weights, ViT_L_16_Weights, vit_b_16, vit_l_16
    from torchvision.models.vision_transformer import VisionTransformer
    from atlaspatch_amd.encoders.vit import ARCHS, random_canonical_state_dict
    meta = {"package": "torchvision", "version": torchvision.__version__, "torch": torch.__version__, "models": {}}
    arrays = {}
    for name, ctor, weights in (("vit_b_16", vit_b_16, ViT_B_16_Weights.IMAGENET1K_V1), ("vit_l_16", vit_l_16, ViT_L_16_Weights.IMAGENET1K_V1)):
        model = ctor(weights=None)
        t = weights.transforms()
        meta["models"][name] = {
            "state_dict": {k: list(v.shape) for k, v in model.state_dict().items()},
            "transforms": {"class": type(t).__name__, "crop_size": list(t.crop_size), "resize_size": list(t.resize_size),
                           "mean": list(t.mean), "std": list(t.std), "interpolation": str(t.interpolation),
                           "antialias": getattr(t, "antialias", None)}}
        # forward on seeded weights, as models/patch/base.py:148-180 builds it: heads -> Identity, preprocess = weights.transforms()
        arch = dict(ARCHS[name]); arch["depth"] = 2
        sd = random_canonical_state_dict(arch, seed=5)
        small = VisionTransformer(image_size=224, patch_size=16, num_layers=2, num_heads=arch["heads"],
                                  hidden_dim=arch["dim"], mlp_dim=arch["mlp_dim"])
        small.heads = torch.nn.Identity()
        missing = small.load_state_dict({k: v.clone() for k, v in canonical_to_torchvision(sd, 2).items()}, strict=True)
        small.eval()
        tiles = seeded_tiles(4)
        with torch.inference_mode():
            x = torch.stack([t(Image.fromarray(p)) for p in tiles], 0)
            arrays[f"{name}__L2_seed5_input"] = x[:1].numpy().astype(np.float32)          # the transform's output for tile 0
            arrays[f"{name}__L2_seed5_out"] = small(x).numpy().astype(np.float32)
        del missing
    np.savez_compressed(out / "torchvision_vit.npz", **arrays)
    with open(out / "torchvision_vit.json", "w") as fh:
        json.dump(meta, fh, indent=1)
    return {"files": ["torchvision_vit.npz", "torchvision_vit.json"]}


# ============================================================================= timm (UNI)
def canonical_to_timm(sd: dict, depth: int, layer_scale: bool) -> dict:
    out = {"patch_embed.proj.weight": sd["patch_embed.weight"], "patch_embed.proj.bias": sd["patch_embed.bias"],
           "cls_token": sd["cls_token"].reshape(1, 1, -1), "pos_embed": sd["pos_embed"][None],
           "norm.weight": sd["norm.weight"], "norm.bias": sd["norm.bias"]}
    for i in range(depth):
        b = f"blocks.{i}."
        out.update({b + "norm1.weight": sd[b + "ln1.weight"], b + "norm1.bias": sd[b + "ln1.bias"],
                    b + "attn.qkv.weight": sd[b + "qkv.weight"], b + "attn.qkv.bias": sd[b + "qkv.bias"],
                    b + "attn.proj.weight": sd[b + "proj.weight"], b + "attn.proj.bias": sd[b + "proj.bias"],
                    b + "norm2.weight": sd[b + "ln2.weight"], b + "norm2.bias": sd[b + "ln2.bias"],
                    b + "mlp.fc1.weight": sd[b + "fc1.weight"], b + "mlp.fc1.bias": sd[b + "fc1.bias"],
                    b + "mlp.fc2.weight": sd[b + "fc2.weight"], b + "mlp.fc2.bias": sd[b + "fc2.bias"]})
        if layer_scale:
            out.update({b + "ls1.gamma": sd[b + "ls1"], b + "ls2.gamma": sd[b + "ls2"]})
    return out


def section_timm(out: Path, timm) -> dict:
    import torch
    from PIL import Image
    from timm.data import resolve_data_config
    from timm.data.transforms_factory import create_transform
    from atlaspatch_amd.encoders.vit import ARCHS, random_canonical_state_dict
    meta = {"package": "timm", "version": timm.__version__, "torch": torch.__version__}
    # the architecture hf-hub:MahmoodLab/uni resolves to (vit_large_patch16_224) with the reference's overrides (uni.py:32-38)
    full = timm.create_model("vit_large_patch16_224", pretrained=False, init_values=1e-5, dynamic_img_size=True, num_classes=0)
    meta["state_dict"] = {k: list(v.shape) for k, v in full.state_dict().items()}
    cfg = resolve_data_config(full.pretrained_cfg, model=full)
    tfm = create_transform(**cfg)
    meta["data_config"] = {k: (list(v) if isinstance(v, (tuple, list)) else v) for k, v in cfg.items()}
    meta["transform_repr"] = repr(tfm)
    meta["note"] = ("data_config is timm's default for vit_large_patch16_224; the hub config of MahmoodLab/uni overrides mean / std / "
                    "interpolation when the checkpoint is reachable -- rerun with HF access to pin that as well")
    arch = dict(ARCHS["uni_v1"]); arch["depth"] = 2
    sd = random_canonical_state_dict(arch, seed=6)
    g = torch.Generator().manual_seed(66)
    for i in range(2):                                        # LayerScale large enough for both branches to matter
        sd[f"blocks.{i}.ls1"] = torch.rand(1024, generator=g) * 0.5 + 0.2
        sd[f"blocks.{i}.ls2"] = torch.rand(1024, generator=g) * 0.5 + 0.2
    small = timm.create_model("vit_large_patch16_224", pretrained=False, init_values=1e-5, dynamic_img_size=True,
                              num_classes=0, depth=2)
    small.load_state_dict({k: v.clone() for k, v in canonical_to_timm(sd, 2, True).items()}, strict=True)
    small.eval()
    tiles = seeded_tiles(4)
    with torch.inference_mode():
        x = torch.stack([tfm(Image.fromarray(p)) for p in tiles], 0)
        arrays = {"uni_v1__L2_seed6_input": x[:1].numpy().astype(np.float32),
                  "uni_v1__L2_seed6_out": small(x).numpy().astype(np.float32)}
    np.savez_compressed(out / "timm_uni.npz", **arrays)
    with open(out / "timm_uni.json", "w") as fh:
        json.dump(meta, fh, indent=1)
    return {"files": ["timm_uni.npz", "timm_uni.json"]}


# ============================================================================= sam2
def section_sam2(out: Path, sam2) -> dict:
    import torch
    from hydra.utils import instantiate
    from omegaconf import OmegaConf
    from sam2.sam2_image_predictor import SAM2ImagePredictor
    from atlaspatch_amd.services.segmentation import random_sam2_state_dict
    # the reference's own yaml when it is on this machine, else the build's restated copy (same keys)
    cands = [Path("/root/reference/atlas_patch/configs/sam2.1_hiera_t.yaml"), ROOT / "atlaspatch_amd/configs/sam2.1_hiera_t.yaml"]
    try:
        import atlas_patch
        cands.insert(0, Path(atlas_patch.__file__).parent / "configs" / "sam2.1_hiera_t.yaml")
    except Exception:  # noqa: BLE001
        pass
    cfg_path = next(p for p in cands if p.exists())
    conf = OmegaConf.load(str(cfg_path))
    model = instantiate(conf.get("model", conf))                 # services/segmentation.py:62-69
    meta = {"package": "sam2", "version": getattr(sam2, "__version__", "git"), "torch": torch.__version__, "config": str(cfg_path),
            "state_dict": {k: list(v.shape) for k, v in model.state_dict().items()}}
    sd = random_sam2_state_dict(0)
    own = model.state_dict()
    used = {k: v for k, v in sd.items() if k in own}
    meta["seeded_keys_loaded"] = sorted(used)
    meta["seeded_keys_not_in_package"] = sorted(set(sd) - set(own))
    own.update({k: v.to(own[k].dtype).reshape(own[k].shape) for k, v in used.items()})
    model.load_state_dict(own, strict=True)
    model.eval()
    predictor = SAM2ImagePredictor(model, mask_threshold=0.0)
    img = np.random.default_rng(8300).integers(0, 256, (1024, 1024, 3), dtype=np.uint8)
    with torch.inference_mode():
        predictor.set_image(img)                                  # segmentation.py:127-136
        box = np.array([0, 0, 1024, 1024], dtype=np.float32)
        masks, scores, logits = predictor.predict(point_coords=None, point_labels=None, box=box[None, :],
                                                  multimask_output=False, return_logits=True)
        emb = predictor._features["image_embed"].float().cpu().numpy()
    arrays = {"low_res_logits": np.asarray(logits, np.float32).reshape(256, 256),
              "mask_logits_1024_f16": np.asarray(masks, np.float32).reshape(1024, 1024).astype(np.float16),
              "image_embed_sample": emb.reshape(256, 64, 64)[::16].astype(np.float32), "scores": np.asarray(scores, np.float32)}
    np.savez_compressed(out / "sam2_hiera_t.npz", **arrays)
    with open(out / "sam2_hiera_t.json", "w") as fh:
        json.dump(meta, fh, indent=1)
    return {"files": ["sam2_hiera_t.npz", "sam2_hiera_t.json"]}


# ============================================================================= conch
def section_conch(out: Path, conch) -> dict:
    import torch
    from PIL import Image
    import conch.open_clip_custom as occ
    from atlaspatch_amd.encoders.vit import ARCHS, random_attn_pool, random_canonical_state_dict
    model, preprocess = occ.create_model_from_pretrained("conch_ViT-B-16", checkpoint_path=None)   # architecture + transform only
    meta = {"package": "conch", "version": getattr(conch, "__version__", "?"), "torch": torch.__version__,
            "visual_state_dict": {k: list(v.shape) for k, v in model.state_dict().items() if k.startswith("visual.")},
            "preprocess_repr": repr(preprocess)}
    arch = ARCHS["conch_v1"]
    trunk = random_canonical_state_dict({k: v for k, v in arch.items() if not k.startswith("pool")}, seed=9)
    pool = random_attn_pool(arch, seed=9)
    own = model.state_dict()
    mapped = {"visual.trunk." + k: v for k, v in canonical_to_timm(trunk, arch["depth"], False).items()}
    mapped.update({"visual.attn_pool_contrast." + k: v for k, v in pool.items() if not k.startswith("ln_out")})
    mapped["visual.ln_contrast.weight"], mapped["visual.ln_contrast.bias"] = pool["ln_out.weight"], pool["ln_out.bias"]
    meta["seeded_keys_not_in_package"] = sorted(set(mapped) - set(own))
    own.update({k: v.to(own[k].dtype).reshape(own[k].shape) for k, v in mapped.items() if k in own})
    model.load_state_dict(own, strict=True)
    model.eval()
    tiles = seeded_tiles(2)
    with torch.inference_mode():
        x = torch.stack([preprocess(Image.fromarray(p)) for p in tiles], 0)
        feats = model.encode_image(x, proj_contrast=False, normalize=False)       # models/patch/conch.py:52
    np.savez_compressed(out / "conch_v1.npz", conch_v1__seed9_input=x[:1].numpy().astype(np.float32),
                        conch_v1__seed9_out=feats.float().numpy())
    with open(out / "conch_v1.json", "w") as fh:
        json.dump(meta, fh, indent=1)
    return {"files": ["conch_v1.npz", "conch_v1.json"]}


SECTIONS = {"cv2": ("cv2", section_cv2), "torchvision": ("torchvision", section_torchvision), "timm": ("timm", section_timm),
            "sam2": ("sam2", section_sam2), "conch": ("conch", section_conch)}


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("sections", nargs="*", choices=[[]] + sorted(SECTIONS), default=[])
    ap.add_argument("--out", default=str(OUT_DEFAULT))
    ap.add_argument("--shim", action="store_true", help="cv2 section with the oracle standing in for cv2 (kit self-test, not a pin)")
    args = ap.parse_args(argv)
    out = Path(args.out)
    out.mkdir(parents=True, exist_ok=True)
    report = {}
    for name in (args.sections or sorted(SECTIONS)):
        module, fn = SECTIONS[name]
        if args.shim:
            if name != "cv2":
                continue
            pkg = oracle_as_cv2()
        else:
            try:
                pkg = __import__(module)
            except Exception as exc:  # noqa: BLE001 -- absent package: report and go on
                report[name] = {"skipped": f"import {module} failed: {type(exc).__name__}: {exc}"}
                continue
        try:
            report[name] = fn(out, pkg)
        except Exception as exc:  # noqa: BLE001
            report[name] = {"failed": f"{type(exc).__name__}: {exc}"}
    print(json.dumps(report, indent=1))
    return 0 if not any("failed" in v for v in report.values()) else 1


if __name__ == "__main__":
    raise SystemExit(main())
