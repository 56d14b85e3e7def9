"""Consumers of the third-party pin kit (tests/golden/gen_golden_3p.py).

Every test reads a fixture that only a machine WITH the package can write (opencv-python, torchvision, timm, sam2, conch
are absent from this image) and compares the oracle restatement -- and, under -m gpu, the device path -- with it.  A
fixture that is not there skips its tests with the reason; `python tests/golden/gen_golden_3p.py` on a machine with the
packages turns the skips into pins.  The kit itself is exercised here with the oracle standing in for cv2
(`--shim`), so the generator / consumer round trip is known to work before anyone runs it for real.
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
PINS = os.path.join(GOLDEN, "third_party")
sys.path.insert(0, GOLDEN)

import gen_golden_3p as kit  # noqa: E402


def _need(*names, root=PINS):
    missing = [n for n in names if not os.path.exists(os.path.join(root, n))]
    if missing:
        pytest.skip(f"third-party fixture(s) {missing} not generated: run `python tests/golden/gen_golden_3p.py` where the "
                    f"package imports (absent from this image)")
    return [os.path.join(root, n) for n in names]


def _split(pts, lens):
    out, off = [], 0
    for n in lens:
        out.append(np.asarray(pts[off:off + n], np.int32).reshape(-1, 1, 2))
        off += int(n)
    return out


# ----------------------------------------------------------------------------- cv2: the five primitives + resize + cvtColor
def check_cv2_primitives_against_oracle(root):
    from oracle import cv2_restated as P
    path, = _need("cv2_primitives.npz", root=root)
    fx = np.load(path)
    masks = {**kit.coords_case_masks(), **kit.random_masks()}
    checked = 0
    for name, img in masks.items():
        contours, hierarchy = P.findContours(img.copy(), P.RETR_CCOMP, P.CHAIN_APPROX_NONE)
        want = _split(fx[f"{name}__pts"], fx[f"{name}__lens"])
        assert len(contours) == len(want), (name, len(contours), len(want))
        for k, (a, b) in enumerate(zip(contours, want)):
            assert np.array_equal(np.asarray(a).reshape(-1, 1, 2), b), (name, k, "findContours order / start point / direction")
        hier = np.zeros((0, 4), np.int32) if hierarchy is None else np.asarray(hierarchy).reshape(-1, 4)
        assert np.array_equal(hier, fx[f"{name}__hier"]), (name, "hierarchy [next, prev, child, parent]")
        assert np.array_equal(np.array([P.contourArea(c) for c in want], np.float64), fx[f"{name}__area"]), (name, "contourArea")
        assert np.array_equal(np.array([P.boundingRect(c) for c in want], np.int64).reshape(-1, 4), fx[f"{name}__rect"]), name
        pip = []
        for k, c in enumerate(want[:40]):
            pts = kit.pip_probe_points(c, seed=9000 + k)
            pip.append(np.array([P.pointPolygonTest(c, (int(x), int(y)), False) for x, y in pts], np.int8))
        got = np.concatenate(pip) if pip else np.zeros((0,), np.int8)
        assert np.array_equal(got, fx[f"{name}__pip"]), (name, "pointPolygonTest", int((got != fx[f"{name}__pip"]).sum()))
        checked += 1
    return checked


def check_cv2_resize_against_oracle(root):
    from oracle import cv2_resize as R
    from oracle import cv2_restated as P
    path, = _need("cv2_resize.npz", root=root)
    fx = np.load(path)
    for i, ((h, w), (oh, ow), interp) in enumerate(kit.RESIZE_CASES):
        want = fx[f"case{i:02d}"]
        got = R.resize(kit.resize_input(i), (ow, oh), interp)
        if interp == 2 and not np.array_equal(got, want):                 # builds without the 8-lane float vertical pass
            got = R.resize(kit.resize_input(i), (ow, oh), interp, cubic_vertical="scalar")
        assert np.array_equal(got, want), (i, (h, w), (oh, ow), interp, int((got != want).sum()))
    tile = kit.color_input()
    assert np.array_equal(P.cvtColor_RGB2GRAY(tile), fx["gray"])
    s, v = P.cvtColor_RGB2HSV_sv(tile)
    assert np.array_equal(s, fx["hsv"][..., 1]) and np.array_equal(v, fx["hsv"][..., 2])
    return len(kit.RESIZE_CASES)


def test_cv2_primitives_pin():
    """findContours(RETR_CCOMP, CHAIN_APPROX_NONE) / contourArea / boundingRect / pointPolygonTest of oracle/cv2_restated.py
    == OpenCV's on the 17 G4 masks + 16 seeded masks (utils/contours.py:37,59,91,104; services/extraction.py:79,94)."""
    assert check_cv2_primitives_against_oracle(PINS) == 33


def test_cv2_resize_and_cvtcolor_pin():
    """oracle/cv2_resize.py == cv2.resize on the 17 shapes / modes of the device tests; RGB2GRAY / RGB2HSV (utils/image.py)."""
    assert check_cv2_resize_against_oracle(PINS) == 17


def test_pin_kit_round_trip_with_the_oracle_standing_in_for_cv2(tmp_path):
    """The generator and the consumers agree on names, seeds, shapes and dtypes: run the cv2 section with the oracle as
    `cv2` into a scratch directory and consume it.  (Proves the kit works; pins nothing -- the output is not committed.)"""
    assert kit.main(["cv2", "--shim", "--out", str(tmp_path)]) == 0
    meta = json.load(open(tmp_path / "cv2.json"))
    assert meta["version"].startswith("oracle-shim") and len(meta["cases"]) == 33 and len(meta["resize_cases"]) == 17
    assert check_cv2_primitives_against_oracle(str(tmp_path)) == 33
    assert check_cv2_resize_against_oracle(str(tmp_path)) == 17
    fx = np.load(tmp_path / "cv2_primitives.npz")
    assert sum(int(fx[k].shape[0]) for k in fx.files if k.endswith("__pip")) > 20000          # a real number of PIP probes
    assert any(fx[k].min() < 0 < fx[k].max() and (fx[k] == 0).any() for k in fx.files if k.endswith("__pip"))


def test_pin_kit_reports_absent_packages_instead_of_failing(tmp_path, capsys):
    assert kit.main(["torchvision", "timm", "sam2", "conch", "--out", str(tmp_path)]) == 0
    report = json.loads(capsys.readouterr().out)
    for name in ("torchvision", "timm", "sam2", "conch"):
        assert "skipped" in report[name] or "files" in report[name]


def _filter_like_the_reference(contours, hier, areas, shape, thresh=0.0, a_h=16, max_holes=10):
    """utils/contours.py:81-114 on fixture data (cv2's contours / hierarchy / areas)."""
    min_area = thresh * float(shape[0] * shape[1])
    tissue_idx, holes = [], {}
    for i in range(len(contours)):
        parent = int(hier[i][3])
        if parent == -1:
            if areas[i] >= min_area:
                tissue_idx.append(i)
        elif areas[i] >= float(a_h):
            holes.setdefault(parent, []).append(i)
    flat = [h for hs in holes.values() for h in hs]
    if max_holes > 0 and len(flat) > max_holes:
        keep = set(sorted(flat, key=lambda i: areas[i], reverse=True)[:max_holes])
        holes = {p: [h for h in hs if h in keep] for p, hs in holes.items()}
    return [contours[i] for i in tissue_idx], [[contours[h] for h in holes.get(i, [])] for i in tissue_idx]


@pytest.mark.gpu
def test_device_contours_equal_opencv_fixture():
    """ap_contours_from_mask (GPU threshold + host C++ border following + the reference's filters) against OpenCV's own
    contours on every fixture mask, bit for bit."""
    from atlaspatch_amd.utils.contours import mask_to_contours
    path, = _need("cv2_primitives.npz")
    fx = np.load(path)
    for name, img in {**kit.coords_case_masks(), **kit.random_masks()}.items():
        contours = _split(fx[f"{name}__pts"], fx[f"{name}__lens"])
        want_t, want_h = _filter_like_the_reference(contours, fx[f"{name}__hier"], fx[f"{name}__area"], img.shape)
        got_t, got_h = mask_to_contours((img > 0).astype(np.float32), tissue_area_thresh=0.0)
        assert len(got_t) == len(want_t), name
        for a, b in zip(got_t, want_t):
            assert np.array_equal(a, b), name
        for ga, wa in zip(got_h, want_h):
            assert len(ga) == len(wa) and all(np.array_equal(a, b) for a, b in zip(ga, wa)), name


@pytest.mark.gpu
def test_device_cv2_resize_equals_opencv_fixture():
    import torch
    from atlaspatch_amd.utils.resample import cv2_resize_device
    path, = _need("cv2_resize.npz")
    fx = np.load(path)
    dev = torch.device("cuda:0")
    for i, ((h, w), (oh, ow), interp) in enumerate(kit.RESIZE_CASES):
        src = torch.from_numpy(kit.resize_input(i)[None]).to(dev)
        got = cv2_resize_device(src, (ow, oh), interp).cpu().numpy()[0]
        if interp == 2 and not np.array_equal(got, fx[f"case{i:02d}"]):
            got = cv2_resize_device(src, (ow, oh), interp, flags=1).cpu().numpy()[0]
        assert np.array_equal(got, fx[f"case{i:02d}"]), (i, (h, w), (oh, ow), interp)


# ----------------------------------------------------------------------------- torchvision / timm: key names, transforms, forward
def test_torchvision_vit_keys_and_transform_pin():
    """The torchvision adapter reads exactly the names a torchvision ViT holds, and TRANSFORM_RESIZE / the crop / the
    normalisation constants are weights.transforms()'s (models/patch/base.py:126-180)."""
    import torch
    from atlaspatch_amd.encoders.vit import ARCHS, IMAGENET_MEAN, IMAGENET_STD, TRANSFORM_RESIZE, canonical_state_dict
    jpath, npath = _need("torchvision_vit.json", "torchvision_vit.npz")
    meta = json.load(open(jpath))
    for name in ("vit_b_16", "vit_l_16"):
        m = meta["models"][name]
        shapes = {k: v for k, v in m["state_dict"].items() if not k.startswith("heads.")}
        fake = {k: torch.zeros(v) for k, v in shapes.items()}
        canon = canonical_state_dict(fake, depth=ARCHS[name]["depth"], layer_scale=False, source="auto")
        assert sum(v.numel() for v in canon.values()) == sum(int(np.prod(v)) for v in shapes.values()), "adapter drops / doubles tensors"
        t = m["transforms"]
        assert t["crop_size"] == [224] and t["resize_size"] == [TRANSFORM_RESIZE[name][0]]
        assert TRANSFORM_RESIZE[name][1] in t["interpolation"].lower()
        assert np.allclose(t["mean"], IMAGENET_MEAN) and np.allclose(t["std"], IMAGENET_STD)


def _oracle_vit_l2(name, seed, x):
    import torch
    from atlaspatch_amd.encoders.vit import ARCHS, random_canonical_state_dict
    from oracle import vit_oracle
    arch = dict(ARCHS[name]); arch["depth"] = 2
    sd = random_canonical_state_dict(arch, seed=seed)
    return arch, sd, lambda sd_: vit_oracle.vit_tokens_canonical(sd_, torch.from_numpy(x), heads=arch["heads"], depth=2)[:, 0].numpy()


@pytest.mark.parametrize("name", ["vit_b_16", "vit_l_16"])
def test_torchvision_forward_pin_oracle(name):
    """Oracle forward == torchvision's VisionTransformer on the same seeded weights (depth 2) and the transform's output."""
    from PIL import Image
    from oracle import vit_oracle
    from atlaspatch_amd.encoders.vit import TRANSFORM_RESIZE
    _, npath = _need("torchvision_vit.json", "torchvision_vit.npz")
    fx = np.load(npath)
    size, filt = TRANSFORM_RESIZE[name]
    tiles = kit.seeded_tiles(4)
    pf = Image.Resampling.BILINEAR if filt == "bilinear" else Image.Resampling.BICUBIC
    res = np.stack([np.asarray(Image.fromarray(t).resize((size, size), pf)) if size != 256 else t for t in tiles], 0)
    x = vit_oracle.preprocess_center_crop(res, crop=224).numpy()
    assert np.abs(x[:1] - fx[f"{name}__L2_seed5_input"]).max() <= 1e-6, "transform (resize / crop / normalise) differs"
    _, sd, fwd = _oracle_vit_l2(name, 5, x)
    want = fx[f"{name}__L2_seed5_out"]
    got = fwd(sd)
    assert np.linalg.norm(got - want) / np.linalg.norm(want) <= 1e-5


def test_timm_uni_keys_transform_and_forward_pin():
    import torch
    from PIL import Image
    from atlaspatch_amd.encoders.vit import ARCHS, TRANSFORM_RESIZE, canonical_state_dict
    from oracle import vit_oracle
    jpath, npath = _need("timm_uni.json", "timm_uni.npz")
    meta, fx = json.load(open(jpath)), np.load(npath)
    fake = {k: torch.zeros(v) for k, v in meta["state_dict"].items()}
    canon = canonical_state_dict(fake, depth=24, layer_scale=True, source="auto")
    assert sum(v.numel() for v in canon.values()) == sum(int(np.prod(v)) for v in meta["state_dict"].values())
    cfg = meta["data_config"]
    assert cfg["input_size"] == [3, 224, 224] and cfg["interpolation"] == TRANSFORM_RESIZE["uni_v1"][1]
    # forward, depth 2, LayerScale in [0.2, 0.7]
    tiles = kit.seeded_tiles(4)
    size = TRANSFORM_RESIZE["uni_v1"][0]
    res = np.stack([np.asarray(Image.fromarray(t).resize((size, size), Image.Resampling.BICUBIC)) for t in tiles], 0)
    x = vit_oracle.preprocess_center_crop(res, crop=224, mean=tuple(cfg["mean"]), std=tuple(cfg["std"])).numpy()
    assert np.abs(x[:1] - fx["uni_v1__L2_seed6_input"]).max() <= 1e-6
    arch, sd, fwd = _oracle_vit_l2("uni_v1", 6, x)
    g = torch.Generator().manual_seed(66)
    for i in range(2):
        sd[f"blocks.{i}.ls1"] = torch.rand(1024, generator=g) * 0.5 + 0.2
        sd[f"blocks.{i}.ls2"] = torch.rand(1024, generator=g) * 0.5 + 0.2
    want = fx["uni_v1__L2_seed6_out"]
    assert np.linalg.norm(fwd(sd) - want) / np.linalg.norm(want) <= 1e-5


# ----------------------------------------------------------------------------- sam2 / conch
def test_sam2_keys_and_forward_pin_oracle():
    """oracle/sam2_oracle.py against the sam2 package on the seeded weights: every tensor name the build reads exists in
    the package with that shape, and the low-resolution mask logits agree."""
    from atlaspatch_amd.services.segmentation import random_sam2_state_dict
    from oracle import sam2_oracle
    jpath, npath = _need("sam2_hiera_t.json", "sam2_hiera_t.npz")
    meta, fx = json.load(open(jpath)), np.load(npath)
    sd = random_sam2_state_dict(0)
    assert not meta["seeded_keys_not_in_package"], meta["seeded_keys_not_in_package"][:5]
    for k, v in sd.items():
        assert list(v.shape) == meta["state_dict"][k] or int(np.prod(v.shape)) == int(np.prod(meta["state_dict"][k])), k
    img = np.random.default_rng(8300).integers(0, 256, (1024, 1024, 3), dtype=np.uint8)
    got = sam2_oracle.predict_logits(sd, img)
    want = fx["low_res_logits"]
    assert np.abs(got - want).max() <= 2e-3 * max(1.0, np.abs(want).max())


@pytest.mark.gpu
def test_sam2_device_logits_equal_package_fixture():
    from atlaspatch_amd.services.sam2_hip import Sam2HipPredictor
    from atlaspatch_amd.services.segmentation import random_sam2_state_dict
    _, npath = _need("sam2_hiera_t.json", "sam2_hiera_t.npz")
    fx = np.load(npath)
    pred = Sam2HipPredictor(random_sam2_state_dict(0), device="cuda")
    img = np.random.default_rng(8300).integers(0, 256, (1024, 1024, 3), dtype=np.uint8)
    mask = pred.predict_image(img)
    want = fx["mask_logits_1024_f16"].astype(np.float32) > 0.0
    pred.close()
    assert (mask.astype(bool) != want).mean() <= 5e-4


def test_conch_keys_and_forward_pin_oracle():
    from atlaspatch_amd.encoders.vit import ARCHS, random_attn_pool, random_canonical_state_dict
    from oracle import vit_oracle
    jpath, npath = _need("conch_v1.json", "conch_v1.npz")
    meta, fx = json.load(open(jpath)), np.load(npath)
    assert not meta["seeded_keys_not_in_package"], meta["seeded_keys_not_in_package"][:5]
    arch = ARCHS["conch_v1"]
    trunk = random_canonical_state_dict({k: v for k, v in arch.items() if not k.startswith("pool")}, seed=9)
    pool = random_attn_pool(arch, seed=9)
    tiles = kit.seeded_tiles(2)
    x = vit_oracle.conch_preprocess(tiles).numpy()
    assert np.abs(x[:1] - fx["conch_v1__seed9_input"]).max() <= 1e-6
    got = vit_oracle.conch_encode_image(trunk, pool, tiles, heads=12, depth=12, pool_heads=8)
    want = fx["conch_v1__seed9_out"]
    assert np.linalg.norm(got - want) / np.linalg.norm(want) <= 1e-5
