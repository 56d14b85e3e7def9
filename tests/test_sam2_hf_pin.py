"""SAM2.1 Hiera-T image path pinned against an INDEPENDENT implementation.

``tests/golden/hf_sam2.npz`` holds inputs / outputs of ``transformers.models.sam2.Sam2Model`` (Hiera-T = its default
config) run in the build container by ``tests/golden/gen_golden_hf_sam2.py`` on the seeded weights of
``oracle.sam2_oracle.random_state_dict`` loaded through ``atlaspatch_amd.services.sam2_keys.facebook_to_hf``.  Here:

* CPU: ``oracle/sam2_oracle.py`` (the restatement) against the fixture; the key map's two directions; the checkpoint loader
  on an HF-layout file.
* GPU (``-m gpu``): ``Sam2HipPredictor`` through the C ABI against the fixture — feature levels <= 1e-4 norm-wise, logits
  <= 2e-4, masks through ``predict_image`` on the non-square thumbnail differing in <= 2e-4 of the pixels (only where the
  upsampled logit is within float rounding of the threshold).
"""
from __future__ import annotations

import io
import json
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "hf_sam2.npz")


def _rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


@pytest.fixture(scope="module")
def pin():
    from PIL import Image
    g = np.load(FIXTURE)
    thumb = np.array(Image.open(io.BytesIO(g["thumbnail_png"].tobytes())).convert("RGB"), copy=True)
    img = np.array(Image.fromarray(thumb).resize((1024, 1024), Image.Resampling.BILINEAR), copy=True)
    from oracle import sam2_oracle as so
    sd = so.random_state_dict(int(g["seed"]))
    checksum = sum(float(v.double().sum()) for v in sd.values())
    assert abs(checksum - float(g["weights_checksum"])) <= 1e-6 * abs(float(g["weights_checksum"])), \
        "oracle.sam2_oracle.random_state_dict no longer reproduces the weights the fixture was generated with"
    return dict(g=g, thumb=thumb, img=img, sd=sd, meta=json.loads(g["meta"].tobytes().decode()))


def _mask_from_logits(logits256: np.ndarray, hw) -> np.ndarray:
    """SAM2ImagePredictor.postprocess_masks + the reference's _resize_mask (segmentation.py:112-118) on reference logits."""
    from PIL import Image
    up = torch.nn.functional.interpolate(torch.from_numpy(logits256)[None, None], (1024, 1024), mode="bilinear", align_corners=False)[0, 0]
    m = (up > 0.0).numpy().astype(np.float32)
    h, w = hw
    return np.asarray(Image.fromarray((m * 255).astype(np.uint8), mode="L").resize((w, h), Image.Resampling.NEAREST), dtype=np.float32) / 255.0


# ----------------------------------------------------------------------------- CPU: oracle and key map vs the HF fixture
def test_sam2_oracle_matches_hf_transformers_fixture(pin):
    from oracle import sam2_oracle as so
    g, sd, img = pin["g"], pin["sd"], pin["img"]
    embed, s0, s1 = so.image_features(sd, img)
    assert _rel(embed[0, :, ::4, ::4].numpy(), g["embed_sample"]) <= 1e-5
    assert _rel(s0[0, :, ::16, ::16].numpy(), g["s0_sample"]) <= 1e-5
    assert _rel(s1[0, :, ::8, ::8].numpy(), g["s1_sample"]) <= 1e-5
    norms = np.array([float(embed.norm()), float(s0.norm()), float(s1.norm())])
    assert np.allclose(norms, g["norms"], rtol=1e-5)
    logits = so.mask_decoder(sd, embed, s0, s1).numpy()
    assert _rel(logits, g["logits"]) <= 1e-5, _rel(logits, g["logits"])          # measured 2.8e-7
    # the binary mask the reference would write, from both sets of logits, on the non-square thumbnail
    hw = pin["thumb"].shape[:2]
    assert (_mask_from_logits(logits, hw) != _mask_from_logits(g["logits"], hw)).mean() <= 1e-5
    assert abs(float((g["logits"] > 0).mean()) - float(g["positive_fraction"])) < 1e-12


def test_sam2_oracle_hiera_stage_outputs_match_hf(pin):
    from oracle import sam2_oracle as so
    g, sd, img = pin["g"], pin["sd"], pin["img"]
    x = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).float().div(255)[None]
    x = (x - torch.tensor(so.MEAN).view(1, 3, 1, 1)) / torch.tensor(so.STD).view(1, 3, 1, 1)
    outs = so.hiera_forward(sd, x)                                           # [1, C, H, W] per stage
    assert np.allclose([float(o.norm()) for o in outs], g["stage_norms"], rtol=1e-5)
    for k, o in enumerate(outs):
        s = o.permute(0, 2, 3, 1)
        got = s[0, ::max(1, s.shape[1] // 8), ::max(1, s.shape[2] // 8), :8].numpy()[:8, :8]
        assert _rel(got, g["stage_samples"][k]) <= 1e-5, k


def test_sam2_key_map_round_trip_and_fixture_names(pin):
    from atlaspatch_amd.services.sam2_hip import required_sam2_keys
    from atlaspatch_amd.services.sam2_keys import facebook_key_to_hf, facebook_to_hf, hf_to_facebook, is_hf_layout
    sd = pin["sd"]
    hf = facebook_to_hf(sd)
    assert is_hf_layout(hf) and not is_hf_layout(sd)
    back = hf_to_facebook(hf)
    assert set(back) == set(sd)
    for k in sd:
        assert torch.equal(back[k], sd[k]), k
    # the map the fixture was generated with (names accepted by transformers' Sam2Model) is the one in the tree
    km = pin["meta"]["keymap"]
    assert set(km) == set(sd)
    for k, v in km.items():
        assert facebook_key_to_hf(k) == v, k
    # every tensor the device path reads has an HF slot (directly, stacked or duplicated)
    special = ("sam_prompt_encoder.pe_layer.", "sam_prompt_encoder.point_embeddings.")
    assert all(k.startswith(special) or facebook_key_to_hf(k) for k in required_sam2_keys())
    # video-memory tensors of a full facebook checkpoint are dropped, not mis-mapped
    extra = dict(sd)
    extra["memory_attention.layers.0.norm1.weight"] = torch.zeros(256)
    extra["obj_ptr_proj.layers.0.weight"] = torch.zeros(256, 256)
    assert set(facebook_to_hf(extra)) == set(hf)


def test_sam2_loader_accepts_the_hf_layout(pin, tmp_path):
    from atlaspatch_amd.services.sam2_hip import load_sam2_state_dict
    from atlaspatch_amd.services.sam2_keys import facebook_to_hf
    sd = pin["sd"]
    hf = {k: v.contiguous() for k, v in facebook_to_hf(sd).items()}
    p = tmp_path / "model.pt"
    torch.save(hf, p)
    got = load_sam2_state_dict(p)
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    from safetensors.torch import save_file
    q = tmp_path / "model.safetensors"
    # HF stores the Gaussian matrix under two names that share storage in the facebook layout: clone for safetensors
    save_file({k: v.clone() for k, v in hf.items()}, str(q))
    got = load_sam2_state_dict(q)
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)


def test_hf_sam2_still_reproduces_the_fixture_when_transformers_is_present(pin):
    """Re-runs the independent implementation when it is importable (it is in the build container and on the GPU box
    image); skipped elsewhere.  Guards the fixture against silent staleness."""
    sam2 = pytest.importorskip("transformers.models.sam2")
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_golden_hf_sam2", os.path.join(HERE, "golden", "gen_golden_hf_sam2.py"))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    model, _ = gen.build_hf_model(pin["sd"])
    x = torch.from_numpy(np.ascontiguousarray(pin["img"])).permute(2, 0, 1).float().div(255)[None]
    x = (x - torch.tensor(gen.MEAN).view(1, 3, 1, 1)) / torch.tensor(gen.STD).view(1, 3, 1, 1)
    with torch.inference_mode():
        out = model(pixel_values=x, input_boxes=torch.tensor([[[0.0, 0.0, 1024.0, 1024.0]]]), multimask_output=False)
    assert _rel(out.pred_masks[0, 0, 0].numpy(), pin["g"]["logits"]) <= 1e-5


# ----------------------------------------------------------------------------- GPU: the HIP path vs the HF fixture
@pytest.mark.gpu
def test_sam2_hip_predictor_matches_hf_transformers_fixture(pin):
    from atlaspatch_amd.services.sam2_hip import Sam2HipPredictor
    g, sd, img, thumb = pin["g"], pin["sd"], pin["img"], pin["thumb"]
    dev = torch.device("cuda:0")
    pred = Sam2HipPredictor(sd, device=dev)
    try:
        embed, s0, s1 = pred.image_features(torch.from_numpy(np.ascontiguousarray(img)).to(dev))     # token-major [HW, C]
        # the device embed already carries no_mask_embed (folded with no_mem_embed); the fixture's does not
        nomask = sd["sam_prompt_encoder.no_mask_embed.weight"].reshape(1, 256).to(dev)
        e = (embed - nomask).reshape(64, 64, 256).permute(2, 0, 1).cpu().numpy()
        a0 = s0.reshape(256, 256, 32).permute(2, 0, 1).cpu().numpy()
        a1 = s1.reshape(128, 128, 64).permute(2, 0, 1).cpu().numpy()
        assert _rel(e[:, ::4, ::4], g["embed_sample"]) <= 1e-4
        assert _rel(a0[:, ::16, ::16], g["s0_sample"]) <= 1e-4
        assert _rel(a1[:, ::8, ::8], g["s1_sample"]) <= 1e-4
        assert np.allclose([np.linalg.norm(e), np.linalg.norm(a0), np.linalg.norm(a1)], g["norms"], rtol=1e-4)
        logits = pred.mask_logits(embed, s0, s1).cpu().numpy()
        assert logits.shape == (256, 256)
        assert _rel(logits, g["logits"]) <= 2e-4, _rel(logits, g["logits"])
        # end to end as the reference drives it: non-square thumbnail -> BILINEAR 1024^2 -> forward -> > 0 -> NEAREST back
        want = _mask_from_logits(g["logits"], thumb.shape[:2])
        got = pred.predict_image(thumb)
        assert got.shape == want.shape == thumb.shape[:2] and set(np.unique(got)) <= {0.0, 1.0}
        assert (got != want).mean() <= 2e-4, (got != want).mean()
        got_dev = pred.predict_device(torch.from_numpy(thumb).to(dev))
        assert np.array_equal(np.asarray(got_dev), got)
        batch = pred.predict_batch_device([torch.from_numpy(thumb).to(dev)] * 3)
        assert all((np.asarray(m) != want).mean() <= 2e-4 for m in batch)
    finally:
        pred.close()


@pytest.mark.gpu
def test_sam2_hip_predictor_from_an_hf_layout_checkpoint(pin, tmp_path):
    """The loader path a user with a transformers export takes: HF names on disk -> facebook names -> device."""
    from atlaspatch_amd.services.sam2_hip import Sam2HipPredictor, load_sam2_state_dict
    from atlaspatch_amd.services.sam2_keys import facebook_to_hf
    p = tmp_path / "hf_model.pt"
    torch.save({k: v.contiguous() for k, v in facebook_to_hf(pin["sd"]).items()}, p)
    pred = Sam2HipPredictor(load_sam2_state_dict(p), device=torch.device("cuda:0"))
    try:
        logits = pred.predict_logits(pin["img"]).cpu().numpy()
        assert _rel(logits.reshape(256, 256), pin["g"]["logits"]) <= 2e-4
    finally:
        pred.close()
