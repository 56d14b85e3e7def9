"""The rest of the reference's ViT encoder files (SURVEY §8 a14-a15 widened, VERDICT r03 "missing #3"):
vit_b_32 / vit_l_32 / vit_h_14 (models/patch/vit.py:9-15) and uni_v2 (models/patch/uni.py:62-125).

CPU (not gpu):
  * ``oracle/vit_oracle.py::vit_tokens_canonical`` -- the restatement the GPU tests compare with -- is pinned against
    INDEPENDENT implementations that are importable here: transformers' ``ViTModel`` (patch 32; patch 14 with 80-wide heads)
    and ``Dinov2WithRegistersModel`` (register tokens + SwiGLU + LayerScale = UNI2-h's block structure), through the product's
    key adapters (``canonical_state_dict``), small depth, seeded weights;
  * head padding (80 -> 128) leaves the function unchanged; registry names; transform table.
GPU (-m gpu): each encoder at its REAL size through the C ABI against the oracle, bounds = measured x headroom.
"""
from __future__ import annotations

import math

import numpy as np
import pytest
import torch


def _rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _seed_params(model, seed=1):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if ("norm" in name and name.endswith("weight")) or "lambda1" in name:
                p.copy_((0.3 if "lambda1" in name else 1.0) + 0.1 * torch.randn(p.shape, generator=g))
            elif name.endswith("bias") or "position_embeddings" in name or "cls_token" in name or "register_tokens" in name:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            elif p.dim() >= 2:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    return model


# ----------------------------------------------------------------------------- CPU: the oracle vs independent implementations
@pytest.mark.parametrize("patch,hidden,heads,image", [(32, 128, 2, 224), (14, 160, 2, 56), (16, 128, 2, 64)])
def test_oracle_matches_hf_vit_for_other_patch_sizes_and_head_widths(patch, hidden, heads, image):
    """transformers ViTModel (what G1 pins the oracle with at patch 16) at patch 32 (vit_b_32 / vit_l_32) and at patch 14 with
    80-wide heads (vit_h_14: 1280 / 16 = 80; here 160 / 2)."""
    from transformers import ViTConfig, ViTModel
    from atlaspatch_amd.encoders.vit import canonical_state_dict
    from oracle import vit_oracle
    torch.manual_seed(0)
    cfg = ViTConfig(hidden_size=hidden, num_hidden_layers=2, num_attention_heads=heads, intermediate_size=4 * hidden,
                    image_size=image, patch_size=patch, layer_norm_eps=1e-6, hidden_act="gelu")
    model = _seed_params(ViTModel(cfg, add_pooling_layer=False).eval())
    x = torch.randn(3, 3, image, image, generator=torch.Generator().manual_seed(2))
    with torch.inference_mode():
        want = model(pixel_values=x).last_hidden_state
    sd = canonical_state_dict(dict(model.state_dict()), depth=2, layer_scale=False, source="hf")
    got = vit_oracle.vit_tokens_canonical(sd, x, heads=heads, depth=2)
    assert got.shape == want.shape == (3, 1 + (image // patch) ** 2, hidden)
    assert _rel(got.numpy(), want.numpy()) <= 2e-6


@pytest.mark.parametrize("swiglu", [True, False])
def test_oracle_matches_hf_dinov2_with_registers(swiglu):
    """UNI2-h's block structure (uni.py:82-96: register tokens, no_embed_class, SwiGLUPacked, LayerScale) as implemented by
    transformers' Dinov2WithRegistersModel; the adapter folds its class position row into the class token."""
    from transformers import Dinov2WithRegistersConfig, Dinov2WithRegistersModel
    from atlaspatch_amd.encoders.vit import canonical_state_dict
    from oracle import vit_oracle
    torch.manual_seed(0)
    cfg = Dinov2WithRegistersConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, mlp_ratio=4,
                                    use_swiglu_ffn=swiglu, num_register_tokens=8, patch_size=14, image_size=56,
                                    layerscale_value=1e-5, layer_norm_eps=1e-6)
    model = _seed_params(Dinov2WithRegistersModel(cfg).eval())
    x = torch.randn(3, 3, 56, 56, generator=torch.Generator().manual_seed(3))
    with torch.inference_mode():
        out = model(pixel_values=x)
    sd = canonical_state_dict(dict(model.state_dict()), depth=2, layer_scale=True)           # auto-detected: hf_dinov2
    assert sd["reg_tokens"].shape == (8, 128) and sd["pos_embed"].shape == (16, 128)
    if swiglu:
        assert sd["blocks.0.fc1.weight"].shape[0] == 2 * sd["blocks.0.fc2.weight"].shape[1]
    got = vit_oracle.vit_tokens_canonical(sd, x, heads=2, depth=2)
    assert got.shape == out.last_hidden_state.shape == (3, 1 + 8 + 16, 128)
    assert _rel(got.numpy(), out.last_hidden_state.numpy()) <= 2e-6
    assert _rel(got[:, 0].numpy(), out.pooler_output.numpy()) <= 2e-6                       # class token = what uni_v2 returns


@pytest.mark.parametrize("swiglu,grid_ckpt,grid_in", [(False, 5, 4), (True, 5, 4), (False, 4, 4)])
def test_oracle_matches_hf_dinov2_with_resampled_positions(swiglu, grid_ckpt, grid_in):
    """models/patch/dinov2.py / phikon.py (phikon_v2): transformers' Dinov2Model -- the module the reference itself calls --
    on an input whose patch grid differs from the checkpoint's (facebook/dinov2-*: 37 x 37 stored, 16 x 16 at the processor's
    224-px crop): the model resamples its position rows in every forward; the adapter does it once (``resample_position_grid``)."""
    from transformers import Dinov2Config, Dinov2Model
    from atlaspatch_amd.encoders.vit import canonical_state_dict
    from oracle import vit_oracle
    torch.manual_seed(0)
    cfg = Dinov2Config(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, mlp_ratio=4, use_swiglu_ffn=swiglu,
                       patch_size=14, image_size=14 * grid_ckpt, layerscale_value=1.0, layer_norm_eps=1e-6)
    model = _seed_params(Dinov2Model(cfg).eval())
    x = torch.randn(3, 3, 14 * grid_in, 14 * grid_in, generator=torch.Generator().manual_seed(4))
    with torch.inference_mode():
        out = model(pixel_values=x)
    sd = canonical_state_dict(dict(model.state_dict()), depth=2, layer_scale=True, grid=grid_in)      # auto-detected: hf_dinov2
    assert "reg_tokens" not in sd and sd["pos_embed"].shape == (grid_in * grid_in, 128)
    got = vit_oracle.vit_tokens_canonical(sd, x, heads=2, depth=2)
    assert got.shape == out.last_hidden_state.shape == (3, 1 + grid_in * grid_in, 128)
    assert _rel(got.numpy(), out.last_hidden_state.numpy()) <= 2e-6
    # what the reference returns: last_hidden_state[:, 0] (dinov2.py:58-60); midnight.py:58-61: class token | mean patch token
    assert _rel(got[:, 0].numpy(), out.last_hidden_state[:, 0].numpy()) <= 2e-6
    lhs = out.last_hidden_state
    want = torch.cat([lhs[:, 0, :], lhs[:, 1:, :].mean(1)], dim=-1).numpy()
    assert _rel(torch.cat([got[:, 0], got[:, 1:].mean(1)], -1).numpy(), want) <= 2e-6


def test_oracle_matches_hf_vit_with_the_phikon_layer_norm_eps():
    """phikon_v1 (phikon.py:36-56) = transformers ViTModel(add_pooling_layer=False) with ViTConfig's default LayerNorm eps 1e-12."""
    from transformers import ViTConfig, ViTModel
    from atlaspatch_amd.encoders.vit import ARCHS, canonical_state_dict
    from oracle import vit_oracle
    torch.manual_seed(0)
    cfg = ViTConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512, image_size=64, patch_size=16)
    assert cfg.layer_norm_eps == ARCHS["phikon_v1"]["ln_eps"] == 1e-12
    model = _seed_params(ViTModel(cfg, add_pooling_layer=False).eval())
    x = torch.randn(3, 3, 64, 64, generator=torch.Generator().manual_seed(2))
    with torch.inference_mode():
        want = model(pixel_values=x).last_hidden_state
    sd = canonical_state_dict(dict(model.state_dict()), depth=2, layer_scale=False, source="hf")
    got = vit_oracle.vit_tokens_canonical(sd, x, heads=2, depth=2, eps=1e-12)
    assert _rel(got.numpy(), want.numpy()) <= 2e-6


def _hf_dinov3(hidden=128, layers=2, heads=2, inter=256, image=64, gated=False):
    from transformers import DINOv3ViTConfig, DINOv3ViTModel
    torch.manual_seed(0)
    cfg = DINOv3ViTConfig(hidden_size=hidden, intermediate_size=inter, num_attention_heads=heads, num_hidden_layers=layers,
                          num_register_tokens=4, image_size=image, patch_size=16, use_gated_mlp=gated, hidden_act="silu" if gated else "gelu")
    return _seed_params(DINOv3ViTModel(cfg).eval())


@pytest.mark.parametrize("gated", [False, True])
def test_oracle_matches_hf_dinov3_with_the_rotary_embedding(gated):
    """models/patch/dinov3.py:52-67 calls transformers' DINOv3ViTModel and returns pooler_output -- importable here: class + 4
    register tokens, no position embedding, rotary embedding on q / k of the patch tokens only, k_proj without bias, LayerScale,
    plain or gated MLP, LayerNorm 1e-5.  The oracle through the hf_dinov3 adapter (rotary tables rebuilt as the HF module builds
    them) reproduces it."""
    from atlaspatch_amd.encoders.vit import canonical_state_dict
    from oracle import vit_oracle
    model = _hf_dinov3(gated=gated)
    x = torch.randn(3, 3, 64, 64, generator=torch.Generator().manual_seed(9))
    with torch.inference_mode():
        out = model(pixel_values=x)
        cos, sin = model.rope_embeddings(x)
    sd = canonical_state_dict(dict(model.state_dict()), depth=2, layer_scale=True, grid=4, heads=2)          # auto-detected: hf_dinov3
    assert torch.equal(sd["rope.cos"], cos) and torch.equal(sd["rope.sin"], sin) and sd["reg_tokens"].shape == (4, 128)
    assert float(sd["pos_embed"].abs().max()) == 0.0 and float(sd["blocks.0.qkv.bias"][128:256].abs().max()) == 0.0    # no k bias
    tok = vit_oracle.vit_tokens_canonical(sd, x, heads=2, depth=2, eps=1e-5)
    assert _rel(tok.numpy(), out.last_hidden_state.numpy()) <= 2e-6
    assert _rel(tok[:, 0].numpy(), out.pooler_output.numpy()) <= 2e-6
    # the published facebook/dinov3-* model.safetensors carry `layer.<i>.` where the in-memory module says `model.layer.<i>.`
    # (transformers renames on load); a checkpoint read raw from disk must be recognised and give the same canonical tensors
    disk = {(k[len("model."):] if k.startswith("model.layer.") else k): v for k, v in model.state_dict().items()}
    assert any(k.startswith("layer.0.") for k in disk) and not any(k.startswith("model.") for k in disk)
    from atlaspatch_amd.encoders.vit import _detect_source
    assert _detect_source(disk) == "hf_dinov3"
    sd_disk = canonical_state_dict(disk, depth=2, layer_scale=True, grid=4, heads=2)
    assert sd_disk.keys() == sd.keys() and all(torch.equal(sd_disk[k], sd[k]) for k in sd)


def test_mlp_padding_leaves_the_function_unchanged():
    """pad_mlp (Virchow's 3416-wide SwiGLU -> 3456): zero rows in fc1, zero columns in fc2, both halves of the packed layer."""
    from atlaspatch_amd.encoders.vit import pad_mlp, random_canonical_state_dict, stored_mlp_dim
    from oracle import vit_oracle
    assert stored_mlp_dim(3416) == 3456 and stored_mlp_dim(4096) == 4096
    for swiglu in (True, False):
        arch = dict(image_size=28, patch_size=14, dim=128, depth=2, heads=2, mlp_dim=200, ln_eps=1e-6, mlp="swiglu" if swiglu else None)
        sd = random_canonical_state_dict(arch, seed=7)
        pad = pad_mlp(sd, mlp_dim=200, depth=2, swiglu=swiglu)
        assert pad["blocks.0.fc1.weight"].shape == ((2 if swiglu else 1) * 256, 128) and pad["blocks.0.fc2.weight"].shape == (128, 256)
        x = torch.randn(2, 3, 28, 28, generator=torch.Generator().manual_seed(8))
        a = vit_oracle.vit_tokens_canonical(sd, x, heads=2, depth=2)
        b = vit_oracle.vit_tokens_canonical(pad, x, heads=2, depth=2)
        assert _rel(b.numpy(), a.numpy()) <= 1e-6


def _hf_clip(hidden=128, layers=2, heads=2, image=64, patch=32, proj=128):
    from transformers import CLIPConfig, CLIPModel
    torch.manual_seed(0)
    vc = dict(hidden_size=hidden, intermediate_size=4 * hidden, num_hidden_layers=layers, num_attention_heads=heads, image_size=image,
              patch_size=patch, projection_dim=proj, hidden_act="quick_gelu", layer_norm_eps=1e-5)
    tc = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2, vocab_size=100,
              max_position_embeddings=16, bos_token_id=1, eos_token_id=2, pad_token_id=0)
    return _seed_params(CLIPModel(CLIPConfig(text_config=tc, vision_config=vc, projection_dim=proj)).eval())


def _clip_image_features(model, x):
    with torch.inference_mode():
        f = model.get_image_features(pixel_values=x)
    return (f if torch.is_tensor(f) else f.pooler_output).numpy()


@pytest.mark.parametrize("patch,image", [(32, 64), (16, 64), (14, 56)])
def test_oracle_matches_hf_clip_get_image_features(patch, image):
    """plip.py:56 / quilt.py:59-60 call transformers' CLIPModel.get_image_features -- importable here: no patch-embedding bias,
    pre_layrnorm, QuickGELU, LayerNorm 1e-5, post_layernorm of the class token, bias-free visual projection.  The oracle through
    the product's hf_clip adapter reproduces it (the same towers serve clip.py's open_clip ViT-B-32 / B-16 / L-14)."""
    from atlaspatch_amd.encoders.vit import canonical_state_dict
    from oracle import vit_oracle
    model = _hf_clip(patch=patch, image=image)
    x = torch.randn(3, 3, image, image, generator=torch.Generator().manual_seed(5))
    want = _clip_image_features(model, x)
    sd = canonical_state_dict(dict(model.state_dict()), depth=2, layer_scale=False)            # auto-detected: hf_clip
    assert "pre_norm.weight" in sd and sd["head_proj.weight"].shape == (128, 128) and float(sd["patch_embed.bias"].abs().max()) == 0.0
    tok = vit_oracle.vit_tokens_canonical(sd, x, heads=2, depth=2, eps=1e-5, act="quick_gelu")
    got = (tok[:, 0] @ sd["head_proj.weight"].T).numpy()
    assert _rel(got, want) <= 2e-6
    # the open_clip layout of the same tower (clip.py) maps to the same canonical tensors
    oc = {"visual.conv1.weight": sd["patch_embed.weight"], "visual.class_embedding": sd["cls_token"],
          "visual.positional_embedding": sd["pos_embed"], "visual.ln_pre.weight": sd["pre_norm.weight"],
          "visual.ln_pre.bias": sd["pre_norm.bias"], "visual.ln_post.weight": sd["norm.weight"], "visual.ln_post.bias": sd["norm.bias"],
          "visual.proj": sd["head_proj.weight"].t().contiguous()}
    for i in range(2):
        b, p = f"blocks.{i}.", f"visual.transformer.resblocks.{i}."
        for src, dst in (("ln1", "ln_1"), ("ln2", "ln_2"), ("proj", "attn.out_proj"), ("fc1", "mlp.c_fc"), ("fc2", "mlp.c_proj")):
            oc[p + dst + ".weight"] = sd[b + src + ".weight"]; oc[p + dst + ".bias"] = sd[b + src + ".bias"]
        oc[p + "attn.in_proj_weight"] = sd[b + "qkv.weight"]; oc[p + "attn.in_proj_bias"] = sd[b + "qkv.bias"]
    back = canonical_state_dict(oc, depth=2, layer_scale=False)                                # auto-detected: open_clip
    assert set(back) == set(sd) and all(torch.equal(back[k], sd[k]) for k in sd)


def test_head_padding_leaves_the_function_unchanged():
    """pad_heads: 80-wide heads stored 96 wide (zero rows in q / k / v, zero columns in proj) with the softmax scale kept at
    1 / sqrt(80) -- checked with an explicit attention on the padded tensors."""
    from atlaspatch_amd.encoders.vit import pad_heads, random_canonical_state_dict, stored_head_dim
    arch = dict(image_size=28, patch_size=14, dim=160, depth=1, heads=2, mlp_dim=320, ln_eps=1e-6)
    sd = random_canonical_state_dict(arch, seed=5)
    assert stored_head_dim(160, 2) == 96 and stored_head_dim(768, 12) == 64 and stored_head_dim(1280, 16) == 96
    assert stored_head_dim(200, 2) == 128 and stored_head_dim(4096, 32) == 128
    pad = pad_heads(sd, dim=160, heads=2, depth=1)
    assert pad["blocks.0.qkv.weight"].shape == (3 * 2 * 96, 160) and pad["blocks.0.proj.weight"].shape == (160, 192)
    h = torch.randn(2, 5, 160, generator=torch.Generator().manual_seed(6))

    def attn(s, hd, scale):
        qkv = h @ s["blocks.0.qkv.weight"].T + s["blocks.0.qkv.bias"]
        q, k, v = qkv.view(2, 5, 3, 2, hd).permute(2, 0, 3, 1, 4)
        ctx = (torch.softmax(q @ k.transpose(-1, -2) * scale, -1) @ v).transpose(1, 2).reshape(2, 5, 2 * hd)
        return ctx @ s["blocks.0.proj.weight"].T + s["blocks.0.proj.bias"]
    assert _rel(attn(pad, 96, 1 / math.sqrt(80)).numpy(), attn(sd, 80, 1 / math.sqrt(80)).numpy()) <= 1e-6
    assert pad_heads(sd | {}, dim=128, heads=2, depth=0) is not None          # 64-wide: returned as is


def test_registry_has_the_reference_names_of_the_three_encoder_files():
    """models/patch/vit.py:9-15 (five names), uni.py (uni_v1, uni_v2), conch.py (conch_v1; conch_v15 = code downloaded from the
    gated MahmoodLab/TITAN repository at run time, conch.py:82-86 -- nothing in the reference to restate)."""
    from atlaspatch_amd.encoders import build_default_registry
    from atlaspatch_amd.encoders.vit import ARCHS, TRANSFORM_RESIZE
    names = build_default_registry(device="cpu").available()
    for n in ("vit_b_16", "vit_b_32", "vit_l_16", "vit_l_32", "vit_h_14", "uni_v1", "uni_v2", "conch_v1",
              "dinov2_small", "dinov2_base", "dinov2_large", "dinov2_giant", "phikon_v1", "phikon_v2",        # dinov2.py:12-17, phikon.py
              "midnight", "h_optimus_0", "h_optimus_1", "prov_gigapath", "lunit_vit_small_patch16_dino",
              "lunit_vit_small_patch8_dino", "pathorchestra",
              "clip_vit_b_32", "clip_vit_b_16", "clip_vit_l_14", "clip_vit_l_14_336", "plip", "quilt_b_32", "quilt_b_16",      # clip.py:16-19
              "biomedclip", "virchow_v1", "virchow_v2", "h0_mini",
              "dinov3_vits16", "dinov3_vits16_plus", "dinov3_vitb16", "dinov3_vitl16", "dinov3_vitl16_sat", "dinov3_vith16_plus",
              "dinov3_vit7b16", "dinov3_vit7b16_sat"):
        assert n in names and n in ARCHS and n in TRANSFORM_RESIZE
    g = ARCHS["dinov2_giant"]
    assert (g["dim"], g["depth"], g["heads"], g["mlp_dim"]) == (1536, 40, 24, (int(1536 * 4 * 2 / 3) + 7) // 8 * 8)
    assert ARCHS["vit_h_14"]["image_size"] == 518 and TRANSFORM_RESIZE["vit_h_14"] == (518, "bicubic")
    a = ARCHS["uni_v2"]
    assert (a["dim"], a["depth"], a["heads"], a["mlp_dim"], a["reg_tokens"], a["patch_size"]) == (1536, 24, 24, 4096, 8, 14)
    assert int(1536 * 2.66667 * 2) == 2 * a["mlp_dim"]                       # timm: hidden_features = int(dim * mlp_ratio)


# ----------------------------------------------------------------------------- GPU: real sizes through the C ABI
# measured on MI355X (profiles/r04_parity_lines.txt): (norm-wise, element-wise max, q99.9)
MEASURED = {
    ("vit_b_32 L12", "float16"): (8.90e-4, 1.19e-2, 8.53e-3),
    ("vit_b_32 L12, f32_stream", "float16"): (9.49e-4, 1.229e-2, 9.15e-3),
    ("vit_b_32 L12", "float32"): (2.30e-6, 2.79e-5, 2.20e-5),
    ("vit_l_32 L24", "float16"): (9.87e-4, 1.24e-2, 1.07e-2),
    ("vit_l_32 L24, f32_stream", "float16"): (1.111e-3, 1.357e-2, 1.134e-2),
    # uni_v2: SwiGLU gate in the fc1 epilogue on the f32 values (round 4: 2.94e-3 with the gate as a separate pass on the
    # rounded fc1 output).  The product of two branches and dim 1536 give ~1.8 x uni_v1's error in every mode (float32: 6e-6)
    ("uni_v2 L24", "float16"): (2.05e-3, 2.72e-2, 2.33e-2),
    ("uni_v2 L24, f32_stream", "float16"): (2.526e-3, 3.558e-2, 2.909e-2),
    ("uni_v2 L24", "float32"): (6.00e-6, 1.117e-4, 7.00e-5),
    ("vit_h_14 L32", "float16"): (8.54e-4, 1.41e-2, 9.90e-3),
    ("vit_h_14 L32, f32_stream", "float16"): (9.04e-4, 2.090e-2, 1.378e-2),
}
MEASURED.update({                     # the transformers-backed encoders (dinov2.py, phikon.py)
    ("dinov2_small L12", "float16"): (7.58e-4, 1.03e-2, 9.58e-3),
    ("dinov2_small L12, f32_stream", "float16"): (8.13e-4, 1.153e-2, 9.96e-3),
    ("dinov2_small L12", "float32"): (1.571e-6, 2.843e-5, 1.952e-5),
    ("dinov2_base L12", "float16"): (8.96e-4, 1.14e-2, 9.99e-3),
    ("dinov2_base L12, f32_stream", "float16"): (9.98e-4, 1.655e-2, 1.123e-2),
    ("dinov2_large L24", "float16"): (8.98e-4, 1.19e-2, 8.52e-3),
    ("dinov2_large L24, f32_stream", "float16"): (9.90e-4, 1.461e-2, 9.70e-3),
    # 40 SwiGLU blocks at dim 1536 with LayerScale drawn in [0.2, 0.7] (the test's, not the checkpoints' 1e-5 .. 1): uni_v2's
    # per-block error (2.8e-3 over 24 blocks) over 40
    ("dinov2_giant L40", "float16"): (3.08e-3, 5.19e-2, 3.19e-2),
    ("dinov2_giant L40, f32_stream", "float16"): (3.315e-3, 4.796e-2, 3.527e-2),
    ("phikon_v1 L12", "float16"): (7.65e-4, 1.03e-2, 8.95e-3),
    ("phikon_v1 L12, f32_stream", "float16"): (9.14e-4, 1.484e-2, 1.025e-2),
    ("phikon_v1 L12", "float32"): (1.949e-6, 2.594e-5, 2.134e-5),
    ("phikon_v2 L24", "float16"): (8.66e-4, 1.42e-2, 9.67e-3),
    ("phikon_v2 L24, f32_stream", "float16"): (9.86e-4, 1.527e-2, 1.200e-2),
    # midnight.py (class token | mean patch token: the mean averages the patch rows' errors), the timm-hub ViTs
    ("midnight L40", "float16"): (2.21e-3, 4.04e-2, 2.38e-2),
    ("midnight L40, f32_stream", "float16"): (2.408e-3, 3.742e-2, 2.504e-2),
    ("h_optimus_0 L40", "float16"): (1.06e-3, 1.69e-2, 1.24e-2),
    ("h_optimus_0 L40, f32_stream", "float16"): (1.214e-3, 1.546e-2, 1.353e-2),
    ("prov_gigapath L40", "float16"): (3.34e-3, 4.41e-2, 3.94e-2),
    ("prov_gigapath L40, f32_stream", "float16"): (3.821e-3, 4.545e-2, 3.699e-2),
    ("lunit_vit_small_patch16_dino L12", "float16"): (7.48e-4, 1.33e-2, 9.40e-3),
    ("lunit_vit_small_patch16_dino L12, f32_stream", "float16"): (8.39e-4, 1.278e-2, 1.089e-2),
    ("lunit_vit_small_patch8_dino L12", "float16"): (7.71e-4, 9.27e-3, 8.24e-3),
    ("lunit_vit_small_patch8_dino L12, f32_stream", "float16"): (8.37e-4, 1.174e-2, 1.005e-2),
    ("pathorchestra L24", "float16"): (8.96e-4, 1.07e-2, 8.83e-3),
    ("pathorchestra L24, f32_stream", "float16"): (1.008e-3, 1.532e-2, 1.154e-2),
    # CLIP towers (ln_pre, QuickGELU epilogue, projection in the compute type)
    ("clip_vit_b_32 L12", "float16"): (5.69e-4, 8.79e-3, 6.33e-3),
    ("clip_vit_b_32 L12, f32_stream", "float16"): (5.95e-4, 1.043e-2, 7.54e-3),
    ("clip_vit_b_32 L12", "float32"): (1.300e-6, 1.883e-5, 1.475e-5),
    ("clip_vit_b_16 L12", "float16"): (5.49e-4, 7.40e-3, 5.76e-3),
    ("clip_vit_b_16 L12, f32_stream", "float16"): (6.05e-4, 1.331e-2, 6.91e-3),
    ("clip_vit_l_14 L24", "float16"): (6.71e-4, 9.90e-3, 7.78e-3),
    ("clip_vit_l_14 L24, f32_stream", "float16"): (6.93e-4, 1.356e-2, 9.09e-3),
    ("clip_vit_l_14_336 L24", "float16"): (7.18e-4, 8.14e-3, 7.18e-3),
    ("clip_vit_l_14_336 L24, f32_stream", "float16"): (7.64e-4, 9.78e-3, 8.78e-3),
    ("plip L12", "float16"): (5.69e-4, 8.79e-3, 6.33e-3),
    ("plip L12, f32_stream", "float16"): (5.95e-4, 1.043e-2, 7.54e-3),
    ("biomedclip L12", "float16"): (8.19e-4, 1.37e-2, 8.86e-3),
    ("biomedclip L12, f32_stream", "float16"): (8.87e-4, 1.722e-2, 1.077e-2),
    ("virchow_v1 L32", "float16"): (1.54e-3, 2.26e-2, 1.90e-2),
    ("virchow_v1 L32, f32_stream", "float16"): (1.788e-3, 2.851e-2, 2.082e-2),
    ("virchow_v2 L32", "float16"): (1.55e-3, 2.92e-2, 1.92e-2),
    ("virchow_v2 L32, f32_stream", "float16"): (1.812e-3, 2.682e-2, 2.235e-2),
    ("h0_mini L12", "float16"): (5.29e-4, 9.99e-3, 6.98e-3),
    ("h0_mini L12, f32_stream", "float16"): (5.61e-4, 1.009e-2, 7.89e-3),
    # DINOv3 (rotary embedding in f32, one rounding)
    ("dinov3_vits16 L12", "float16"): (7.60e-4, 1.24e-2, 8.52e-3),
    ("dinov3_vits16 L12, f32_stream", "float16"): (8.05e-4, 1.204e-2, 9.60e-3),
    ("dinov3_vits16 L12", "float32"): (1.381e-6, 2.111e-5, 1.757e-5),
    ("dinov3_vits16_plus L12", "float16"): (7.07e-4, 8.77e-3, 7.54e-3),
    ("dinov3_vits16_plus L12, f32_stream", "float16"): (8.18e-4, 8.69e-3, 7.55e-3),
    ("dinov3_vitb16 L12", "float16"): (8.36e-4, 1.27e-2, 1.03e-2),
    ("dinov3_vitb16 L12, f32_stream", "float16"): (8.99e-4, 1.632e-2, 1.145e-2),
    ("dinov3_vitl16 L24", "float16"): (8.80e-4, 1.18e-2, 9.79e-3),
    ("dinov3_vitl16 L24, f32_stream", "float16"): (9.62e-4, 1.380e-2, 1.103e-2),
    ("dinov3_vith16_plus L32", "float16"): (2.17e-3, 2.90e-2, 2.02e-2),
    ("dinov3_vit7b16 width L3", "float16"): (1.469e-3, 1.766e-2, 1.343e-2),        # real width (4096 / 8192), 3 of the 40 blocks
    ("dinov3_vith16_plus L32, f32_stream", "float16"): (2.457e-3, 2.948e-2, 2.717e-2),
})
HEADROOM = (1.2, 1.5, 1.25)
FALLBACK = {"float32": (3.0e-6, 6e-5, 4e-5), "float16": (3.6e-3, 5.2e-2, 3.8e-2)}


def _elem(a, b, floor=0.05, q=None):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    e = np.abs(a - b) / (np.abs(b) + floor * np.abs(b).max())
    return float(e.max() if q is None else np.quantile(e, q))


def _check(got, want, dtype, what):
    key = str(dtype).split(".")[-1]
    r, e, eq = _rel(got, want), _elem(got, want), _elem(got, want, q=0.999)
    m = MEASURED.get((what, key))
    br, be, bq = (tuple(v * h for v, h in zip(m, HEADROOM)) if m else FALLBACK[key])
    print(f"PARITY {what} {key}: norm-wise {r:.3e} (bound {br:.2e}) element-wise max {e:.3e} ({be:.2e}) q99.9 {eq:.3e} ({bq:.2e})")
    assert r <= br and e <= be and eq <= bq, (what, key, (r, br), (e, be), (eq, bq))
    if dtype == torch.float32:
        assert max(r, e, eq) <= 1e-3


def _tiles(n, seed):
    rng = np.random.default_rng(seed)
    return [rng.integers(0, 256, (256, 256, 3), dtype=np.uint8) for _ in range(n)]


def _with_layer_scale(sd, arch, seed):
    g = torch.Generator().manual_seed(seed)
    if arch.get("layer_scale"):
        for i in range(arch["depth"]):
            sd[f"blocks.{i}.ls1"] = torch.rand(arch["dim"], generator=g) * 0.5 + 0.2
            sd[f"blocks.{i}.ls2"] = torch.rand(arch["dim"], generator=g) * 0.5 + 0.2
    return sd


@pytest.mark.gpu
@pytest.mark.parametrize("name,dtype,n", [("vit_b_32", torch.float16, 8), ("vit_b_32", torch.float32, 8), ("vit_l_32", torch.float16, 8),
                                          ("uni_v2", torch.float16, 6), ("uni_v2", torch.float32, 4), ("vit_h_14", torch.float16, 3),
                                          ("dinov2_small", torch.float16, 8), ("dinov2_small", torch.float32, 8),
                                          ("dinov2_base", torch.float16, 8), ("dinov2_large", torch.float16, 6),
                                          ("dinov2_giant", torch.float16, 4), ("phikon_v1", torch.float16, 8),
                                          ("phikon_v1", torch.float32, 8), ("phikon_v2", torch.float16, 6),
                                          ("midnight", torch.float16, 4), ("h_optimus_0", torch.float16, 4),
                                          ("prov_gigapath", torch.float16, 4), ("lunit_vit_small_patch16_dino", torch.float16, 8),
                                          ("lunit_vit_small_patch8_dino", torch.float16, 4), ("pathorchestra", torch.float16, 6),
                                          ("clip_vit_b_32", torch.float16, 8), ("clip_vit_b_32", torch.float32, 8),
                                          ("clip_vit_b_16", torch.float16, 8), ("clip_vit_l_14", torch.float16, 6),
                                          ("clip_vit_l_14_336", torch.float16, 4), ("plip", torch.float16, 8),
                                          ("biomedclip", torch.float16, 8), ("virchow_v1", torch.float16, 3),
                                          ("virchow_v2", torch.float16, 3), ("h0_mini", torch.float16, 8),
                                          ("dinov3_vits16", torch.float16, 8), ("dinov3_vits16", torch.float32, 8),
                                          ("dinov3_vits16_plus", torch.float16, 8), ("dinov3_vitb16", torch.float16, 8),
                                          ("dinov3_vitl16", torch.float16, 6), ("dinov3_vith16_plus", torch.float16, 4)])
def test_encoder_at_real_size_vs_fp32_oracle(name, dtype, n):
    from atlaspatch_amd.encoders.vit import (ARCHS, IMAGENET_MEAN, IMAGENET_STD, TRANSFORM_NORM, TRANSFORM_RESIZE, build_hip_vit_extractor,
                                             random_canonical_state_dict)
    from oracle import vit_oracle
    torch.set_num_threads(min(32, torch.get_num_threads()))
    arch = dict(ARCHS[name])
    mean, std = TRANSFORM_NORM.get(name, (IMAGENET_MEAN, IMAGENET_STD))
    pool, act = arch.get("pool", "cls"), arch.get("act", "gelu")
    sd = _with_layer_scale(random_canonical_state_dict(arch, seed=41), arch, 42)
    ex = build_hip_vit_extractor(name=name, arch=arch, state_dict=sd, source="canonical", device=torch.device("cuda:0"), dtype=dtype,
                                 resize=TRANSFORM_RESIZE[name], expect_size=None, max_batch=64, mean=mean, std=std)
    tiles = _tiles(n, 43)
    got = ex.extract_batch(tiles, batch_size=32)
    got_f32s = None
    if dtype != torch.float32:
        ex.vit.set_option("f32_stream", True)
        got_f32s = ex.extract_batch(tiles, batch_size=32)
        ex.vit.set_option("f32_stream", False)
        ex.vit.set_option("full_last_block", True)
        got_full = ex.extract_batch(tiles, batch_size=32)
        assert _rel(got_full, got) <= 2e-3                          # the CLS-only tail and the full last block agree
        if pool == "cls_mean":
            assert np.array_equal(got_full, got)                    # (this pooling always runs the full last block)
    ex.cleanup()
    want = vit_oracle.canonical_extract(sd, tiles, heads=arch["heads"], depth=arch["depth"], image_size=arch["image_size"],
                                        resize=TRANSFORM_RESIZE[name], batch=2, eps=arch["ln_eps"], pool=pool, mean=mean, std=std, act=act)
    assert got.shape == want.shape == (n, arch.get("proj_dim") or arch["dim"] * (2 if pool == "cls_mean" else 1)) and got.dtype == np.float32
    _check(got, want, dtype, f"{name} L{arch['depth']}")
    if got_f32s is not None:
        _check(got_f32s, want, dtype, f"{name} L{arch['depth']}, f32_stream")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16])          # float32: the f32 attention kernel serves 64-wide heads only
def test_dinov3_vit7b16_width_at_reduced_depth(dtype):
    """dinov3.py:20-21 (dinov3_vit7b16, dinov3_vit7b16_sat): dim 4096, 32 heads of 128, gated MLP 8192, 4 register tokens, rotary
    embedding -- the real width at depth 3 (0.5 G parameters; the 40-block model is 6.7 G): rows wider than 2048 in the LayerNorm /
    stream kernels, K = 4096 / 8192 GEMMs, the 128-wide attention kernel without padding.  Against the fp32 oracle, whose rotary
    and gated-MLP forms are pinned against transformers' DINOv3ViTModel above."""
    from atlaspatch_amd.encoders.vit import ARCHS, TRANSFORM_RESIZE, build_hip_vit_extractor, random_canonical_state_dict
    from oracle import vit_oracle
    torch.set_num_threads(min(32, torch.get_num_threads()))
    name = "dinov3_vit7b16"
    arch = dict(ARCHS[name], depth=3)
    assert (ARCHS[name]["dim"], ARCHS[name]["depth"], ARCHS[name]["heads"], ARCHS[name]["mlp_dim"]) == (4096, 40, 32, 8192)
    sd = _with_layer_scale(random_canonical_state_dict(arch, seed=41), arch, 42)
    ex = build_hip_vit_extractor(name=name, arch=arch, state_dict=sd, source="canonical", device=torch.device("cuda:0"), dtype=dtype,
                                 resize=TRANSFORM_RESIZE[name], expect_size=None, max_batch=16)
    tiles = _tiles(3, 43)
    got = ex.extract_batch(tiles, batch_size=32)
    assert np.array_equal(got, ex.extract_batch(tiles, batch_size=2))
    ex.cleanup()
    want = vit_oracle.canonical_extract(sd, tiles, heads=arch["heads"], depth=arch["depth"], image_size=arch["image_size"],
                                        resize=TRANSFORM_RESIZE[name], batch=1, eps=arch["ln_eps"], pool="cls")
    assert got.shape == want.shape == (3, 4096)
    _check(got, want, dtype, "dinov3_vit7b16 width L3")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
def test_dinov2_with_registers_checkpoint_layout_on_the_device_vs_the_hf_model(dtype, tol):
    """The path a DINOv2-with-registers export takes (HF key names -> canonical -> device) against the HF model ITSELF: an
    independent implementation of register tokens + SwiGLU + LayerScale checks the device kernels directly (3 blocks, dim 384)."""
    from transformers import Dinov2WithRegistersConfig, Dinov2WithRegistersModel
    from atlaspatch_amd.encoders.vit import build_hip_vit_extractor
    from oracle import vit_oracle
    torch.manual_seed(0)
    cfg = Dinov2WithRegistersConfig(hidden_size=384, num_hidden_layers=3, num_attention_heads=6, mlp_ratio=4, use_swiglu_ffn=True,
                                    num_register_tokens=8, patch_size=14, image_size=224, layerscale_value=1e-5, layer_norm_eps=1e-6)
    model = _seed_params(Dinov2WithRegistersModel(cfg).eval())
    assert model.state_dict()["encoder.layer.0.mlp.weights_out.weight"].shape == (384, 1024)
    arch = dict(image_size=224, patch_size=14, dim=384, depth=3, heads=6, mlp_dim=1024, ln_eps=1e-6, layer_scale=True,
                reg_tokens=8, no_embed_class=True, mlp="swiglu")
    ex = build_hip_vit_extractor(name="dinov2_small", arch=arch, state_dict=dict(model.state_dict()), device=torch.device("cuda:0"),
                                 dtype=dtype, resize=(224, "bicubic"), expect_size=None, max_batch=64)
    tiles = _tiles(5, 47)
    got = ex.extract_batch(tiles, batch_size=32)
    ex.cleanup()
    x = vit_oracle.transform_resize_crop(tiles, resize=(224, "bicubic"), crop=224)
    with torch.inference_mode():
        want = model(pixel_values=x).pooler_output.numpy()
    assert got.shape == want.shape == (5, 384)
    assert _rel(got, want) <= tol, _rel(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
def test_dinov2_checkpoint_with_a_518px_position_grid_on_the_device_vs_the_hf_model(dtype, tol):
    """The registered ``dinov2_small`` shape fed an HF Dinov2Model state dict whose position embedding is the checkpoints' 37 x 37
    grid: the adapter resamples it to 16 x 16, the device result is compared with the HF model ITSELF at 224 px (which resamples
    in its forward) -- the module the reference calls (dinov2.py:46-60), 4 blocks."""
    from transformers import Dinov2Config, Dinov2Model
    from atlaspatch_amd.encoders.vit import ARCHS, TRANSFORM_RESIZE, build_hip_vit_extractor
    from oracle import vit_oracle
    torch.manual_seed(0)
    cfg = Dinov2Config(hidden_size=384, num_hidden_layers=4, num_attention_heads=6, mlp_ratio=4, patch_size=14, image_size=518,
                       layerscale_value=1.0, layer_norm_eps=1e-6)
    model = _seed_params(Dinov2Model(cfg).eval())
    assert model.state_dict()["embeddings.position_embeddings"].shape == (1, 1 + 37 * 37, 384)
    arch = dict(ARCHS["dinov2_small"], depth=4)
    ex = build_hip_vit_extractor(name="dinov2_small", arch=arch, state_dict=dict(model.state_dict()), device=torch.device("cuda:0"),
                                 dtype=dtype, resize=TRANSFORM_RESIZE["dinov2_small"], expect_size=None, max_batch=64)
    tiles = _tiles(5, 53)
    got = ex.extract_batch(tiles, batch_size=32)
    ex.cleanup()
    x = vit_oracle.transform_resize_crop(tiles, resize=TRANSFORM_RESIZE["dinov2_small"], crop=224)
    with torch.inference_mode():
        want = model(pixel_values=x).last_hidden_state[:, 0].numpy()
    assert got.shape == want.shape == (5, 384)
    assert _rel(got, want) <= tol, _rel(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
def test_class_token_plus_mean_patch_token_on_the_device_vs_the_hf_model(dtype, tol):
    """midnight.py:56-61 on an HF Dinov2Model (SwiGLU, LayerScale, 37 x 37 position grid), 4 blocks of dim 384: the device's
    AP_POOL_CLS_MEAN output against torch.cat([cls, patch_tokens.mean(1)], -1) of the HF model itself."""
    from transformers import Dinov2Config, Dinov2Model
    from atlaspatch_amd.encoders.vit import build_hip_vit_extractor
    from oracle import vit_oracle
    torch.manual_seed(0)
    cfg = Dinov2Config(hidden_size=384, num_hidden_layers=4, num_attention_heads=6, mlp_ratio=4, use_swiglu_ffn=True, patch_size=14,
                       image_size=518, layerscale_value=1.0, layer_norm_eps=1e-6)
    model = _seed_params(Dinov2Model(cfg).eval())
    arch = dict(image_size=224, patch_size=14, dim=384, depth=4, heads=6, mlp_dim=1024, ln_eps=1e-6, layer_scale=True,
                no_embed_class=True, mlp="swiglu", pool="cls_mean")
    ex = build_hip_vit_extractor(name="midnight", arch=arch, state_dict=dict(model.state_dict()), device=torch.device("cuda:0"),
                                 dtype=dtype, resize=(224, "bilinear"), expect_size=None, max_batch=64, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5))
    assert ex.embedding_dim == 768
    tiles = _tiles(5, 59)
    got = ex.extract_batch(tiles, batch_size=32)
    ex.cleanup()
    x = vit_oracle.transform_resize_crop(tiles, resize=(224, "bilinear"), crop=224, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5))
    with torch.inference_mode():
        lhs = model(x).last_hidden_state
        want = torch.cat([lhs[:, 0, :], lhs[:, 1:, :].mean(1)], dim=-1).numpy()
    assert got.shape == want.shape == (5, 768)
    assert _rel(got, want) <= tol, _rel(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
def test_clip_tower_on_the_device_vs_hf_get_image_features(dtype, tol):
    """plip.py:56 on the device: an HF CLIPModel state dict (ViT-B/32 shape at dim 384, 4 blocks) through the hf_clip adapter,
    ln_pre + QuickGELU epilogue + projection, against CLIPModel.get_image_features itself."""
    from atlaspatch_amd.encoders.vit import OPENAI_CLIP_MEAN, OPENAI_CLIP_STD, build_hip_vit_extractor
    from oracle import vit_oracle
    model = _hf_clip(hidden=384, layers=4, heads=6, image=224, patch=32, proj=256)
    arch = dict(image_size=224, patch_size=32, dim=384, depth=4, heads=6, mlp_dim=1536, ln_eps=1e-5, layer_scale=False,
                pre_norm=True, act="quick_gelu", proj_dim=256)
    ex = build_hip_vit_extractor(name="plip", arch=arch, state_dict=dict(model.state_dict()), device=torch.device("cuda:0"), dtype=dtype,
                                 resize=(224, "bicubic"), expect_size=None, max_batch=64, mean=OPENAI_CLIP_MEAN, std=OPENAI_CLIP_STD)
    assert ex.embedding_dim == 256
    tiles = _tiles(6, 61)
    got = ex.extract_batch(tiles, batch_size=32)
    if dtype != torch.float32:
        ex.vit.set_option("f32_stream", True)
        got2 = ex.extract_batch(tiles, batch_size=32)
    ex.cleanup()
    x = vit_oracle.transform_resize_crop(tiles, resize=(224, "bicubic"), crop=224, mean=OPENAI_CLIP_MEAN, std=OPENAI_CLIP_STD)
    want = _clip_image_features(model, x)
    assert got.shape == want.shape == (6, 256)
    assert _rel(got, want) <= tol, _rel(got, want)
    if dtype != torch.float32:
        assert _rel(got2, want) <= tol, _rel(got2, want)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("gated", [False, True])
def test_dinov3_on_the_device_vs_the_hf_model(dtype, tol, gated):
    """dinov3.py:63-67 on the device: an HF DINOv3ViTModel state dict (dim 384, 4 blocks, 4 register tokens) through the hf_dinov3
    adapter -- rotary embedding applied in place after the qkv GEMM -- against the HF model's pooler_output, both dataflows and
    the full last block."""
    from atlaspatch_amd.encoders.vit import build_hip_vit_extractor
    from oracle import vit_oracle
    model = _hf_dinov3(hidden=384, layers=4, heads=6, inter=1536 if not gated else 1024, image=224, gated=gated)
    arch = dict(image_size=224, patch_size=16, dim=384, depth=4, heads=6, mlp_dim=1536 if not gated else 1024, ln_eps=1e-5, layer_scale=True,
                reg_tokens=4, no_embed_class=True, rope=True, mlp="swiglu" if gated else None)
    ex = build_hip_vit_extractor(name="dinov3_vits16", arch=arch, state_dict=dict(model.state_dict()), device=torch.device("cuda:0"),
                                 dtype=dtype, resize=(224, "bilinear"), expect_size=None, max_batch=64)
    tiles = _tiles(6, 67)
    got = ex.extract_batch(tiles, batch_size=32)
    ex.vit.set_option("full_last_block", True)
    got_full = ex.extract_batch(tiles, batch_size=32)
    got2 = None
    if dtype != torch.float32:
        ex.vit.set_option("full_last_block", False)
        ex.vit.set_option("f32_stream", True)
        got2 = ex.extract_batch(tiles, batch_size=32)
    ex.cleanup()
    x = vit_oracle.transform_resize_crop(tiles, resize=(224, "bilinear"), crop=224)
    with torch.inference_mode():
        want = model(pixel_values=x).pooler_output.numpy()
    assert got.shape == want.shape == (6, 384)
    assert _rel(got, want) <= tol and _rel(got_full, want) <= tol, (_rel(got, want), _rel(got_full, want))
    if got2 is not None:
        assert _rel(got2, want) <= tol, _rel(got2, want)
