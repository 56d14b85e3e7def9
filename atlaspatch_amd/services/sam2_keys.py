"""State-dict key map between the two public layouts of SAM2(.1) Hiera image-path weights.

* "facebook" layout: the ``sam2`` package's ``SAM2Base.state_dict()`` — what the reference loads with
  ``predictor.model.load_state_dict(checkpoint["model"], strict=True)`` (/root/reference services/segmentation.py:64-68)
  and what ``Sam2HipPredictor`` reads.
* "hf" layout: ``transformers.models.sam2.Sam2Model.state_dict()`` (transformers >= 4.56).

The map is verified in both directions by ``tests/golden/gen_golden_hf_sam2.py``: the HF model, loaded through
``facebook_to_hf`` with the weights of a facebook-layout state dict, reproduces the restated forward
(``oracle/sam2_oracle.py``) and the device path — a wrong pair would change the outputs.  It also lets
``load_sam2_state_dict`` accept an HF checkpoint (``model.safetensors`` of a ``sam2.1_hiera_tiny`` export).

Only tensors of the image path (Hiera trunk, FPN neck, prompt encoder, mask decoder, ``no_mem_embed``) are mapped; the
video-memory modules of the facebook checkpoint (``memory_attention.*``, ``memory_encoder.*``, ``obj_ptr_*``,
``maskmem_tpos_enc`` …) have no counterpart in ``Sam2Model`` and are dropped.
"""
from __future__ import annotations

import re

# (facebook regex, hf replacement) — applied in order, first match wins.  Every pattern is anchored.
_RULES = [
    (r"image_encoder\.trunk\.patch_embed\.proj\.(weight|bias)", r"vision_encoder.backbone.patch_embed.projection.\1"),
    (r"image_encoder\.trunk\.(pos_embed|pos_embed_window)", r"vision_encoder.backbone.\1"),
    (r"image_encoder\.trunk\.blocks\.(\d+)\.norm([12])\.(weight|bias)", r"vision_encoder.backbone.blocks.\1.layer_norm\2.\3"),
    (r"image_encoder\.trunk\.blocks\.(\d+)\.mlp\.layers\.0\.(weight|bias)", r"vision_encoder.backbone.blocks.\1.mlp.proj_in.\2"),
    (r"image_encoder\.trunk\.blocks\.(\d+)\.mlp\.layers\.1\.(weight|bias)", r"vision_encoder.backbone.blocks.\1.mlp.proj_out.\2"),
    (r"image_encoder\.trunk\.blocks\.(\d+)\.(attn\.qkv|attn\.proj|proj)\.(weight|bias)", r"vision_encoder.backbone.blocks.\1.\2.\3"),
    (r"image_encoder\.neck\.convs\.(\d+)\.conv\.(weight|bias)", r"vision_encoder.neck.convs.\1.\2"),
    (r"no_mem_embed", r"no_memory_embedding"),
    (r"sam_prompt_encoder\.(no_mask_embed|not_a_point_embed)\.weight", r"prompt_encoder.\1.weight"),
    (r"sam_prompt_encoder\.mask_downscaling\.0\.(weight|bias)", r"prompt_encoder.mask_embed.conv1.\1"),
    (r"sam_prompt_encoder\.mask_downscaling\.1\.(weight|bias)", r"prompt_encoder.mask_embed.layer_norm1.\1"),
    (r"sam_prompt_encoder\.mask_downscaling\.3\.(weight|bias)", r"prompt_encoder.mask_embed.conv2.\1"),
    (r"sam_prompt_encoder\.mask_downscaling\.4\.(weight|bias)", r"prompt_encoder.mask_embed.layer_norm2.\1"),
    (r"sam_prompt_encoder\.mask_downscaling\.6\.(weight|bias)", r"prompt_encoder.mask_embed.conv3.\1"),
    (r"sam_mask_decoder\.transformer\.layers\.(\d+)\.norm([1-4])\.(weight|bias)", r"mask_decoder.transformer.layers.\1.layer_norm\2.\3"),
    (r"sam_mask_decoder\.transformer\.layers\.(\d+)\.mlp\.layers\.0\.(weight|bias)", r"mask_decoder.transformer.layers.\1.mlp.proj_in.\2"),
    (r"sam_mask_decoder\.transformer\.layers\.(\d+)\.mlp\.layers\.1\.(weight|bias)", r"mask_decoder.transformer.layers.\1.mlp.proj_out.\2"),
    (r"sam_mask_decoder\.transformer\.layers\.(\d+)\.(self_attn|cross_attn_token_to_image|cross_attn_image_to_token)\.out_proj\.(weight|bias)",
     r"mask_decoder.transformer.layers.\1.\2.o_proj.\3"),
    (r"sam_mask_decoder\.transformer\.layers\.(\d+)\.(self_attn|cross_attn_token_to_image|cross_attn_image_to_token)\.([qkv]_proj)\.(weight|bias)",
     r"mask_decoder.transformer.layers.\1.\2.\3.\4"),
    (r"sam_mask_decoder\.transformer\.final_attn_token_to_image\.out_proj\.(weight|bias)", r"mask_decoder.transformer.final_attn_token_to_image.o_proj.\1"),
    (r"sam_mask_decoder\.transformer\.final_attn_token_to_image\.([qkv]_proj)\.(weight|bias)", r"mask_decoder.transformer.final_attn_token_to_image.\1.\2"),
    (r"sam_mask_decoder\.transformer\.norm_final_attn\.(weight|bias)", r"mask_decoder.transformer.layer_norm_final_attn.\1"),
    (r"sam_mask_decoder\.(iou_token|mask_tokens|obj_score_token)\.weight", r"mask_decoder.\1.weight"),
    (r"sam_mask_decoder\.(conv_s0|conv_s1)\.(weight|bias)", r"mask_decoder.\1.\2"),
    (r"sam_mask_decoder\.output_upscaling\.0\.(weight|bias)", r"mask_decoder.upscale_conv1.\1"),
    (r"sam_mask_decoder\.output_upscaling\.1\.(weight|bias)", r"mask_decoder.upscale_layer_norm.\1"),
    (r"sam_mask_decoder\.output_upscaling\.3\.(weight|bias)", r"mask_decoder.upscale_conv2.\1"),
    # three-layer MLP heads: layers.0 / layers.1 / layers.2  ->  proj_in / layers.0 / proj_out
    (r"sam_mask_decoder\.(output_hypernetworks_mlps\.\d+|iou_prediction_head|pred_obj_score_head)\.layers\.0\.(weight|bias)", r"mask_decoder.\1.proj_in.\2"),
    (r"sam_mask_decoder\.(output_hypernetworks_mlps\.\d+|iou_prediction_head|pred_obj_score_head)\.layers\.1\.(weight|bias)", r"mask_decoder.\1.layers.0.\2"),
    (r"sam_mask_decoder\.(output_hypernetworks_mlps\.\d+|iou_prediction_head|pred_obj_score_head)\.layers\.2\.(weight|bias)", r"mask_decoder.\1.proj_out.\2"),
]
_COMPILED = [(re.compile("^" + a + "$"), b) for a, b in _RULES]

_GAUSS = "sam_prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"
_HF_GAUSS = ("prompt_encoder.shared_embedding.positional_embedding", "shared_image_embedding.positional_embedding")
_POINT = re.compile(r"^sam_prompt_encoder\.point_embeddings\.(\d)\.weight$")
_HF_POINT = "prompt_encoder.point_embed.weight"


def facebook_key_to_hf(key: str):
    """One facebook-layout name -> its HF name, or None for tensors ``Sam2Model`` has no slot for.  The two names that
    do not map 1:1 (the Gaussian matrix, which HF stores twice, and the four ``point_embeddings.{i}`` rows, which HF
    stacks) are handled by ``facebook_to_hf`` / ``hf_to_facebook``."""
    for rx, repl in _COMPILED:
        if rx.match(key):
            return rx.sub(repl, key)
    return None


def facebook_to_hf(sd: dict) -> dict:
    import torch
    out = {}
    points = {}
    for k, v in sd.items():
        if k == _GAUSS:
            for name in _HF_GAUSS:
                out[name] = v
            continue
        m = _POINT.match(k)
        if m:
            points[int(m.group(1))] = v.reshape(1, -1)
            continue
        hk = facebook_key_to_hf(k)
        if hk is not None:
            out[hk] = v
    if points:
        if sorted(points) != [0, 1, 2, 3]:
            raise KeyError(f"sam_prompt_encoder.point_embeddings: expected rows 0..3, found {sorted(points)}")
        out[_HF_POINT] = torch.cat([points[i] for i in range(4)], 0)
    return out


def hf_to_facebook(sd: dict) -> dict:
    """Inverse of ``facebook_to_hf`` over the image path (built by inverting the forward map on the facebook names the
    image path can contain, so the two directions cannot drift apart)."""
    inverse = {}
    for fk in _facebook_image_path_names():
        hk = facebook_key_to_hf(fk)
        if hk is not None:
            inverse[hk] = fk
    out = {}
    for k, v in sd.items():
        if k == _HF_GAUSS[0]:
            out[_GAUSS] = v
        elif k == _HF_GAUSS[1]:
            if _HF_GAUSS[0] not in sd:
                out[_GAUSS] = v
        elif k == _HF_POINT:
            for i in range(4):
                out[f"sam_prompt_encoder.point_embeddings.{i}.weight"] = v[i:i + 1]
        elif k in inverse:
            out[inverse[k]] = v
    return out


def is_hf_layout(sd: dict) -> bool:
    return any(k.startswith("vision_encoder.backbone.") for k in sd)


def _facebook_image_path_names(depth: int = 64) -> list:
    """Every facebook-layout name the rules can produce for block / layer indices below ``depth`` (a superset of any
    Hiera size: T has 12 blocks, L 48)."""
    names = ["no_mem_embed"]
    t = "image_encoder.trunk."
    names += [t + "patch_embed.proj.weight", t + "patch_embed.proj.bias", t + "pos_embed", t + "pos_embed_window"]
    wb = ("weight", "bias")
    for i in range(depth):
        b = f"{t}blocks.{i}."
        for mod in ("norm1", "norm2", "attn.qkv", "attn.proj", "mlp.layers.0", "mlp.layers.1", "proj"):
            names += [f"{b}{mod}.{w}" for w in wb]
    for n in range(4):
        names += [f"image_encoder.neck.convs.{n}.conv.{w}" for w in wb]
    p = "sam_prompt_encoder."
    names += [p + "no_mask_embed.weight", p + "not_a_point_embed.weight"]
    names += [f"{p}mask_downscaling.{i}.{w}" for i in (0, 1, 3, 4, 6) for w in wb]
    d = "sam_mask_decoder."
    attn = ("self_attn", "cross_attn_token_to_image", "cross_attn_image_to_token")
    for l in range(2):
        b = f"{d}transformer.layers.{l}."
        names += [f"{b}{a}.{pj}.{w}" for a in attn for pj in ("q_proj", "k_proj", "v_proj", "out_proj") for w in wb]
        names += [f"{b}norm{k}.{w}" for k in range(1, 5) for w in wb]
        names += [f"{b}mlp.layers.{k}.{w}" for k in range(2) for w in wb]
    names += [f"{d}transformer.final_attn_token_to_image.{pj}.{w}" for pj in ("q_proj", "k_proj", "v_proj", "out_proj") for w in wb]
    names += [f"{d}transformer.norm_final_attn.{w}" for w in wb]
    names += [d + "iou_token.weight", d + "mask_tokens.weight", d + "obj_score_token.weight"]
    names += [f"{d}{c}.{w}" for c in ("conv_s0", "conv_s1") for w in wb]
    names += [f"{d}output_upscaling.{i}.{w}" for i in (0, 1, 3) for w in wb]
    heads = [f"output_hypernetworks_mlps.{i}" for i in range(4)] + ["iou_prediction_head", "pred_obj_score_head"]
    names += [f"{d}{h}.layers.{k}.{w}" for h in heads for k in range(3) for w in wb]
    return names
