"""ViT-family encoders on the native HIP path.

* ``ARCHS``: the shapes the reference registers as ``vit_b_16`` / ``vit_l_16``
  (/root/reference/atlas_patch/models/patch/vit.py:9-15, torchvision) and ``uni_v1``
  (models/patch/uni.py:13-60, timm ViT-L/16 with LayerScale).
* ``canonical_state_dict``: adapters from torchvision / timm / HF ``ViTModel`` key names to the
  parameter names of ``ap_vit_set_param``.  QKV is packed ``[q; k; v]`` rows.
* ``HipViT``: owns the ``ap_vit`` handle and its HBM workspace; ``forward_u8`` / ``forward_chw``
  launch on torch's current stream.
"""
from __future__ import annotations

import ctypes as C
import re
import logging
import os
from pathlib import Path
from typing import Optional

import numpy as np
import torch

from .. import _lib
from .base import HipViTFeatureExtractor

logger = logging.getLogger("atlaspatch_amd.encoders.vit")

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)

# Leading Resize of the reference transform per registered name: (shorter side, Pillow filter).
# torchvision (models/patch/base.py:126-145 prefers <enum>.IMAGENET1K_V1, base.py:170 takes weights.transforms()) [3P]:
#   ViT_B_16_Weights.IMAGENET1K_V1 = ImageClassification(crop_size=224)                  -> resize_size 256 (default), bilinear
#   ViT_L_16_Weights.IMAGENET1K_V1 = ImageClassification(crop_size=224, resize_size=242) -> 256-px tiles ARE resampled to 242
# timm / open_clip: uni_v1 Resize(224, bicubic), conch_v1 Resize(448, bicubic).
#   ViT_B_32 / ViT_L_32 IMAGENET1K_V1 = ImageClassification(crop_size=224)               -> resize 256, bilinear
#   ViT_H_14 has no IMAGENET1K_V1: base.py:131-137 falls back to DEFAULT = IMAGENET1K_SWAG_E2E_V1 =
#     ImageClassification(crop_size=518, resize_size=518, interpolation=BICUBIC), model image_size 518 (1370 tokens)
# uni_v2 (uni.py:62-125): timm create_transform of the hub config = Resize(224, bicubic) + CenterCrop(224) (unverifiable offline)
TRANSFORM_RESIZE = {
    "vit_b_16": (256, "bilinear"),
    "vit_b_32": (256, "bilinear"),
    "vit_l_16": (242, "bilinear"),
    "vit_l_32": (256, "bilinear"),
    "vit_h_14": (518, "bicubic"),
    "uni_v1": (224, "bicubic"),
    "uni_v2": (224, "bicubic"),
    "conch_v1": (448, "bicubic"),
    # transformers image processors (dinov2.py:20-25, phikon.py:16-21: AutoImageProcessor(use_fast=True), configs from the public
    # model cards, unverifiable offline): facebook/dinov2-* = BitImageProcessor(shortest_edge 256, bicubic, crop 224);
    # owkin/phikon = ViTImageProcessor(224 x 224, bilinear, no crop); owkin/phikon-v2 = BitImageProcessor(224, bicubic, crop 224).
    # Deviation, stated: the fast processors resample with torchvision's tensor kernels (antialias on), this path with the
    # Pillow-exact device resize; on the default 256-px tiles the dinov2 resize is a no-op.
    "dinov2_small": (256, "bicubic"), "dinov2_base": (256, "bicubic"), "dinov2_large": (256, "bicubic"),
    "dinov2_giant": (256, "bicubic"),
    "phikon_v1": (224, "bilinear"), "phikon_v2": (224, "bicubic"),
    # torchvision transforms written out in the loader files (PIL input -> Pillow filters):
    #   midnight.py:14-24 Resize(224) + CenterCrop(224), Normalize(0.5, 0.5); hoptimus.py:15-30 Resize((224, 224)) + its own
    #   mean / std; gigapath.py:15-26 Resize(256, BICUBIC) + CenterCrop(224); pathorchestra.py:52-58 Resize(224)
    "midnight": (224, "bilinear"), "h_optimus_0": (224, "bilinear"), "h_optimus_1": (224, "bilinear"),
    # virchow.py:14-19: timm create_transform of the hub config (expected: Resize(224, bicubic) + CenterCrop(224), ImageNet
    # mean / std; unverifiable offline)
    "virchow_v1": (224, "bicubic"), "virchow_v2": (224, "bicubic"), "h0_mini": (224, "bicubic"),
    "prov_gigapath": (256, "bicubic"), "pathorchestra": (224, "bilinear"),
    # lunit.py:58-59: timm create_transform of the hub data config (expected: Resize(256, bicubic) + CenterCrop(224), the
    # checkpoints' own mean / std; unverifiable offline)
    "lunit_vit_small_patch16_dino": (256, "bicubic"), "lunit_vit_small_patch8_dino": (256, "bicubic"),
    # open_clip image transform (clip.py:38-40) / transformers CLIPProcessor (plip.py:35, quilt.py): Resize(S, bicubic) +
    # CenterCrop(S), OpenAI CLIP mean / std
    "clip_vit_b_32": (224, "bicubic"), "clip_vit_b_16": (224, "bicubic"), "clip_vit_l_14": (224, "bicubic"),
    "clip_vit_l_14_336": (336, "bicubic"), "plip": (224, "bicubic"), "quilt_b_32": (224, "bicubic"), "quilt_b_16": (224, "bicubic"),
    "biomedclip": (224, "bicubic"),
    # dinov3.py:52 AutoImageProcessor of facebook/dinov3-*: 224 x 224, bilinear, no crop (public model cards, unverifiable offline)
    "dinov3_vits16": (224, "bilinear"), "dinov3_vits16_plus": (224, "bilinear"), "dinov3_vitb16": (224, "bilinear"),
    "dinov3_vitl16": (224, "bilinear"), "dinov3_vitl16_sat": (224, "bilinear"), "dinov3_vith16_plus": (224, "bilinear"),
    "dinov3_vit7b16": (224, "bilinear"), "dinov3_vit7b16_sat": (224, "bilinear"),
}

# Normalize() constants per registered name (default: ImageNet)
TRANSFORM_NORM = {
    "conch_v1": (OPENAI_CLIP_MEAN, OPENAI_CLIP_STD),
    "clip_vit_b_32": (OPENAI_CLIP_MEAN, OPENAI_CLIP_STD), "clip_vit_b_16": (OPENAI_CLIP_MEAN, OPENAI_CLIP_STD),
    "clip_vit_l_14": (OPENAI_CLIP_MEAN, OPENAI_CLIP_STD), "clip_vit_l_14_336": (OPENAI_CLIP_MEAN, OPENAI_CLIP_STD),
    "plip": (OPENAI_CLIP_MEAN, OPENAI_CLIP_STD), "quilt_b_32": (OPENAI_CLIP_MEAN, OPENAI_CLIP_STD),
    "quilt_b_16": (OPENAI_CLIP_MEAN, OPENAI_CLIP_STD), "biomedclip": (OPENAI_CLIP_MEAN, OPENAI_CLIP_STD),
    "dinov3_vitl16_sat": ((0.430, 0.411, 0.296), (0.213, 0.156, 0.143)),        # the satellite (SAT-493M) checkpoints' statistics
    "dinov3_vit7b16_sat": ((0.430, 0.411, 0.296), (0.213, 0.156, 0.143)),       # (their AutoImageProcessor config, dinov3.py:39-41)
    "midnight": ((0.5, 0.5, 0.5), (0.5, 0.5, 0.5)),                                                   # midnight.py:22
    "h_optimus_0": ((0.707223, 0.578729, 0.703617), (0.211883, 0.230117, 0.177517)),                  # hoptimus.py:24-27
    "h_optimus_1": ((0.707223, 0.578729, 0.703617), (0.211883, 0.230117, 0.177517)),
    "h0_mini": ((0.707223, 0.578729, 0.703617), (0.211883, 0.230117, 0.177517)),          # the hub config's (H-optimus statistics)
    "lunit_vit_small_patch16_dino": ((0.70322989, 0.53606487, 0.66096631), (0.21716536, 0.26081574, 0.20723464)),
    "lunit_vit_small_patch8_dino": ((0.70322989, 0.53606487, 0.66096631), (0.21716536, 0.26081574, 0.20723464)),
}

ARCHS = {
    # name: image, patch, dim, depth, heads, mlp, eps, layer_scale
    "vit_b_16": dict(image_size=224, patch_size=16, dim=768, depth=12, heads=12, mlp_dim=3072,
                     ln_eps=1e-6, layer_scale=False),
    "vit_l_16": dict(image_size=224, patch_size=16, dim=1024, depth=24, heads=16, mlp_dim=4096,
                     ln_eps=1e-6, layer_scale=False),
    "uni_v1": dict(image_size=224, patch_size=16, dim=1024, depth=24, heads=16, mlp_dim=4096,
                   ln_eps=1e-6, layer_scale=True),
    # the rest of models/patch/vit.py:9-15 (torchvision): patch 32 (50 tokens); ViT-H/14 at the SWAG end-to-end weights'
    # 518 px (37 x 37 + 1 = 1370 tokens), heads 80 wide (stored zero-padded to 96 on the device, softmax scale 1/sqrt(80))
    "vit_b_32": dict(image_size=224, patch_size=32, dim=768, depth=12, heads=12, mlp_dim=3072,
                     ln_eps=1e-6, layer_scale=False),
    "vit_l_32": dict(image_size=224, patch_size=32, dim=1024, depth=24, heads=16, mlp_dim=4096,
                     ln_eps=1e-6, layer_scale=False),
    "vit_h_14": dict(image_size=518, patch_size=14, dim=1280, depth=32, heads=16, mlp_dim=5120,
                     ln_eps=1e-6, layer_scale=False),
    # UNI2-h (models/patch/uni.py:62-125, timm kwargs :82-96): ViT-H/14 at 224 px, 8 register tokens, no_embed_class (the
    # position embedding covers the 256 patch tokens only), SwiGLUPacked MLP (fc1 1536 -> 2 x 4096, silu(x1) * x2, fc2
    # 4096 -> 1536; mlp_ratio 2.66667 * 2 -> int(1536 * 5.33334) = 8192 packed), LayerScale 1e-5, class-token pooling
    "uni_v2": dict(image_size=224, patch_size=14, dim=1536, depth=24, heads=24, mlp_dim=4096,
                   ln_eps=1e-6, layer_scale=True, reg_tokens=8, no_embed_class=True, mlp="swiglu"),
    # CONCH v1 visual tower (models/patch/conch.py:20-64 -> conch.open_clip_custom "conch_ViT-B-16" [3P, package
    # absent offline]): timm ViT-B/16 trunk at 448 px (785 tokens) + open_clip AttentionalPooler with ONE
    # contrastive query (d_model 512, 8 heads, context 768) + LayerNorm; encode_image(proj_contrast=False,
    # normalize=False) returns that 512-vector.
    "conch_v1": dict(image_size=448, patch_size=16, dim=768, depth=12, heads=12, mlp_dim=3072,
                     ln_eps=1e-6, layer_scale=False, pool="attn", pool_dim=512, pool_heads=8, pool_ln_eps=1e-5),
    # transformers Dinov2Model (models/patch/dinov2.py:12-17,46-60: AutoModel.from_pretrained(facebook/dinov2-*),
    # last_hidden_state[:, 0]): patch 14 at the processor's 224-px crop = 256 + 1 tokens, LayerScale, LayerNorm 1e-6; the
    # checkpoints hold a 37 x 37 position grid (518 px) that the model resamples to the input's 16 x 16 with torch's bicubic
    # interpolate in every forward (Dinov2Embeddings.interpolate_pos_encoding) -- done once at load here, with the same call.
    # Canonical form: no_embed_class with the class position row folded into the class token (the hf_dinov2 adapter).
    # giant: SwiGLUFFN, hidden = (int(1536 * 4 * 2 / 3) + 7) // 8 * 8 = 4096, silu(x1) * x2 on the two halves of weights_in
    "dinov2_small": dict(image_size=224, patch_size=14, dim=384, depth=12, heads=6, mlp_dim=1536, ln_eps=1e-6,
                         layer_scale=True, no_embed_class=True),
    "dinov2_base": dict(image_size=224, patch_size=14, dim=768, depth=12, heads=12, mlp_dim=3072, ln_eps=1e-6,
                        layer_scale=True, no_embed_class=True),
    "dinov2_large": dict(image_size=224, patch_size=14, dim=1024, depth=24, heads=16, mlp_dim=4096, ln_eps=1e-6,
                         layer_scale=True, no_embed_class=True),
    "dinov2_giant": dict(image_size=224, patch_size=14, dim=1536, depth=40, heads=24, mlp_dim=4096, ln_eps=1e-6,
                         layer_scale=True, no_embed_class=True, mlp="swiglu"),
    # models/patch/phikon.py: phikon_v1 = transformers ViTModel(owkin/phikon, add_pooling_layer=False) = ViT-B/16, LayerNorm
    # 1e-12 (ViTConfig default), last_hidden_state[:, 0] (after the final LayerNorm); phikon_v2 = AutoModel(owkin/phikon-v2) =
    # Dinov2Model ViT-L/16 at 224 px (197 tokens)
    "phikon_v1": dict(image_size=224, patch_size=16, dim=768, depth=12, heads=12, mlp_dim=3072, ln_eps=1e-12, layer_scale=False),
    "phikon_v2": dict(image_size=224, patch_size=16, dim=1024, depth=24, heads=16, mlp_dim=4096, ln_eps=1e-6,
                      layer_scale=True, no_embed_class=True),
    # models/patch/midnight.py: transformers AutoModel(kaiko-ai/midnight) = Dinov2Model ViT-g/14 (the loader's emb_dim 3072 = 2 x
    # 1536), features = cat(last_hidden_state[:, 0], last_hidden_state[:, 1:].mean(1))
    "midnight": dict(image_size=224, patch_size=14, dim=1536, depth=40, heads=24, mlp_dim=4096, ln_eps=1e-6,
                     layer_scale=True, no_embed_class=True, mlp="swiglu", pool="cls_mean"),
    # timm hub models; the architecture comes from the hub's config.json, not from the reference's text (public model cards,
    # unverifiable offline -- same standing as uni_v1 / uni_v2).  hoptimus.py:53-58 (init_values 1e-5): H-optimus-0 / -1 =
    # vit_giant_patch14_reg4_dinov2 at 224 px (1536 / 40 / 24, SwiGLUPacked 4096, 4 register tokens, no_embed_class);
    # gigapath.py:46: vit_giant_patch14_dinov2 with patch 16 (1536 / 40 / 24, SwiGLUPacked 4096, class position row);
    # lunit.py:10-16: timm vit_small patch 16 / 8 (384 / 12 / 6); pathorchestra.py:38-43: ViT-L/16 with LayerScale
    "h_optimus_0": dict(image_size=224, patch_size=14, dim=1536, depth=40, heads=24, mlp_dim=4096, ln_eps=1e-6,
                        layer_scale=True, reg_tokens=4, no_embed_class=True, mlp="swiglu"),
    "h_optimus_1": dict(image_size=224, patch_size=14, dim=1536, depth=40, heads=24, mlp_dim=4096, ln_eps=1e-6,
                        layer_scale=True, reg_tokens=4, no_embed_class=True, mlp="swiglu"),
    "prov_gigapath": dict(image_size=224, patch_size=16, dim=1536, depth=40, heads=24, mlp_dim=4096, ln_eps=1e-6,
                          layer_scale=True, mlp="swiglu"),
    # virchow.py:41-46,94-99 (mlp_layer=SwiGLUPacked, act_layer=SiLU; hub config: ViT-H/14, 1280 / 32 / 16 -> 80-wide heads,
    # mlp_ratio 5.3375 -> packed 6832 = 2 x 3416, init_values 1e-5; Virchow2: 4 register tokens).  Features: class token |
    # mean of the patch tokens (virchow.py:58-61; Virchow2 skips its registers, :111-114) = 2560-d
    # hoptimus.py:141-146,158-161: H0-mini = vit_base_patch14_reg4_dinov2 with SwiGLUPacked (768 / 12 / 12, packed 3072 = 2 x
    # 1536, 4 register tokens, no_embed_class), features = class token | mean of the patch tokens (1536-d)
    "h0_mini": dict(image_size=224, patch_size=14, dim=768, depth=12, heads=12, mlp_dim=1536, ln_eps=1e-6, layer_scale=True,
                    reg_tokens=4, no_embed_class=True, mlp="swiglu", pool="cls_mean"),
    "virchow_v1": dict(image_size=224, patch_size=14, dim=1280, depth=32, heads=16, mlp_dim=3416, ln_eps=1e-6, layer_scale=True,
                       mlp="swiglu", pool="cls_mean"),
    "virchow_v2": dict(image_size=224, patch_size=14, dim=1280, depth=32, heads=16, mlp_dim=3416, ln_eps=1e-6, layer_scale=True,
                       mlp="swiglu", pool="cls_mean", reg_tokens=4),
    "lunit_vit_small_patch16_dino": dict(image_size=224, patch_size=16, dim=384, depth=12, heads=6, mlp_dim=1536, ln_eps=1e-6,
                                         layer_scale=False),
    "lunit_vit_small_patch8_dino": dict(image_size=224, patch_size=8, dim=384, depth=12, heads=6, mlp_dim=1536, ln_eps=1e-6,
                                        layer_scale=False),
    "pathorchestra": dict(image_size=224, patch_size=16, dim=1024, depth=24, heads=16, mlp_dim=4096, ln_eps=1e-6,
                          layer_scale=True),
    # CLIP vision towers: models/patch/clip.py:16-19 (open_clip ViT-B-32 / ViT-B-16 / ViT-L-14 / ViT-L-14-336, "openai"),
    # plip.py (transformers CLIPModel vinid/plip = ViT-B/32), quilt.py (CLIPModel wisdomik/QuiltNet-B-32 / B-16).  No
    # patch-embedding bias, LayerNorm on the embedded tokens before the blocks (ln_pre), QuickGELU, LayerNorm 1e-5, final
    # LayerNorm of the class token times the bias-free visual projection = encode_image / get_image_features
    "clip_vit_b_32": dict(image_size=224, patch_size=32, dim=768, depth=12, heads=12, mlp_dim=3072, ln_eps=1e-5, layer_scale=False,
                          pre_norm=True, act="quick_gelu", proj_dim=512),
    "clip_vit_b_16": dict(image_size=224, patch_size=16, dim=768, depth=12, heads=12, mlp_dim=3072, ln_eps=1e-5, layer_scale=False,
                          pre_norm=True, act="quick_gelu", proj_dim=512),
    "clip_vit_l_14": dict(image_size=224, patch_size=14, dim=1024, depth=24, heads=16, mlp_dim=4096, ln_eps=1e-5, layer_scale=False,
                          pre_norm=True, act="quick_gelu", proj_dim=768),
    "clip_vit_l_14_336": dict(image_size=336, patch_size=14, dim=1024, depth=24, heads=16, mlp_dim=4096, ln_eps=1e-5,
                              layer_scale=False, pre_norm=True, act="quick_gelu", proj_dim=768),
    "plip": dict(image_size=224, patch_size=32, dim=768, depth=12, heads=12, mlp_dim=3072, ln_eps=1e-5, layer_scale=False,
                 pre_norm=True, act="quick_gelu", proj_dim=512),
    "quilt_b_32": dict(image_size=224, patch_size=32, dim=768, depth=12, heads=12, mlp_dim=3072, ln_eps=1e-5, layer_scale=False,
                       pre_norm=True, act="quick_gelu", proj_dim=512),
    "quilt_b_16": dict(image_size=224, patch_size=16, dim=768, depth=12, heads=12, mlp_dim=3072, ln_eps=1e-5, layer_scale=False,
                       pre_norm=True, act="quick_gelu", proj_dim=512),
    # biomedclip.py: open_clip TimmModel = timm vit_base_patch16_224 (erf GELU, LayerNorm 1e-6, class token) + linear projection
    # 768 -> 512 without bias; encode_image (not normalised)
    "biomedclip": dict(image_size=224, patch_size=16, dim=768, depth=12, heads=12, mlp_dim=3072, ln_eps=1e-6, layer_scale=False,
                       proj_dim=512),
    # models/patch/dinov3.py:12-21: transformers DINOv3ViTModel, pooler_output.  Class + 4 register tokens, no position embedding,
    # rotary embedding (theta 100) on q / k of the patch tokens, LayerScale, LayerNorm 1e-5; "plus" = gated MLP.  The two
    # dinov3_vit7b16 names (dinov3.py:20-21): dim 4096, 40 blocks, 32 heads of 128, gated MLP 8192 (6.7 G parameters, 13.4 GB in
    # float16; public hub config, unverifiable offline) -- the rows wider than 2048 take the LayerNorm kernels' 16-vector form.
    "dinov3_vits16": dict(image_size=224, patch_size=16, dim=384, depth=12, heads=6, mlp_dim=1536, ln_eps=1e-5, layer_scale=True,
                          reg_tokens=4, no_embed_class=True, rope=True),
    "dinov3_vits16_plus": dict(image_size=224, patch_size=16, dim=384, depth=12, heads=6, mlp_dim=1536, ln_eps=1e-5, layer_scale=True,
                               reg_tokens=4, no_embed_class=True, rope=True, mlp="swiglu"),
    "dinov3_vitb16": dict(image_size=224, patch_size=16, dim=768, depth=12, heads=12, mlp_dim=3072, ln_eps=1e-5, layer_scale=True,
                          reg_tokens=4, no_embed_class=True, rope=True),
    "dinov3_vitl16": dict(image_size=224, patch_size=16, dim=1024, depth=24, heads=16, mlp_dim=4096, ln_eps=1e-5, layer_scale=True,
                          reg_tokens=4, no_embed_class=True, rope=True),
    "dinov3_vitl16_sat": dict(image_size=224, patch_size=16, dim=1024, depth=24, heads=16, mlp_dim=4096, ln_eps=1e-5, layer_scale=True,
                              reg_tokens=4, no_embed_class=True, rope=True),
    "dinov3_vith16_plus": dict(image_size=224, patch_size=16, dim=1280, depth=32, heads=20, mlp_dim=5120, ln_eps=1e-5, layer_scale=True,
                               reg_tokens=4, no_embed_class=True, rope=True, mlp="swiglu"),
    "dinov3_vit7b16": dict(image_size=224, patch_size=16, dim=4096, depth=40, heads=32, mlp_dim=8192, ln_eps=1e-5, layer_scale=True,
                           reg_tokens=4, no_embed_class=True, rope=True, mlp="swiglu"),
    "dinov3_vit7b16_sat": dict(image_size=224, patch_size=16, dim=4096, depth=40, heads=32, mlp_dim=8192, ln_eps=1e-5, layer_scale=True,
                               reg_tokens=4, no_embed_class=True, rope=True, mlp="swiglu"),
}


# ----------------------------------------------------------------------------- adapters
def _detect_source(sd: dict) -> str:
    keys = sd.keys()
    if "conv_proj.weight" in keys:
        return "torchvision"
    if "visual.trunk.patch_embed.proj.weight" in keys and "visual.head.proj.weight" in keys:
        return "open_clip_timm"
    if "patch_embed.proj.weight" in keys:
        return "timm"
    if "vision_model.pre_layrnorm.weight" in keys or "pre_layrnorm.weight" in keys:
        return "hf_clip"
    if "visual.ln_pre.weight" in keys or "ln_pre.weight" in keys:
        return "open_clip"
    # DINOv3ViTModel: `model.layer.<i>.` in memory (transformers 5), `layer.<i>.` in the published model.safetensors
    # (transformers renames on load: conversion_mapping `(?<!model\.)layer.` -> `model.layer.`); accept both
    if any(re.match(r"(model\.)?layer\.\d+\.attention\.q_proj\.weight$", k) for k in keys) and "embeddings.cls_token" in keys:
        return "hf_dinov3"
    if "embeddings.register_tokens" in keys or any(".layer_scale1.lambda1" in k for k in keys):
        return "hf_dinov2"
    if any(k.startswith("embeddings.patch_embeddings") for k in keys):
        return "hf"
    if "patch_embed.weight" in keys:
        return "canonical"
    raise ValueError("unrecognised ViT state dict (expected torchvision, timm, HF ViTModel or canonical keys)")


def resample_position_grid(pos: torch.Tensor, grid: int) -> torch.Tensor:
    """Patch position rows [g0 * g0, D] -> [grid * grid, D] exactly as transformers' ``interpolate_pos_encoding`` does it
    (Dinov2Embeddings: float32, ``F.interpolate(mode="bicubic", align_corners=False)`` on the [1, D, g0, g0] view)."""
    g0 = int(round(pos.shape[0] ** 0.5))
    if g0 * g0 != pos.shape[0]:
        raise ValueError(f"position embedding with {pos.shape[0]} patch rows is not a square grid")
    if g0 == grid:
        return pos
    x = pos.detach().to(torch.float32).reshape(1, g0, g0, -1).permute(0, 3, 1, 2)
    y = torch.nn.functional.interpolate(x, size=(grid, grid), mode="bicubic", align_corners=False)
    return y.permute(0, 2, 3, 1).reshape(grid * grid, -1).contiguous()


def dinov3_rope_tables(grid: int, head_dim: int, theta: float = 100.0) -> tuple[torch.Tensor, torch.Tensor]:
    """cos / sin [grid * grid, head_dim] of transformers' DINOv3ViTRopePositionEmbedding in eval mode (float32, as the module forces):
    patch centres (i + 0.5) / grid mapped to [-1, 1], angles = 2 pi * coord * inv_freq with inv_freq = theta ** -arange(0, 1,
    4 / head_dim), (y | x) blocks flattened to head_dim / 2 and tiled twice."""
    import math
    c = torch.arange(0.5, grid, dtype=torch.float32) / grid
    coords = torch.stack(torch.meshgrid(c, c, indexing="ij"), dim=-1).flatten(0, 1)
    coords = 2.0 * coords - 1.0
    inv_freq = 1 / theta ** torch.arange(0, 1, 4 / head_dim, dtype=torch.float32)
    angles = 2 * math.pi * coords[:, :, None] * inv_freq[None, None, :]
    angles = angles.flatten(1, 2).tile(2)
    return torch.cos(angles).contiguous(), torch.sin(angles).contiguous()


def canonical_state_dict(sd: dict, *, depth: int, layer_scale: bool, source: str = "auto", grid: Optional[int] = None,
                         heads: Optional[int] = None, rope_theta: float = 100.0) -> dict:
    """Return ``{canonical name: float32 CPU tensor}`` for ``ap_vit_set_param``.  ``grid``: patches per side of the input the
    encoder will see; an HF DINOv2 checkpoint trained on another grid has its position rows resampled to it (as the model
    itself does in every forward)."""
    sd = {k: v for k, v in sd.items()}
    if source == "auto":
        source = _detect_source(sd)
    out: dict[str, torch.Tensor] = {}

    def put(name, tensor):
        out[name] = tensor.detach().to(dtype=torch.float32, device="cpu").contiguous()

    if source == "canonical":
        for k, v in sd.items():
            put(k, v)
    elif source == "torchvision":
        put("patch_embed.weight", sd["conv_proj.weight"]); put("patch_embed.bias", sd["conv_proj.bias"])
        put("cls_token", sd["class_token"].reshape(-1)); put("pos_embed", sd["encoder.pos_embedding"][0])
        put("norm.weight", sd["encoder.ln.weight"]); put("norm.bias", sd["encoder.ln.bias"])
        for i in range(depth):
            p, b = f"encoder.layers.encoder_layer_{i}.", f"blocks.{i}."
            put(b + "ln1.weight", sd[p + "ln_1.weight"]); put(b + "ln1.bias", sd[p + "ln_1.bias"])
            put(b + "qkv.weight", sd[p + "self_attention.in_proj_weight"])
            put(b + "qkv.bias", sd[p + "self_attention.in_proj_bias"])
            put(b + "proj.weight", sd[p + "self_attention.out_proj.weight"])
            put(b + "proj.bias", sd[p + "self_attention.out_proj.bias"])
            put(b + "ln2.weight", sd[p + "ln_2.weight"]); put(b + "ln2.bias", sd[p + "ln_2.bias"])
            f1 = "mlp.0" if p + "mlp.0.weight" in sd else "mlp.linear_1"
            f2 = "mlp.3" if p + "mlp.3.weight" in sd else "mlp.linear_2"
            put(b + "fc1.weight", sd[p + f1 + ".weight"]); put(b + "fc1.bias", sd[p + f1 + ".bias"])
            put(b + "fc2.weight", sd[p + f2 + ".weight"]); put(b + "fc2.bias", sd[p + f2 + ".bias"])
    elif source == "open_clip_timm":
        # open_clip TimmModel (biomedclip.py:40: hf-hub:microsoft/BiomedCLIP-...-vit_base_patch16_224): visual.trunk = a timm ViT
        # (class-token pooling), visual.head.proj = the bias-free linear projection [embed_dim, width] [3P, unverified offline]
        trunk = {k[len("visual.trunk."):]: v for k, v in sd.items() if k.startswith("visual.trunk.")}
        out.update(canonical_state_dict(trunk, depth=depth, layer_scale=layer_scale, source="timm", grid=grid))
        put("head_proj.weight", sd["visual.head.proj.weight"])
    elif source == "timm":
        put("patch_embed.weight", sd["patch_embed.proj.weight"]); put("patch_embed.bias", sd["patch_embed.proj.bias"])
        put("cls_token", sd["cls_token"].reshape(-1)); put("pos_embed", sd["pos_embed"][0])
        if "reg_token" in sd:                                         # timm reg_tokens > 0: [1, R, D]
            put("reg_tokens", sd["reg_token"][0])
        put("norm.weight", sd["norm.weight"]); put("norm.bias", sd["norm.bias"])
        for i in range(depth):
            p, b = f"blocks.{i}.", f"blocks.{i}."
            put(b + "ln1.weight", sd[p + "norm1.weight"]); put(b + "ln1.bias", sd[p + "norm1.bias"])
            put(b + "qkv.weight", sd[p + "attn.qkv.weight"]); put(b + "qkv.bias", sd[p + "attn.qkv.bias"])
            put(b + "proj.weight", sd[p + "attn.proj.weight"]); put(b + "proj.bias", sd[p + "attn.proj.bias"])
            put(b + "ln2.weight", sd[p + "norm2.weight"]); put(b + "ln2.bias", sd[p + "norm2.bias"])
            put(b + "fc1.weight", sd[p + "mlp.fc1.weight"]); put(b + "fc1.bias", sd[p + "mlp.fc1.bias"])
            put(b + "fc2.weight", sd[p + "mlp.fc2.weight"]); put(b + "fc2.bias", sd[p + "mlp.fc2.bias"])
            if layer_scale:
                put(b + "ls1", sd[p + "ls1.gamma"]); put(b + "ls2", sd[p + "ls2.gamma"])
    elif source == "hf":
        put("patch_embed.weight", sd["embeddings.patch_embeddings.projection.weight"])
        put("patch_embed.bias", sd["embeddings.patch_embeddings.projection.bias"])
        put("cls_token", sd["embeddings.cls_token"].reshape(-1))
        put("pos_embed", sd["embeddings.position_embeddings"][0])
        put("norm.weight", sd["layernorm.weight"]); put("norm.bias", sd["layernorm.bias"])
        for i in range(depth):
            b = f"blocks.{i}."
            if f"layers.{i}.layernorm_before.weight" in sd:          # transformers >= 5
                p = f"layers.{i}."
                q, k, v, o = (p + "attention." + n for n in ("q_proj", "k_proj", "v_proj", "o_proj"))
                f1, f2 = p + "mlp.fc1", p + "mlp.fc2"
            else:                                                    # transformers 4.x
                p = f"encoder.layer.{i}."
                q, k, v = (p + "attention.attention." + n for n in ("query", "key", "value"))
                o, f1, f2 = p + "attention.output.dense", p + "intermediate.dense", p + "output.dense"
            put(b + "ln1.weight", sd[p + "layernorm_before.weight"]); put(b + "ln1.bias", sd[p + "layernorm_before.bias"])
            put(b + "qkv.weight", torch.cat([sd[q + ".weight"], sd[k + ".weight"], sd[v + ".weight"]], 0))
            put(b + "qkv.bias", torch.cat([sd[q + ".bias"], sd[k + ".bias"], sd[v + ".bias"]], 0))
            put(b + "proj.weight", sd[o + ".weight"]); put(b + "proj.bias", sd[o + ".bias"])
            put(b + "ln2.weight", sd[p + "layernorm_after.weight"]); put(b + "ln2.bias", sd[p + "layernorm_after.bias"])
            put(b + "fc1.weight", sd[f1 + ".weight"]); put(b + "fc1.bias", sd[f1 + ".bias"])
            put(b + "fc2.weight", sd[f2 + ".weight"]); put(b + "fc2.bias", sd[f2 + ".bias"])
    elif source == "hf_clip":
        # transformers CLIPModel / CLIPVisionModelWithProjection (plip.py:34, quilt.py: CLIPModel.from_pretrained): the vision
        # tower + visual_projection.  No patch-embedding bias; class_embedding + position_embedding rows; pre_layrnorm (sic)
        v = "vision_model." if "vision_model.pre_layrnorm.weight" in sd else ""
        d = sd[v + "embeddings.class_embedding"].shape[0]
        put("patch_embed.weight", sd[v + "embeddings.patch_embedding.weight"]); put("patch_embed.bias", torch.zeros(d))
        put("cls_token", sd[v + "embeddings.class_embedding"].reshape(-1))
        put("pos_embed", sd[v + "embeddings.position_embedding.weight"])
        put("pre_norm.weight", sd[v + "pre_layrnorm.weight"]); put("pre_norm.bias", sd[v + "pre_layrnorm.bias"])
        put("norm.weight", sd[v + "post_layernorm.weight"]); put("norm.bias", sd[v + "post_layernorm.bias"])
        if "visual_projection.weight" in sd:
            put("head_proj.weight", sd["visual_projection.weight"])
        for i in range(depth):
            p, b = f"{v}encoder.layers.{i}.", f"blocks.{i}."
            a = p + "self_attn."
            put(b + "ln1.weight", sd[p + "layer_norm1.weight"]); put(b + "ln1.bias", sd[p + "layer_norm1.bias"])
            put(b + "qkv.weight", torch.cat([sd[a + "q_proj.weight"], sd[a + "k_proj.weight"], sd[a + "v_proj.weight"]], 0))
            put(b + "qkv.bias", torch.cat([sd[a + "q_proj.bias"], sd[a + "k_proj.bias"], sd[a + "v_proj.bias"]], 0))
            put(b + "proj.weight", sd[a + "out_proj.weight"]); put(b + "proj.bias", sd[a + "out_proj.bias"])
            put(b + "ln2.weight", sd[p + "layer_norm2.weight"]); put(b + "ln2.bias", sd[p + "layer_norm2.bias"])
            put(b + "fc1.weight", sd[p + "mlp.fc1.weight"]); put(b + "fc1.bias", sd[p + "mlp.fc1.bias"])
            put(b + "fc2.weight", sd[p + "mlp.fc2.weight"]); put(b + "fc2.bias", sd[p + "mlp.fc2.bias"])
    elif source == "open_clip":
        # open_clip VisionTransformer (clip.py:38-40, the OpenAI checkpoints; key names of the public package [3P], unverified
        # offline): visual.conv1 (no bias), class_embedding, positional_embedding, ln_pre, transformer.resblocks.<i>.{ln_1,
        # attn.in_proj_weight | in_proj_bias | out_proj, ln_2, mlp.c_fc, mlp.c_proj}, ln_post, proj [width, output_dim]
        v = "visual." if "visual.ln_pre.weight" in sd else ""
        d = sd[v + "class_embedding"].shape[0]
        put("patch_embed.weight", sd[v + "conv1.weight"]); put("patch_embed.bias", torch.zeros(d))
        put("cls_token", sd[v + "class_embedding"].reshape(-1)); put("pos_embed", sd[v + "positional_embedding"])
        put("pre_norm.weight", sd[v + "ln_pre.weight"]); put("pre_norm.bias", sd[v + "ln_pre.bias"])
        put("norm.weight", sd[v + "ln_post.weight"]); put("norm.bias", sd[v + "ln_post.bias"])
        if v + "proj" in sd:
            put("head_proj.weight", sd[v + "proj"].t())
        for i in range(depth):
            p, b = f"{v}transformer.resblocks.{i}.", f"blocks.{i}."
            put(b + "ln1.weight", sd[p + "ln_1.weight"]); put(b + "ln1.bias", sd[p + "ln_1.bias"])
            put(b + "qkv.weight", sd[p + "attn.in_proj_weight"]); put(b + "qkv.bias", sd[p + "attn.in_proj_bias"])
            put(b + "proj.weight", sd[p + "attn.out_proj.weight"]); put(b + "proj.bias", sd[p + "attn.out_proj.bias"])
            put(b + "ln2.weight", sd[p + "ln_2.weight"]); put(b + "ln2.bias", sd[p + "ln_2.bias"])
            put(b + "fc1.weight", sd[p + "mlp.c_fc.weight"]); put(b + "fc1.bias", sd[p + "mlp.c_fc.bias"])
            put(b + "fc2.weight", sd[p + "mlp.c_proj.weight"]); put(b + "fc2.bias", sd[p + "mlp.c_proj.bias"])
    elif source == "hf_dinov3":
        # transformers DINOv3ViTModel (dinov3.py:52-67): class + register tokens, NO position embedding (zeros here; the rotary
        # tables rope.cos | rope.sin carry the positions), k_proj without bias, LayerScale, plain or gated (silu(gate) * up = the
        # packed SwiGLU layout [gate; up]) MLP, LayerNorm 1e-5, pooler_output = final LayerNorm of the class token
        if grid is None or heads is None:
            raise ValueError("hf_dinov3: the patch grid and the head count are needed for the rotary tables")
        d = sd["embeddings.cls_token"].shape[-1]
        put("patch_embed.weight", sd["embeddings.patch_embeddings.weight"]); put("patch_embed.bias", sd["embeddings.patch_embeddings.bias"])
        put("cls_token", sd["embeddings.cls_token"].reshape(-1))
        if sd["embeddings.register_tokens"].shape[1] > 0:
            put("reg_tokens", sd["embeddings.register_tokens"][0])
        put("pos_embed", torch.zeros(int(grid) * int(grid), d))
        put("norm.weight", sd["norm.weight"]); put("norm.bias", sd["norm.bias"])
        cos, sin = dinov3_rope_tables(int(grid), d // int(heads), rope_theta)
        put("rope.cos", cos); put("rope.sin", sin)
        root = "model.layer." if any(k.startswith("model.layer.") for k in sd) else "layer."      # in-memory / on-disk names
        for i in range(depth):
            p, b = f"{root}{i}.", f"blocks.{i}."
            a = p + "attention."
            zeros = torch.zeros(d)
            bias = lambda n: sd[a + n + ".bias"] if a + n + ".bias" in sd else zeros
            put(b + "ln1.weight", sd[p + "norm1.weight"]); put(b + "ln1.bias", sd[p + "norm1.bias"])
            put(b + "qkv.weight", torch.cat([sd[a + "q_proj.weight"], sd[a + "k_proj.weight"], sd[a + "v_proj.weight"]], 0))
            put(b + "qkv.bias", torch.cat([bias("q_proj"), bias("k_proj"), bias("v_proj")], 0))
            put(b + "proj.weight", sd[a + "o_proj.weight"]); put(b + "proj.bias", sd[a + "o_proj.bias"])
            put(b + "ln2.weight", sd[p + "norm2.weight"]); put(b + "ln2.bias", sd[p + "norm2.bias"])
            if p + "mlp.gate_proj.weight" in sd:
                put(b + "fc1.weight", torch.cat([sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"]], 0))
                put(b + "fc1.bias", torch.cat([sd[p + "mlp.gate_proj.bias"], sd[p + "mlp.up_proj.bias"]], 0))
            else:
                put(b + "fc1.weight", sd[p + "mlp.up_proj.weight"]); put(b + "fc1.bias", sd[p + "mlp.up_proj.bias"])
            put(b + "fc2.weight", sd[p + "mlp.down_proj.weight"]); put(b + "fc2.bias", sd[p + "mlp.down_proj.bias"])
            if layer_scale:
                put(b + "ls1", sd[p + "layer_scale1.lambda1"]); put(b + "ls2", sd[p + "layer_scale2.lambda1"])
    elif source == "hf_dinov2":
        # transformers Dinov2Model / Dinov2WithRegistersModel: the position embedding has a class row and patch rows but none
        # for the register tokens, which are inserted AFTER it is added -> canonical no_embed_class form with the class row
        # folded into the class token (one f32 add, the same one the model performs)
        pos = sd["embeddings.position_embeddings"][0]
        put("patch_embed.weight", sd["embeddings.patch_embeddings.projection.weight"])
        put("patch_embed.bias", sd["embeddings.patch_embeddings.projection.bias"])
        put("cls_token", (sd["embeddings.cls_token"].reshape(-1).float() + pos[0].float()))
        if "embeddings.register_tokens" in sd:
            put("reg_tokens", sd["embeddings.register_tokens"][0])
        put("pos_embed", pos[1:] if grid is None else resample_position_grid(pos[1:], int(grid)))
        put("norm.weight", sd["layernorm.weight"]); put("norm.bias", sd["layernorm.bias"])
        for i in range(depth):
            p, b = f"encoder.layer.{i}.", f"blocks.{i}."
            a = p + "attention.attention."
            put(b + "ln1.weight", sd[p + "norm1.weight"]); put(b + "ln1.bias", sd[p + "norm1.bias"])
            put(b + "qkv.weight", torch.cat([sd[a + "query.weight"], sd[a + "key.weight"], sd[a + "value.weight"]], 0))
            put(b + "qkv.bias", torch.cat([sd[a + "query.bias"], sd[a + "key.bias"], sd[a + "value.bias"]], 0))
            put(b + "proj.weight", sd[p + "attention.output.dense.weight"]); put(b + "proj.bias", sd[p + "attention.output.dense.bias"])
            put(b + "ln2.weight", sd[p + "norm2.weight"]); put(b + "ln2.bias", sd[p + "norm2.bias"])
            f1, f2 = (("mlp.weights_in", "mlp.weights_out") if p + "mlp.weights_in.weight" in sd else ("mlp.fc1", "mlp.fc2"))
            put(b + "fc1.weight", sd[p + f1 + ".weight"]); put(b + "fc1.bias", sd[p + f1 + ".bias"])
            put(b + "fc2.weight", sd[p + f2 + ".weight"]); put(b + "fc2.bias", sd[p + f2 + ".bias"])
            if layer_scale:
                put(b + "ls1", sd[p + "layer_scale1.lambda1"]); put(b + "ls2", sd[p + "layer_scale2.lambda1"])
    else:
        raise ValueError(f"unknown state-dict source '{source}'")
    return out


def stored_head_dim(dim: int, heads: int) -> int:
    """Width of one q / k / v head as the device stores it: 64, 96 or 128 (other true widths are zero-padded up; 96 -- the 80-wide
    heads of ViT-H/14 and Virchow -- exists in the float16 / bfloat16 attention kernels only, like 128)."""
    hd = dim // heads
    if hd <= 64:
        return 64
    if hd <= 96:
        return 96
    if hd <= 128:
        return 128
    raise ValueError(f"head width {hd} (dim {dim} / heads {heads}) exceeds 128")


def pad_heads(state: dict, *, dim: int, heads: int, depth: int) -> dict:
    """Canonical state dict -> the same model with every attention head zero-padded from dim / heads to
    ``stored_head_dim`` channels: qkv rows [3, heads, hd, :] -> [3, heads, hdp, :], proj columns likewise.  Zero q / k
    channels add nothing to q.k, zero v channels give zero outputs that meet zero proj columns: the function is unchanged
    as long as the softmax scale stays 1 / sqrt(hd) (``attn_scale``)."""
    hd, hdp = dim // heads, stored_head_dim(dim, heads)
    if hd == hdp:
        return state
    out = dict(state)
    for i in range(depth):
        b = f"blocks.{i}."
        w = state[b + "qkv.weight"].reshape(3, heads, hd, -1)
        wp = torch.zeros((3, heads, hdp, w.shape[-1]), dtype=w.dtype); wp[:, :, :hd] = w
        out[b + "qkv.weight"] = wp.reshape(3 * heads * hdp, -1).contiguous()
        bb = state[b + "qkv.bias"].reshape(3, heads, hd)
        bp = torch.zeros((3, heads, hdp), dtype=bb.dtype); bp[:, :, :hd] = bb
        out[b + "qkv.bias"] = bp.reshape(-1).contiguous()
        pw = state[b + "proj.weight"].reshape(-1, heads, hd)
        pp = torch.zeros((pw.shape[0], heads, hdp), dtype=pw.dtype); pp[:, :, :hd] = pw
        out[b + "proj.weight"] = pp.reshape(pw.shape[0], heads * hdp).contiguous()
    return out


def stored_mlp_dim(mlp_dim: int) -> int:
    """Hidden width of the MLP as the device stores it: the next multiple of 128 (the GEMM kernels' N / K granularity)."""
    return (int(mlp_dim) + 127) // 128 * 128


def pad_mlp(state: dict, *, mlp_dim: int, depth: int, swiglu: bool) -> dict:
    """Canonical state dict -> the same model with the MLP's hidden width zero-padded to ``stored_mlp_dim`` (Virchow: timm
    ``mlp_ratio=5.3375`` on dim 1280 gives a packed width of 6832 = 2 x 3416, and 3416 is no multiple of 64).  Padded fc1 rows have
    zero weights and biases, so their activations are gelu(0) = 0 (or silu(0) * 0 = 0) and meet zero fc2 columns: the function
    is unchanged.  SwiGLU: the two halves x1 | x2 are padded separately."""
    hp = stored_mlp_dim(mlp_dim)
    if hp == mlp_dim:
        return state
    out = dict(state)
    for i in range(depth):
        b = f"blocks.{i}."
        w1, b1, w2 = state[b + "fc1.weight"], state[b + "fc1.bias"], state[b + "fc2.weight"]
        halves = 2 if swiglu else 1
        w1p = torch.zeros((halves, hp, w1.shape[1]), dtype=w1.dtype); w1p[:, :mlp_dim] = w1.reshape(halves, mlp_dim, -1)
        b1p = torch.zeros((halves, hp), dtype=b1.dtype); b1p[:, :mlp_dim] = b1.reshape(halves, mlp_dim)
        w2p = torch.zeros((w2.shape[0], hp), dtype=w2.dtype); w2p[:, :mlp_dim] = w2
        out[b + "fc1.weight"] = w1p.reshape(halves * hp, -1).contiguous()
        out[b + "fc1.bias"] = b1p.reshape(-1).contiguous()
        out[b + "fc2.weight"] = w2p.contiguous()
    return out


def attn_pool_canonical(pool: dict, *, pool_eps: float = 1e-5) -> dict:
    """open_clip ``AttentionalPooler`` (+ the LayerNorm after it) -> ``attn_pool.*`` parameters.

    ``pool`` holds the module's own tensors: ``query`` [1, P], ``ln_q.weight|bias``, ``ln_k.weight|bias``,
    ``attn.q_proj_weight`` [P, P], ``attn.k_proj_weight`` / ``attn.v_proj_weight`` [P, C], ``attn.in_proj_bias``
    [3P], ``attn.out_proj.weight|bias`` and ``ln_out.weight|bias`` (CONCH: ``visual.ln_contrast``).
    The projected query is input independent and is folded here (float64 on the host)."""
    f = lambda k: pool[k].detach().to(torch.float64, copy=True).cpu()
    P = pool["attn.q_proj_weight"].shape[0]
    q = torch.nn.functional.layer_norm(f("query").reshape(1, P), (P,), f("ln_q.weight"), f("ln_q.bias"), pool_eps)
    q = q @ f("attn.q_proj_weight").T + f("attn.in_proj_bias")[:P]
    out = {"attn_pool.q": q.reshape(-1),
           "attn_pool.ln_k.weight": f("ln_k.weight"), "attn_pool.ln_k.bias": f("ln_k.bias"),
           "attn_pool.kv.weight": torch.cat([f("attn.k_proj_weight"), f("attn.v_proj_weight")], 0),
           "attn_pool.kv.bias": f("attn.in_proj_bias")[P:],
           "attn_pool.out.weight": f("attn.out_proj.weight"), "attn_pool.out.bias": f("attn.out_proj.bias"),
           "attn_pool.ln_out.weight": f("ln_out.weight"), "attn_pool.ln_out.bias": f("ln_out.bias")}
    return {k: v.to(torch.float32).contiguous() for k, v in out.items()}


def conch_state_dicts(sd: dict) -> tuple[dict, dict]:
    """Split a CONCH v1 checkpoint (open_clip_custom CoCa; key names from the public package [3P], unverified
    offline) into the timm trunk state dict and the pooler tensors ``attn_pool_canonical`` expects."""
    trunk = {k[len("visual.trunk."):]: v for k, v in sd.items() if k.startswith("visual.trunk.")}
    pre = "visual.attn_pool_contrast."
    pool = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    pool["ln_out.weight"] = sd["visual.ln_contrast.weight"]
    pool["ln_out.bias"] = sd["visual.ln_contrast.bias"]
    return trunk, pool


def random_attn_pool(arch: dict, seed: int = 0) -> dict:
    """Seeded AttentionalPooler tensors (module-style names) for tests / benchmarks."""
    g = torch.Generator().manual_seed(seed + 7919)
    P, C = arch["pool_dim"], arch["dim"]
    w = lambda *shape, s=0.02: torch.randn(*shape, generator=g) * s
    return {"query": torch.randn(1, P, generator=g), "ln_q.weight": 1.0 + w(P, s=0.1), "ln_q.bias": w(P),
            "ln_k.weight": 1.0 + w(C, s=0.1), "ln_k.bias": w(C),
            "attn.q_proj_weight": w(P, P, s=0.05), "attn.k_proj_weight": w(P, C, s=0.05),
            "attn.v_proj_weight": w(P, C, s=0.05), "attn.in_proj_bias": w(3 * P),
            "attn.out_proj.weight": w(P, P, s=0.05), "attn.out_proj.bias": w(P),
            "ln_out.weight": 1.0 + w(P, s=0.1), "ln_out.bias": w(P)}


def random_canonical_state_dict(arch: dict, seed: int = 0) -> dict:
    """Seeded random-init weights (trunc-normal-ish sigma 0.02, LN gamma ~ 1): there are no
    pretrained checkpoints offline."""
    g = torch.Generator().manual_seed(seed)
    d, mlp, ps = arch["dim"], arch["mlp_dim"], arch["patch_size"]
    reg = int(arch.get("reg_tokens", 0))
    patches = (arch["image_size"] // ps) ** 2
    tokens = patches if arch.get("no_embed_class") else 1 + reg + patches          # rows of the position embedding
    f1 = 2 * mlp if arch.get("mlp") == "swiglu" else mlp

    def w(*shape, s=0.02):
        return torch.randn(*shape, generator=g) * s

    sd = {"patch_embed.weight": w(d, 3, ps, ps), "patch_embed.bias": w(d), "cls_token": w(d),
          "pos_embed": w(tokens, d), "norm.weight": 1.0 + w(d, s=0.1), "norm.bias": w(d)}
    if reg:
        sd["reg_tokens"] = w(reg, d)
    for i in range(arch["depth"]):
        b = f"blocks.{i}."
        sd[b + "ln1.weight"] = 1.0 + w(d, s=0.1); sd[b + "ln1.bias"] = w(d)
        sd[b + "qkv.weight"] = w(3 * d, d); sd[b + "qkv.bias"] = w(3 * d)
        sd[b + "proj.weight"] = w(d, d); sd[b + "proj.bias"] = w(d)
        sd[b + "ln2.weight"] = 1.0 + w(d, s=0.1); sd[b + "ln2.bias"] = w(d)
        sd[b + "fc1.weight"] = w(f1, d); sd[b + "fc1.bias"] = w(f1)
        sd[b + "fc2.weight"] = w(d, mlp); sd[b + "fc2.bias"] = w(d)
        if arch.get("layer_scale"):
            sd[b + "ls1"] = torch.full((d,), 1e-5) + w(d, s=1e-6)
            sd[b + "ls2"] = torch.full((d,), 1e-5) + w(d, s=1e-6)
    if arch.get("pool") == "attn":
        sd.update(attn_pool_canonical(random_attn_pool(arch, seed), pool_eps=arch.get("pool_ln_eps", 1e-5)))
    if arch.get("rope"):
        sd["pos_embed"] = torch.zeros_like(sd["pos_embed"])
        sd["rope.cos"], sd["rope.sin"] = dinov3_rope_tables(arch["image_size"] // ps, d // arch["heads"], float(arch.get("rope_theta", 100.0)))
    if arch.get("pre_norm"):
        sd["pre_norm.weight"] = 1.0 + w(d, s=0.1); sd["pre_norm.bias"] = w(d)
    if arch.get("proj_dim"):
        sd["head_proj.weight"] = w(arch["proj_dim"], d, s=d ** -0.5)
    return sd


# ----------------------------------------------------------------------------- device object
class HipViT:
    """Device-resident ViT encoder behind ``ap_vit_*``."""

    def __init__(self, arch: dict, state: dict, *, device: torch.device, dtype: torch.dtype) -> None:
        if device.type != "cuda":
            raise _lib.HipLibraryError("HipViT needs a HIP device ('cuda' on PyTorch-ROCm); "
                                       "there is no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.dtype = dtype
        self.arch = dict(arch)
        attn_pool = arch.get("pool") == "attn"
        cls_mean = arch.get("pool") == "cls_mean"          # [class token | mean of the patch tokens] (midnight.py:58-61)
        self.embed_dim = int(arch["pool_dim"] if attn_pool else (2 * arch["dim"] if cls_mean else arch["dim"]))
        if arch.get("proj_dim"):                           # CLIP visual projection
            self.embed_dim = int(arch["proj_dim"])
        hd_true = arch["dim"] // arch["heads"]
        hd_stored = stored_head_dim(arch["dim"], arch["heads"])
        cfg = _lib.VitConfig(arch["image_size"], arch["patch_size"], arch["dim"], arch["depth"],
                             arch["heads"], stored_mlp_dim(arch["mlp_dim"]), float(arch["ln_eps"]),
                             1 if arch.get("layer_scale") else 0, _lib.torch_dtype_code(dtype),
                             1 if attn_pool else (2 if cls_mean else 0), int(arch.get("pool_dim", 0)), int(arch.get("pool_heads", 0)),
                             float(arch.get("pool_ln_eps", 1e-5)),
                             int(arch.get("reg_tokens", 0)), 1 if arch.get("no_embed_class") else 0,
                             1 if arch.get("mlp") == "swiglu" else 0, hd_stored,
                             0.0 if hd_stored == hd_true else float(1.0 / np.sqrt(np.float32(hd_true))),
                             1 if arch.get("pre_norm") else 0, 1 if arch.get("act") == "quick_gelu" else 0,
                             int(arch.get("proj_dim", 0)), 1 if arch.get("rope") else 0)
        state = pad_heads(state, dim=arch["dim"], heads=arch["heads"], depth=arch["depth"])
        state = pad_mlp(state, mlp_dim=arch["mlp_dim"], depth=arch["depth"], swiglu=arch.get("mlp") == "swiglu")
        handle = C.c_void_p()
        # hipMalloc / hipMemcpy on the legacy stream must not fall into another thread's stream capture (the SAM2 hipGraph):
        # both sides hold _lib.HIP_CAPTURE_LOCK for their device section
        with _lib.HIP_CAPTURE_LOCK, torch.cuda.device(self.device):
            _lib.check(self.lib.ap_vit_create(C.byref(cfg), C.byref(handle)), "ap_vit_create")
            self._handle = handle
            # one native call for the whole checkpoint: the uploads run outside the interpreter lock (the encoder is built
            # on a side thread while the CLI's phase 1 runs)
            arrs = [np.ascontiguousarray(t.detach().to(torch.float32).cpu().numpy()) for t in state.values()]
            n = len(arrs)
            names = (C.c_char_p * n)(*[k.encode() for k in state])
            ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
            counts = (C.c_size_t * n)(*[a.size for a in arrs])
            _lib.check(self.lib.ap_vit_set_params(self._handle, names, ptrs, counts, n), "ap_vit_set_params")
            del arrs
            _lib.check(self.lib.ap_vit_finalize(self._handle), "ap_vit_finalize")
        self._workspace: Optional[torch.Tensor] = None

    def _ws(self, n: int) -> torch.Tensor:
        need = int(self.lib.ap_vit_workspace_bytes(self._handle, n))
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = None
            self._workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._workspace

    def forward_u8(self, tiles: torch.Tensor, mean, std, out: torch.Tensor) -> torch.Tensor:
        """tiles: uint8 [n, H, W, 3] on the device; out: float32 [n, D] on the device (written)."""
        if self._handle is None:
            raise _lib.HipLibraryError("HipViT used after release()")
        assert tiles.dtype == torch.uint8 and tiles.is_contiguous() and tiles.dim() == 4 and tiles.shape[3] == 3
        assert out.dtype == torch.float32 and out.is_contiguous() and out.shape == (tiles.shape[0], self.embed_dim)
        n, h, w, _ = tiles.shape
        if n == 0:
            return out
        ws = self._ws(n)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ap_vit_forward_u8(self._handle, tiles.data_ptr(), n, h, w, _lib.f3(mean), _lib.f3(std),
                                                  out.data_ptr(), ws.data_ptr(), ws.numel(),
                                                  _lib.current_stream_ptr(self.device)), "ap_vit_forward_u8")
        return out

    def forward_chw(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x: [n, 3, S, S] float32 or compute dtype, already normalised."""
        if self._handle is None:
            raise _lib.HipLibraryError("HipViT used after release()")
        assert x.is_contiguous() and x.dim() == 4
        n = x.shape[0]
        if out is None:
            out = torch.empty((n, self.embed_dim), dtype=torch.float32, device=self.device)
        if n == 0:
            return out
        ws = self._ws(n)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ap_vit_forward_chw(self._handle, x.data_ptr(), _lib.torch_dtype_code(x.dtype), n,
                                                   out.data_ptr(), ws.data_ptr(), ws.numel(),
                                                   _lib.current_stream_ptr(self.device)), "ap_vit_forward_chw")
        return out

    OPTIONS = {"full_last_block": 0, "two_half_overlap": 1, "f32_stream": 2, "exact_cls": 3, "split_f16": 4}

    def set_option(self, name: str, on: bool) -> None:
        """``full_last_block``: the last block for every token instead of the CLS row only; ``two_half_overlap``: a
        batch >= 512 as two halves on two streams.  Both leave the features unchanged (tested).  ``f32_stream``
        (f16 / bf16): float32 residual stream with standalone add+LayerNorm launches instead of the fused-LayerNorm
        dataflow (slower).  ``exact_cls`` (default ON; f16 / bf16 fused dataflow with a class-token pooling): the class rows'
        residual stream is additionally carried in float32, which removes most of the 16-bit stream's error from the features
        (ViT-B/16 float16: 1.26e-3 -> 8.0e-4 against the CPU fp32 path, 1.5 % of the step); off = the plain 16-bit stream.
        ``split_f16`` (float32 only): the GEMMs' inner products as three f16 MFMA passes on hi / lo halves with f32
        accumulation instead of the exact f32 MFMA -- every buffer, LayerNorm, softmax and the stream stay float32."""
        value = int(on) if not isinstance(on, bool) else (1 if on else 0)
        _lib.check(self.lib.ap_vit_set_option(self._handle, self.OPTIONS[name], value), "ap_vit_set_option")

    def profile(self, on: bool) -> None:
        _lib.check(self.lib.ap_vit_profile_enable(self._handle, 1 if on else 0), "ap_vit_profile_enable")

    def profile_read(self) -> dict:
        """{kind: (milliseconds, launches)} accumulated since the last read (HIP events)."""
        k = len(_lib.PROF_KINDS)
        ms = (C.c_double * k)()
        cnt = (C.c_longlong * k)()
        _lib.check(self.lib.ap_vit_profile_read(self._handle, ms, cnt, k), "ap_vit_profile_read")
        return {name: (float(ms[i]), int(cnt[i])) for i, name in enumerate(_lib.PROF_KINDS)}

    def release(self) -> None:
        if getattr(self, "_handle", None) is not None:
            torch.cuda.synchronize(self.device)
            self.lib.ap_vit_destroy(self._handle)
            self._handle = None
            self._workspace = None

    def __del__(self):  # pragma: no cover
        try:
            self.release()
        except Exception:
            pass


# ----------------------------------------------------------------------------- builders
def weights_path(name: str) -> Optional[Path]:
    """``$ATLASPATCH_WEIGHTS_DIR/<name>.{safetensors,pt,pth}`` if present."""
    root = os.environ.get("ATLASPATCH_WEIGHTS_DIR")
    if not root:
        return None
    for ext in (".safetensors", ".pt", ".pth"):
        candidate = Path(root) / f"{name}{ext}"
        if candidate.exists():
            return candidate
    return None


def load_checkpoint(path: Path) -> dict:
    if path.suffix == ".safetensors":
        from safetensors.torch import load_file
        return load_file(str(path))
    obj = torch.load(str(path), map_location="cpu", weights_only=True)
    for key in ("model", "state_dict"):
        if isinstance(obj, dict) and key in obj and isinstance(obj[key], dict):
            obj = obj[key]
    return obj


def build_hip_vit_extractor(*, name: str, arch, device, dtype, state_dict: Optional[dict] = None,
                            mean=None, std=None, source: str = "auto", max_batch: int = 1024,
                            random_init_seed: Optional[int] = None, resize=None,
                            expect_size: Optional[int] = 256, **arch_overrides) -> HipViTFeatureExtractor:
    spec = dict(ARCHS[arch]) if isinstance(arch, str) else dict(arch)
    spec.update(arch_overrides)
    if state_dict is None:
        path = weights_path(name)
        if path is not None:
            state_dict = load_checkpoint(path)
        elif random_init_seed is not None:
            logger.warning("%s: using seeded RANDOM weights (seed %d); features are not meaningful",
                           name, random_init_seed)
            state_dict = random_canonical_state_dict(spec, random_init_seed)
            source = "canonical"
        else:
            raise FileNotFoundError(
                f"No weights for '{name}': set ATLASPATCH_WEIGHTS_DIR to a directory holding "
                f"{name}.safetensors/.pt (torchvision, timm or HF ViT key names), or set "
                "ATLASPATCH_RANDOM_INIT=<seed> for seeded random weights (benchmarks/tests).")
    pool_state = None
    if spec.get("pool") == "attn" and source != "canonical":
        if any(k.startswith("visual.trunk.") for k in state_dict):          # a CONCH checkpoint
            state_dict, pool = conch_state_dicts(state_dict)
            pool_state = attn_pool_canonical(pool, pool_eps=spec.get("pool_ln_eps", 1e-5))
        else:                                                                # trunk dict + "attn_pool.*" tensors
            pool_state = {k: v for k, v in state_dict.items() if k.startswith("attn_pool.")}
            state_dict = {k: v for k, v in state_dict.items() if not k.startswith("attn_pool.")}
    state = canonical_state_dict(state_dict, depth=spec["depth"], layer_scale=bool(spec.get("layer_scale")),
                                 source=source, grid=spec["image_size"] // spec["patch_size"], heads=spec["heads"],
                                 rope_theta=float(spec.get("rope_theta", 100.0)))
    if pool_state:
        state.update({k: v.detach().to(torch.float32).cpu().contiguous() for k, v in pool_state.items()})
    vit = HipViT(spec, state, device=torch.device(device), dtype=dtype)
    return HipViTFeatureExtractor(name=name, vit=vit, mean=mean or IMAGENET_MEAN, std=std or IMAGENET_STD,
                                  max_batch=max_batch, resize=resize, expect_size=expect_size)


def register_conch(registry, *, device, dtype=torch.float32, num_workers: int = 0) -> None:
    """conch_v1 (models/patch/conch.py:20-64): open_clip image transform = Resize(448, bicubic) + CenterCrop(448)
    + ToTensor + Normalize(OpenAI CLIP mean / std) [3P]; all of it on the device (the resize is Pillow-exact,
    ap_resample_u8).  float16 / bfloat16 only (the reference's config 5 runs it in fp16)."""
    registry.register("conch_v1", lambda: build_hip_vit_extractor(
        name="conch_v1", arch="conch_v1", device=device, dtype=dtype, random_init_seed=_env_seed(),
        mean=OPENAI_CLIP_MEAN, std=OPENAI_CLIP_STD, resize=TRANSFORM_RESIZE["conch_v1"], expect_size=None,
        max_batch=256))


def _env_seed() -> Optional[int]:
    raw = os.environ.get("ATLASPATCH_RANDOM_INIT")
    return int(raw) if raw not in (None, "") else None


def register_vits(registry, *, device, dtype=torch.float32, num_workers: int = 0) -> None:
    """vit_b_16 / vit_l_16 with torchvision's transform semantics: ImageClassification(crop 224, resize R, bilinear)
    on the PIL tile (models/patch/base.py:42-45 hands the transform a PIL image, so the resize is Pillow's BILINEAR),
    R = 256 for vit_b_16 and 242 for vit_l_16 (``TRANSFORM_RESIZE``).  vit_b_16 on the default 256-px tiles is a pure
    centre crop; vit_l_16 resamples every 256-px tile to 242 first; any other --patch-size goes through the same
    Pillow-exact device resize (shorter side -> R)."""
    for name in ("vit_b_16", "vit_l_16"):
        registry.register(name, lambda n=name: build_hip_vit_extractor(
            name=n, arch=n, device=device, dtype=dtype, random_init_seed=_env_seed(), resize=TRANSFORM_RESIZE[n],
            expect_size=None, max_batch=2048))
    # the other three names of models/patch/vit.py:9-15.  vit_b_32 / vit_l_32: 32-px patches, 50 tokens, same transform as
    # vit_b_16.  vit_h_14: torchvision has no IMAGENET1K_V1 entry for it, so the reference (base.py:123-143) takes DEFAULT =
    # IMAGENET1K_SWAG_E2E_V1: 518-px input (Pillow BICUBIC resize of the tile, on the device), 1370 tokens, 80-wide heads
    for name, cap in (("vit_b_32", 4096), ("vit_l_32", 4096), ("vit_h_14", 128)):
        registry.register(name, lambda n=name, c=cap: build_hip_vit_extractor(
            name=n, arch=n, device=device, dtype=dtype, random_init_seed=_env_seed(), resize=TRANSFORM_RESIZE[n],
            expect_size=None, max_batch=c))


def register_uni(registry, *, device, dtype=torch.float32, num_workers: int = 0) -> None:
    """uni_v1: timm ViT-L/16 + LayerScale; timm transform Resize(224, bicubic) + CenterCrop(224):
    the resize runs on the device bit-identically to Pillow (ap_resample_u8), the crop in the preprocess kernel."""
    registry.register("uni_v1", lambda: build_hip_vit_extractor(
        name="uni_v1", arch="uni_v1", device=device, dtype=dtype, random_init_seed=_env_seed(),
        resize=TRANSFORM_RESIZE["uni_v1"], expect_size=None, max_batch=2048))
    # uni_v2 = UNI2-h (uni.py:62-125): ViT-H/14 at 224 px with 8 register tokens, SwiGLUPacked MLP, LayerScale; same timm
    # transform as uni_v1.  Checkpoint: timm key names (pytorch_model.bin of MahmoodLab/UNI2-h) in ATLASPATCH_WEIGHTS_DIR
    registry.register("uni_v2", lambda: build_hip_vit_extractor(
        name="uni_v2", arch="uni_v2", device=device, dtype=dtype, random_init_seed=_env_seed(),
        resize=TRANSFORM_RESIZE["uni_v2"], expect_size=None, max_batch=1024))


def register_dinov2(registry, *, device, dtype=torch.float32, num_workers: int = 0) -> None:
    """dinov2_small / base / large / giant (models/patch/dinov2.py:12-17): the reference runs transformers' Dinov2Model on the
    processor's output and returns the class token of ``last_hidden_state``.  Same device kernels as the other encoders (LayerScale
    folded, giant: SwiGLU gate in the fc1 epilogue); checkpoints: the HF state dict (model.safetensors of facebook/dinov2-*) as
    ``$ATLASPATCH_WEIGHTS_DIR/<name>.safetensors``."""
    for name, cap in (("dinov2_small", 4096), ("dinov2_base", 2048), ("dinov2_large", 2048), ("dinov2_giant", 512)):
        registry.register(name, lambda n=name, c=cap: build_hip_vit_extractor(
            name=n, arch=n, device=device, dtype=dtype, random_init_seed=_env_seed(), resize=TRANSFORM_RESIZE[n],
            expect_size=None, max_batch=c))


def register_phikon(registry, *, device, dtype=torch.float32, num_workers: int = 0) -> None:
    """phikon_v1 (transformers ViTModel, ViT-B/16, LayerNorm eps 1e-12) and phikon_v2 (Dinov2Model ViT-L/16), models/patch/phikon.py:
    class token of ``last_hidden_state``.  HF state dicts in ATLASPATCH_WEIGHTS_DIR."""
    for name in ("phikon_v1", "phikon_v2"):
        registry.register(name, lambda n=name: build_hip_vit_extractor(
            name=n, arch=n, device=device, dtype=dtype, random_init_seed=_env_seed(), resize=TRANSFORM_RESIZE[n],
            expect_size=None, max_batch=2048))


def register_more_vits(registry, *, device, dtype=torch.float32, num_workers: int = 0) -> None:
    """The other loader files whose model is a plain ViT this engine runs: midnight (models/patch/midnight.py, transformers
    Dinov2Model giant, class token + mean patch token), h_optimus_0 / h_optimus_1 (hoptimus.py), prov_gigapath (gigapath.py),
    virchow_v1 / virchow_v2 (virchow.py: 80-wide heads and a 3416-wide SwiGLU stored zero-padded, class | mean patch token),
    the two Lunit ViT-S (lunit.py) and pathorchestra (pathorchestra.py).  Checkpoints: HF (midnight) or timm key names in
    ATLASPATCH_WEIGHTS_DIR; each with the transform its loader file writes out (``TRANSFORM_RESIZE`` / ``TRANSFORM_NORM``)."""
    for name, cap in (("midnight", 512), ("h_optimus_0", 512), ("h_optimus_1", 512), ("prov_gigapath", 512),
                      ("virchow_v1", 512), ("virchow_v2", 512), ("h0_mini", 2048),
                      ("lunit_vit_small_patch16_dino", 4096), ("lunit_vit_small_patch8_dino", 512), ("pathorchestra", 2048)):
        mean, std = TRANSFORM_NORM.get(name, (IMAGENET_MEAN, IMAGENET_STD))
        registry.register(name, lambda n=name, c=cap, mu=mean, sd=std: build_hip_vit_extractor(
            name=n, arch=n, device=device, dtype=dtype, random_init_seed=_env_seed(), resize=TRANSFORM_RESIZE[n],
            expect_size=None, max_batch=c, mean=mu, std=sd))


def register_clip(registry, *, device, dtype=torch.float32, num_workers: int = 0) -> None:
    """The CLIP vision towers with a plain ViT: clip_vit_b_32 / b_16 / l_14 / l_14_336 (models/patch/clip.py, open_clip, OpenAI
    weights), plip (plip.py) and quilt_b_32 / quilt_b_16 (quilt.py), both transformers CLIPModel; features = encode_image /
    get_image_features (projected, not normalised); biomedclip (biomedclip.py: timm ViT-B/16 trunk + linear projection).  Checkpoints: open_clip or HF CLIP state dicts in ATLASPATCH_WEIGHTS_DIR.
    (The ResNet CLIPs, quilt_b_16_pmb and omiclip are other architectures.)"""
    for name, cap in (("clip_vit_b_32", 4096), ("clip_vit_b_16", 2048), ("clip_vit_l_14", 1024), ("clip_vit_l_14_336", 256),
                      ("plip", 4096), ("quilt_b_32", 4096), ("quilt_b_16", 2048), ("biomedclip", 2048)):
        mean, std = TRANSFORM_NORM[name]
        registry.register(name, lambda n=name, c=cap, mu=mean, sd=std: build_hip_vit_extractor(
            name=n, arch=n, device=device, dtype=dtype, random_init_seed=_env_seed(), resize=TRANSFORM_RESIZE[n],
            expect_size=None, max_batch=c, mean=mu, std=sd))


def register_dinov3(registry, *, device, dtype=torch.float32, num_workers: int = 0) -> None:
    """dinov3_vits16 / vits16_plus / vitb16 / vitl16 / vitl16_sat / vith16_plus / vit7b16 / vit7b16_sat (models/patch/dinov3.py:12-21):
    transformers' DINOv3ViTModel, ``pooler_output``.  Rotary position embedding applied in place on q / k after the qkv GEMM (``ap::launch_rope``);
    HF state dicts in ATLASPATCH_WEIGHTS_DIR."""
    for name, cap in (("dinov3_vits16", 4096), ("dinov3_vits16_plus", 4096), ("dinov3_vitb16", 2048), ("dinov3_vitl16", 2048),
                      ("dinov3_vitl16_sat", 2048), ("dinov3_vith16_plus", 1024), ("dinov3_vit7b16", 512), ("dinov3_vit7b16_sat", 512)):
        mean, std = TRANSFORM_NORM.get(name, (IMAGENET_MEAN, IMAGENET_STD))
        registry.register(name, lambda n=name, c=cap, mu=mean, sd=std: build_hip_vit_extractor(
            name=n, arch=n, device=device, dtype=dtype, random_init_seed=_env_seed(), resize=TRANSFORM_RESIZE[n],
            expect_size=None, max_batch=c, mean=mu, std=sd))
