"""CPU oracle: SAM2.1 Hiera-T image path as the reference drives it (box prompt = whole image).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PINNED against an independent implementation: the ``sam2`` package
(facebookresearch/sam2 @ git HEAD, /root/reference/pyproject.toml:63-64) and the AtlasPatch checkpoint
(``AtlasAnalyticsLab/AtlasPatch:model.pth``, services/segmentation.py:28-29) are absent from this image, but Hugging Face
``transformers.models.sam2.Sam2Model`` (default config = Hiera-T) is present: ``tests/golden/gen_golden_hf_sam2.py`` loads
the seeded weights below into it through ``atlaspatch_amd/services/sam2_keys.py`` and records its feature levels and mask
logits (``tests/golden/hf_sam2.npz``); ``tests/test_sam2_hf_pin.py`` holds this file to 1e-5 of them (measured 2.8e-7) and the
HIP path to 1e-4 / 2e-4.  Still unpinned: the real checkpoint's tensors (never seen here) — the key NAMES are checked only
through the HF map, whose facebook side is from the public sources.
This file restates, in explicit torch fp32 ops, the modules the reference instantiates from
``atlas_patch/configs/sam2.1_hiera_t.yaml`` and the calls it makes (services/segmentation.py:104-140):

  _resize_input_for_sam  (:104-110)  PIL BILINEAR -> 1024 x 1024 (aspect ignored)
  predictor.set_image    SAM2Transforms: ToTensor, Normalize(ImageNet) -> image_encoder (Hiera trunk + FpnNeck,
                         scalp 1) -> conv_s0 / conv_s1 on the two high-resolution levels, + no_mem_embed on the
                         stride-16 level (directly_add_no_mem_embed)
  predictor.predict      box = [0, 0, w, h] -> two corner points (labels 2, 3) + one padding point (label -1)
                         through the prompt encoder; no mask prompt -> no_mask_embed; mask decoder (two-way
                         transformer depth 2, high-res features, 4 mask tokens, multimask_output=False -> token 0:
                         the reference instantiates the yaml directly (segmentation.py:62-64), so build_sam2's
                         dynamic_multimask_via_stability override is NOT active)
                         -> 256 x 256 logits -> bilinear x4 (align_corners False) -> > mask_threshold (0.0)
  _resize_mask           (:112-118)  mask * 255 -> uint8 -> PIL NEAREST back to the thumbnail -> / 255

Parameter names follow the package's state dict (image_encoder.trunk.*, image_encoder.neck.*,
sam_prompt_encoder.*, sam_mask_decoder.*, no_mem_embed) so a real checkpoint can be dropped in; they are from the
public sources and unverified offline.  Structure notes that matter numerically:
  * Hiera: block i uses the PREVIOUS stage's window size in the first block of a stage ("lags by a block"), query
    pooling (2x2 max pool on q, and on the projected shortcut) in blocks 1, 3, 10; blocks 5, 7, 9 are global;
    window partition zero-pads AFTER norm1 (padded tokens take part in attention with qkv = bias).
  * positional embedding = bicubic-interpolated 7x7 background + tiled 8x8 window embedding.
  * FpnNeck: lateral 1x1 convs, top-down nearest x2 only into level 2 (stride 16); levels 0, 1 are lateral only.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)

STAGES = (1, 2, 7, 2)
EMBED = 96
WINDOW_SPEC = (8, 4, 14, 7)
GLOBAL_BLOCKS = (5, 7, 9)
BKG = (7, 7)


def block_plan():
    """[(dim_in, dim_out, heads, window, q_pool)] for the 12 Hiera-T blocks (hieradet.py Hiera.__init__)."""
    depth = sum(STAGES)
    stage_ends = [sum(STAGES[:i]) - 1 for i in range(1, len(STAGES) + 1)]
    q_pool_blocks = [x + 1 for x in stage_ends[:-1]][:3]
    plan, dim, heads, cur_stage = [], EMBED, 1, 1
    for i in range(depth):
        dim_out = dim
        window = WINDOW_SPEC[cur_stage - 1]                     # lags by a block
        if i in GLOBAL_BLOCKS:
            window = 0
        if i - 1 in stage_ends:
            dim_out, heads, cur_stage = dim * 2, heads * 2, cur_stage + 1
        plan.append((dim, dim_out, heads, window, i in q_pool_blocks))
        dim = dim_out
    return plan, stage_ends


def random_state_dict(seed: int = 0) -> dict:
    g = torch.Generator().manual_seed(seed)
    w = lambda *s, sc=0.02: torch.randn(*s, generator=g) * sc
    sd = {}
    lin = lambda name, o, i, sc=None: sd.update({name + ".weight": w(o, i, sc=sc or (1.0 / math.sqrt(i))), name + ".bias": w(o)})
    ln = lambda name, d: sd.update({name + ".weight": 1.0 + w(d, sc=0.1), name + ".bias": w(d)})
    t = "image_encoder.trunk."
    sd[t + "patch_embed.proj.weight"] = w(EMBED, 3, 7, 7, sc=0.08)
    sd[t + "patch_embed.proj.bias"] = w(EMBED)
    sd[t + "pos_embed"] = w(1, EMBED, *BKG, sc=0.2)
    sd[t + "pos_embed_window"] = w(1, EMBED, WINDOW_SPEC[0], WINDOW_SPEC[0], sc=0.2)
    plan, _ = block_plan()
    for i, (din, dout, heads, window, qpool) in enumerate(plan):
        b = f"{t}blocks.{i}."
        ln(b + "norm1", din)
        lin(b + "attn.qkv", 3 * dout, din)
        lin(b + "attn.proj", dout, dout)
        ln(b + "norm2", dout)
        lin(b + "mlp.layers.0", 4 * dout, dout)
        lin(b + "mlp.layers.1", dout, 4 * dout)
        if din != dout:
            lin(b + "proj", dout, din)
    for n, c in enumerate((768, 384, 192, 96)):
        sd[f"image_encoder.neck.convs.{n}.conv.weight"] = w(256, c, 1, 1, sc=1.0 / math.sqrt(c))
        sd[f"image_encoder.neck.convs.{n}.conv.bias"] = w(256)
    sd["no_mem_embed"] = w(1, 1, 256)
    p = "sam_prompt_encoder."
    sd[p + "pe_layer.positional_encoding_gaussian_matrix"] = torch.randn(2, 128, generator=g)
    for i in range(4):
        sd[p + f"point_embeddings.{i}.weight"] = w(1, 256, sc=0.5)
    sd[p + "not_a_point_embed.weight"] = w(1, 256, sc=0.5)
    sd[p + "no_mask_embed.weight"] = w(1, 256, sc=0.5)
    d = "sam_mask_decoder."

    def attn(name, internal):
        lin(name + ".q_proj", internal, 256); lin(name + ".k_proj", internal, 256)
        lin(name + ".v_proj", internal, 256); lin(name + ".out_proj", 256, internal)
    for l in range(2):
        b = f"{d}transformer.layers.{l}."
        attn(b + "self_attn", 256); attn(b + "cross_attn_token_to_image", 128); attn(b + "cross_attn_image_to_token", 128)
        for k in range(1, 5):
            ln(b + f"norm{k}", 256)
        lin(b + "mlp.layers.0", 2048, 256); lin(b + "mlp.layers.1", 256, 2048)
    attn(d + "transformer.final_attn_token_to_image", 128)
    ln(d + "transformer.norm_final_attn", 256)
    sd[d + "iou_token.weight"] = w(1, 256, sc=0.5)
    sd[d + "mask_tokens.weight"] = w(4, 256, sc=0.5)
    sd[d + "obj_score_token.weight"] = w(1, 256, sc=0.5)
    sd[d + "output_upscaling.0.weight"] = w(256, 64, 2, 2, sc=1.0 / 16); sd[d + "output_upscaling.0.bias"] = w(64)
    ln(d + "output_upscaling.1", 64)
    sd[d + "output_upscaling.3.weight"] = w(64, 32, 2, 2, sc=1.0 / 8); sd[d + "output_upscaling.3.bias"] = w(32)
    sd[d + "conv_s0.weight"] = w(32, 256, 1, 1, sc=1.0 / 16); sd[d + "conv_s0.bias"] = w(32)
    sd[d + "conv_s1.weight"] = w(64, 256, 1, 1, sc=1.0 / 16); sd[d + "conv_s1.bias"] = w(64)
    for i in range(4):
        b = f"{d}output_hypernetworks_mlps.{i}."
        lin(b + "layers.0", 256, 256); lin(b + "layers.1", 256, 256); lin(b + "layers.2", 32, 256)
    lin(d + "iou_prediction_head.layers.0", 256, 256); lin(d + "iou_prediction_head.layers.1", 256, 256)
    lin(d + "iou_prediction_head.layers.2", 4, 256)
    return sd


# ----------------------------------------------------------------------------- Hiera trunk
def window_partition(x, ws):
    b, h, w, c = x.shape
    ph, pw = (ws - h % ws) % ws, (ws - w % ws) % ws
    if ph or pw:
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
    hp, wp = h + ph, w + pw
    x = x.view(b, hp // ws, ws, wp // ws, ws, c)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, c), (hp, wp)


def window_unpartition(win, ws, pad_hw, hw):
    hp, wp = pad_hw
    h, w = hw
    b = win.shape[0] // (hp * wp // ws // ws)
    x = win.reshape(b, hp // ws, wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(b, hp, wp, -1)
    return x[:, :h, :w, :]


def do_pool(x):                                        # [B, H, W, C] max pool 2x2 stride 2
    return F.max_pool2d(x.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)


def hiera_pos_embed(sd, h, w):
    t = "image_encoder.trunk."
    pos = F.interpolate(sd[t + "pos_embed"], size=(h, w), mode="bicubic")
    win = sd[t + "pos_embed_window"]
    pos = pos + win.tile([x // y for x, y in zip(pos.shape, win.shape)])
    return pos.permute(0, 2, 3, 1)


@torch.inference_mode()
def hiera_forward(sd, x):
    """x [1, 3, 1024, 1024] normalised -> feature maps [1, C, H, W] at strides 4, 8, 16, 32."""
    t = "image_encoder.trunk."
    x = F.conv2d(x, sd[t + "patch_embed.proj.weight"], sd[t + "patch_embed.proj.bias"], stride=4, padding=3)
    x = x.permute(0, 2, 3, 1)
    x = x + hiera_pos_embed(sd, x.shape[1], x.shape[2])
    plan, stage_ends = block_plan()
    outs = []
    for i, (din, dout, heads, window, qpool) in enumerate(plan):
        b = f"{t}blocks.{i}."
        shortcut = x
        xn = F.layer_norm(x, (din,), sd[b + "norm1.weight"], sd[b + "norm1.bias"], 1e-6)
        if din != dout:
            shortcut = F.linear(xn, sd[b + "proj.weight"], sd[b + "proj.bias"])
            if qpool:
                shortcut = do_pool(shortcut)
        h, w = xn.shape[1], xn.shape[2]
        ws = window
        if ws > 0:
            xn, pad_hw = window_partition(xn, ws)
        bb, hh, ww, _ = xn.shape
        qkv = F.linear(xn, sd[b + "attn.qkv.weight"], sd[b + "attn.qkv.bias"]).reshape(bb, hh * ww, 3, heads, -1)
        q, k, v = torch.unbind(qkv, 2)
        if qpool:
            q = do_pool(q.reshape(bb, hh, ww, -1))
            hh, ww = q.shape[1:3]
            q = q.reshape(bb, hh * ww, heads, -1)
        a = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
        a = a.transpose(1, 2).reshape(bb, hh, ww, -1)
        a = F.linear(a, sd[b + "attn.proj.weight"], sd[b + "attn.proj.bias"])
        if qpool:
            ws = window // 2
            h, w = shortcut.shape[1:3]
            pad_hw = (h + (ws - h % ws) % ws, w + (ws - w % ws) % ws) if ws > 0 else None
        if window > 0:
            a = window_unpartition(a, ws, pad_hw, (h, w))
        x = shortcut + a
        xn2 = F.layer_norm(x, (dout,), sd[b + "norm2.weight"], sd[b + "norm2.bias"], 1e-6)
        m = F.linear(F.gelu(F.linear(xn2, sd[b + "mlp.layers.0.weight"], sd[b + "mlp.layers.0.bias"])),
                     sd[b + "mlp.layers.1.weight"], sd[b + "mlp.layers.1.bias"])
        x = x + m
        if i in stage_ends:
            outs.append(x.permute(0, 3, 1, 2))
    return outs


@torch.inference_mode()
def image_features(sd, img_u8_1024: np.ndarray):
    """set_image: -> (image_embed [1,256,64,64], feat_s0 [1,32,256,256], feat_s1 [1,64,128,128])."""
    x = torch.from_numpy(np.ascontiguousarray(img_u8_1024)).permute(2, 0, 1).float().div(255)[None]
    x = (x - torch.tensor(MEAN).view(1, 3, 1, 1)) / torch.tensor(STD).view(1, 3, 1, 1)
    xs = hiera_forward(sd, x)                                   # strides 4, 8, 16, 32: 96, 192, 384, 768 channels
    n = 3
    out = [None] * 4
    prev = None
    for i in range(n, -1, -1):
        lat = F.conv2d(xs[i], sd[f"image_encoder.neck.convs.{n - i}.conv.weight"], sd[f"image_encoder.neck.convs.{n - i}.conv.bias"])
        if i in (2, 3) and prev is not None:
            prev = lat + F.interpolate(prev, scale_factor=2.0, mode="nearest")
        else:
            prev = lat
        out[i] = prev
    feats = out[:-1]                                            # scalp = 1
    d = "sam_mask_decoder."
    s0 = F.conv2d(feats[0], sd[d + "conv_s0.weight"], sd[d + "conv_s0.bias"])
    s1 = F.conv2d(feats[1], sd[d + "conv_s1.weight"], sd[d + "conv_s1.bias"])
    embed = feats[2] + sd["no_mem_embed"].view(1, 256, 1, 1)
    return embed, s0, s1


# ----------------------------------------------------------------------------- prompt encoder (constants of the model)
def pe_encoding(sd, coords01):
    g = sd["sam_prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"]
    c = (2 * coords01 - 1) @ g
    c = 2 * math.pi * c
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1)


def box_prompt_tokens(sd, size: int = 1024):
    """Sparse embeddings [3, 256] for box [0, 0, size, size] (+ padding point) and the dense PE [4096, 256]."""
    p = "sam_prompt_encoder."
    # _embed_points: corners shifted by +0.5; the padding point (0, 0) is appended after the shift and its
    # positional encoding is zeroed (label -1) before not_a_point_embed is added
    pts = torch.tensor([[0.5, 0.5], [size + 0.5, size + 0.5], [0.0, 0.0]])
    emb = pe_encoding(sd, pts / float(size))
    emb[2] = 0.0
    emb[2] += sd[p + "not_a_point_embed.weight"][0]
    emb[0] += sd[p + "point_embeddings.2.weight"][0]
    emb[1] += sd[p + "point_embeddings.3.weight"][0]
    grid = (torch.arange(64, dtype=torch.float32) + 0.5) / 64
    yy, xx = torch.meshgrid(grid, grid, indexing="ij")
    dense_pe = pe_encoding(sd, torch.stack([xx, yy], dim=-1)).reshape(4096, 256)
    return emb, dense_pe


# ----------------------------------------------------------------------------- mask decoder
def _attn(sd, name, q, k, v, heads=8):
    q = F.linear(q, sd[name + ".q_proj.weight"], sd[name + ".q_proj.bias"])
    k = F.linear(k, sd[name + ".k_proj.weight"], sd[name + ".k_proj.bias"])
    v = F.linear(v, sd[name + ".v_proj.weight"], sd[name + ".v_proj.bias"])
    sep = lambda x: x.reshape(x.shape[0], heads, -1).transpose(0, 1)               # [T, C] -> [H, T, d]
    o = F.scaled_dot_product_attention(sep(q)[None], sep(k)[None], sep(v)[None])[0]
    o = o.transpose(0, 1).reshape(q.shape[0], -1)
    return F.linear(o, sd[name + ".out_proj.weight"], sd[name + ".out_proj.bias"])


@torch.inference_mode()
def mask_decoder(sd, embed, s0, s1):
    """-> low-resolution mask logits [256, 256] of mask token 0 (multimask_output=False)."""
    d = "sam_mask_decoder."
    sparse, image_pe = box_prompt_tokens(sd)
    tokens = torch.cat([sd[d + "obj_score_token.weight"], sd[d + "iou_token.weight"], sd[d + "mask_tokens.weight"], sparse], 0)
    src = embed[0].flatten(1).t() + sd["sam_prompt_encoder.no_mask_embed.weight"]       # [4096, 256]
    queries, keys = tokens, src
    ln = lambda x, n: F.layer_norm(x, (256,), sd[n + ".weight"], sd[n + ".bias"], 1e-5)
    for l in range(2):
        b = f"{d}transformer.layers.{l}."
        if l == 0:
            queries = _attn(sd, b + "self_attn", queries, queries, queries)
        else:
            q = queries + tokens
            queries = queries + _attn(sd, b + "self_attn", q, q, queries)
        queries = ln(queries, b + "norm1")
        q, k = queries + tokens, keys + image_pe
        queries = ln(queries + _attn(sd, b + "cross_attn_token_to_image", q, k, keys), b + "norm2")
        m = F.linear(F.relu(F.linear(queries, sd[b + "mlp.layers.0.weight"], sd[b + "mlp.layers.0.bias"])),
                     sd[b + "mlp.layers.1.weight"], sd[b + "mlp.layers.1.bias"])
        queries = ln(queries + m, b + "norm3")
        q, k = queries + tokens, keys + image_pe
        keys = ln(keys + _attn(sd, b + "cross_attn_image_to_token", k, q, queries), b + "norm4")
    q, k = queries + tokens, keys + image_pe
    queries = ln(queries + _attn(sd, d + "transformer.final_attn_token_to_image", q, k, keys), d + "transformer.norm_final_attn")
    mask_token = queries[2]                                                            # obj, iou, mask0..3, prompts
    src = keys.t().reshape(1, 256, 64, 64)
    up = F.conv_transpose2d(src, sd[d + "output_upscaling.0.weight"], sd[d + "output_upscaling.0.bias"], stride=2) + s1
    u = up.permute(0, 2, 3, 1)
    u = F.layer_norm(u, (64,), sd[d + "output_upscaling.1.weight"], sd[d + "output_upscaling.1.bias"], 1e-6)
    up = F.gelu(u.permute(0, 3, 1, 2))
    up = F.gelu(F.conv_transpose2d(up, sd[d + "output_upscaling.3.weight"], sd[d + "output_upscaling.3.bias"], stride=2) + s0)
    h = mask_token
    b = d + "output_hypernetworks_mlps.0."
    h = F.relu(F.linear(h, sd[b + "layers.0.weight"], sd[b + "layers.0.bias"]))
    h = F.relu(F.linear(h, sd[b + "layers.1.weight"], sd[b + "layers.1.bias"]))
    h = F.linear(h, sd[b + "layers.2.weight"], sd[b + "layers.2.bias"])                # [32]
    return (h @ up[0].flatten(1)).reshape(256, 256)


@torch.inference_mode()
def predict_logits(sd, img_u8_1024: np.ndarray) -> np.ndarray:
    embed, s0, s1 = image_features(sd, img_u8_1024)
    return mask_decoder(sd, embed, s0, s1).numpy()


def logits_to_mask(logits256: np.ndarray, threshold: float = 0.0) -> np.ndarray:
    """postprocess_masks: bilinear x4 (align_corners False), then > mask_threshold -> float32 {0, 1} [1024, 1024]."""
    up = F.interpolate(torch.from_numpy(logits256)[None, None], (1024, 1024), mode="bilinear", align_corners=False)[0, 0]
    return (up > threshold).numpy().astype(np.float32)


def predict_image(sd, thumb_u8: np.ndarray, threshold: float = 0.0) -> np.ndarray:
    """services/segmentation.py:120-140 (predict_image with resize_to_input=True)."""
    from PIL import Image
    h, w = thumb_u8.shape[:2]
    img = thumb_u8 if (h, w) == (1024, 1024) else np.array(Image.fromarray(thumb_u8).resize((1024, 1024), Image.Resampling.BILINEAR))
    mask = logits_to_mask(predict_logits(sd, img), threshold)
    if (h, w) != (1024, 1024):
        mask = np.asarray(Image.fromarray((mask * 255).astype(np.uint8), mode="L").resize((w, h), Image.Resampling.NEAREST),
                          dtype=np.float32) / 255.0
    return mask
