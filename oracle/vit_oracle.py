"""CPU oracle: patch preprocess + ViT encoder forward in explicit fp32 ops.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Floating-point kernel, so the
oracle is a plain torch-CPU fp32 restatement of what the reference makes PyTorch
run:

  preprocess_center_crop   /root/reference/atlas_patch/models/patch/base.py:42-45 with the
                           torchvision ``ImageClassification(crop=224, resize=256)`` transform
                           [3P torchvision, unpinned]: centre crop, ``x/255``, ``(x-mean)/std``
  extract_batch            base.py:76-107 (batching remainder, concat order, f32 output)
  vit_forward              the module the reference's ``vit_b_16`` / ``uni_v1`` wrap
                           (models/patch/vit.py:9-38, uni.py:13-60): conv patch-embed, CLS +
                           pos-embed, pre-LN blocks (MHA + erf-GELU MLP, optional LayerScale),
                           final LN, CLS token.

Pinned: ``tests/golden/extract_batch.npz`` holds outputs of the reference's own
``PatchFeatureExtractor.extract_batch`` driving a seeded HF ``ViTModel`` (the only
ViT implementation importable in this image); this file reproduces them from the
same seeded state dict.  torchvision / timm checkpoints: same math, different key
names (unpinned, no weights offline).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def preprocess_center_crop(patches_u8: np.ndarray, crop: int = 224,
                           mean=IMAGENET_MEAN, std=IMAGENET_STD) -> torch.Tensor:
    """uint8 [n, H, W, 3] -> float32 [n, 3, crop, crop]; y = ((x / 255) - mean) / std."""
    x = torch.from_numpy(np.ascontiguousarray(patches_u8))
    n, h, w, _ = x.shape
    top = int(round((h - crop) / 2.0))
    left = int(round((w - crop) / 2.0))
    x = x[:, top:top + crop, left:left + crop, :].permute(0, 3, 1, 2).to(torch.float32).div(255)
    m = torch.tensor(mean, dtype=torch.float32).view(1, 3, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(1, 3, 1, 1)
    return x.sub(m).div(s)


def make_hf_vit(layers: int = 12, hidden: int = 768, heads: int = 12, mlp: int = 3072,
                seed: int = 0, eps: float = 1e-6, image_size: int = 224):
    """The seeded HF ViTModel used by tests/golden/gen_golden.py (same construction)."""
    from transformers import ViTConfig, ViTModel

    torch.manual_seed(seed)
    cfg = ViTConfig(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
                    intermediate_size=mlp, image_size=image_size, patch_size=16,
                    layer_norm_eps=eps, hidden_act="gelu")
    model = ViTModel(cfg, add_pooling_layer=False).eval()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "layernorm" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif name.endswith("bias") or "position_embeddings" in name or "cls_token" in name:
                p.copy_(0.02 * torch.randn(p.shape, generator=g))
    return model


def _layer_keys(sd: dict, layer: int):
    """Per-layer key prefixes for the two HF ViTModel naming schemes (transformers 4.x / 5.x)."""
    p = f"layers.{layer}."
    if p + "layernorm_before.weight" in sd:                       # transformers >= 5
        a = p + "attention."
        return (p, a + "q_proj", a + "k_proj", a + "v_proj", a + "o_proj", p + "mlp.fc1", p + "mlp.fc2")
    p = f"encoder.layer.{layer}."
    if p + "layernorm_before.weight" in sd:                       # transformers 4.x
        a = p + "attention.attention."
        return (p, a + "query", a + "key", a + "value", p + "attention.output.dense",
                p + "intermediate.dense", p + "output.dense")
    return None


@torch.inference_mode()
def vit_forward_hf(sd: dict, x: torch.Tensor, *, heads: int, eps: float = 1e-6,
                   layer_scale: dict | None = None) -> torch.Tensor:
    """Explicit-op ViT forward from an HF ``ViTModel`` state dict.  x: [n,3,H,W] f32 -> [n, D]."""
    w = sd["embeddings.patch_embeddings.projection.weight"]
    b = sd["embeddings.patch_embeddings.projection.bias"]
    d = w.shape[0]
    pe = F.conv2d(x, w, b, stride=w.shape[-1]).flatten(2).transpose(1, 2)          # [n, P, D]
    n = x.shape[0]
    tok = torch.cat([sd["embeddings.cls_token"].expand(n, -1, -1), pe], dim=1)
    tok = tok + sd["embeddings.position_embeddings"]
    dh = d // heads
    layer = 0
    while True:
        names = _layer_keys(sd, layer)
        if names is None:
            break
        p, q_, k_, v_, o_, f1, f2 = names
        h = F.layer_norm(tok, (d,), sd[p + "layernorm_before.weight"], sd[p + "layernorm_before.bias"], eps)
        q = h @ sd[q_ + ".weight"].T + sd[q_ + ".bias"]
        k = h @ sd[k_ + ".weight"].T + sd[k_ + ".bias"]
        v = h @ sd[v_ + ".weight"].T + sd[v_ + ".bias"]
        t = tok.shape[1]
        q = q.view(n, t, heads, dh).transpose(1, 2)
        k = k.view(n, t, heads, dh).transpose(1, 2)
        v = v.view(n, t, heads, dh).transpose(1, 2)
        att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(dh), dim=-1)
        ctx = (att @ v).transpose(1, 2).reshape(n, t, d)
        out = ctx @ sd[o_ + ".weight"].T + sd[o_ + ".bias"]
        if layer_scale is not None:
            out = out * layer_scale[f"ls1.{layer}"]
        tok = tok + out
        h = F.layer_norm(tok, (d,), sd[p + "layernorm_after.weight"], sd[p + "layernorm_after.bias"], eps)
        m = h @ sd[f1 + ".weight"].T + sd[f1 + ".bias"]
        m = F.gelu(m)                                                              # exact erf
        m = m @ sd[f2 + ".weight"].T + sd[f2 + ".bias"]
        if layer_scale is not None:
            m = m * layer_scale[f"ls2.{layer}"]
        tok = tok + m
        layer += 1
    tok = F.layer_norm(tok, (d,), sd["layernorm.weight"], sd["layernorm.bias"], eps)
    return tok[:, 0]


def extract_batch(sd: dict, patches, *, heads: int, batch_size: int | None = None,
                  eps: float = 1e-6, layer_scale: dict | None = None) -> np.ndarray:
    """base.py:76-107: empty -> (0, D); chunks of min(len, batch_size); concat; float32 numpy."""
    d = sd["embeddings.patch_embeddings.projection.weight"].shape[0]
    if len(patches) == 0:
        return np.empty((0, d), dtype=np.float32)
    bs = min(len(patches), batch_size or len(patches))
    outs = []
    for s in range(0, len(patches), bs):
        chunk = np.stack([np.asarray(p) for p in patches[s:s + bs]], 0)
        x = preprocess_center_crop(chunk)
        outs.append(vit_forward_hf(sd, x, heads=heads, eps=eps, layer_scale=layer_scale))
    return torch.cat(outs, 0).to(torch.float32).numpy()


# ----------------------------------------------------------------------------- CONCH v1 visual tower
# models/patch/conch.py:20-64 calls conch.open_clip_custom.create_model_from_pretrained("conch_ViT-B-16") and
# ``model.encode_image(x, proj_contrast=False, normalize=False)``.  The ``conch`` package is absent here (parity
# unpinned); this restates its published structure: timm ViT-B/16 trunk at 448 px (forward_features: all tokens,
# final LayerNorm) -> open_clip ``AttentionalPooler`` with one query -> LayerNorm.  The pooler's attention is run by
# torch's own ``multi_head_attention_forward`` with separate q/k/v projection weights, i.e. exactly what the
# ``nn.MultiheadAttention(d_model, n_head, kdim=context_dim, vdim=context_dim)`` inside that module executes.
OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


@torch.inference_mode()
def vit_tokens_canonical(sd: dict, x: torch.Tensor, *, heads: int, depth: int, eps: float = 1e-6, act: str = "gelu") -> torch.Tensor:
    """timm ``VisionTransformer.forward_features`` from canonical parameter names: [n,3,S,S] -> [n, prefix+P, D].

    Covers what the reference's encoder files instantiate (models/patch/vit.py:9-15, uni.py:13-125):
      * ``reg_tokens`` [R, D] present: register tokens between the class token and the patches (timm ``reg_tokens``);
      * ``pos_embed`` with exactly P rows: timm ``no_embed_class`` -- the position embedding is added to the patch tokens
        only (``_pos_embed``: x = x + pos_embed, THEN the prefix tokens are concatenated); with prefix + P rows it is added
        to every token after the concatenation (torchvision, timm default);
      * ``fc1.weight`` with twice as many rows as ``fc2.weight`` has columns: timm ``SwiGLUPacked`` (GluMlp with
        gate_last=False): x1, x2 = fc1(x).chunk(2, -1); fc2(silu(x1) * x2); otherwise fc2(gelu_erf(fc1(x)));
      * ``ls1`` / ``ls2``: LayerScale;
      * ``pre_norm.weight`` present: CLIP's ln_pre -- LayerNorm on the embedded tokens before the first block (open_clip
        VisionTransformer.forward, transformers CLIPVisionTransformer.pre_layrnorm); ``act="quick_gelu"``: x * sigmoid(1.702 x)
        (the OpenAI CLIP weights; models/patch/clip.py, plip.py, quilt.py).
    Pinned against transformers' ViTModel (patch 16 / 32 / 14, 80-wide heads) and Dinov2WithRegistersModel (register tokens,
    SwiGLU, LayerScale) by tests/test_encoder_zoo.py."""
    w, b = sd["patch_embed.weight"], sd["patch_embed.bias"]
    d = w.shape[0]
    n = x.shape[0]
    pe = F.conv2d(x, w, b, stride=w.shape[-1]).flatten(2).transpose(1, 2)
    prefix = [sd["cls_token"].view(1, 1, d).expand(n, -1, -1)]
    if "reg_tokens" in sd:
        prefix.append(sd["reg_tokens"].view(1, -1, d).expand(n, -1, -1))
    pos = sd["pos_embed"]
    if pos.shape[0] == pe.shape[1]:
        tok = torch.cat(prefix + [pe + pos[None]], dim=1)
    else:
        tok = torch.cat(prefix + [pe], dim=1) + pos[None]
    dh = d // heads
    if "pre_norm.weight" in sd:
        tok = F.layer_norm(tok, (d,), sd["pre_norm.weight"], sd["pre_norm.bias"], eps)
    for i in range(depth):
        p = f"blocks.{i}."
        h = F.layer_norm(tok, (d,), sd[p + "ln1.weight"], sd[p + "ln1.bias"], eps)
        qkv = h @ sd[p + "qkv.weight"].T + sd[p + "qkv.bias"]
        t = tok.shape[1]
        q, k, v = qkv.view(n, t, 3, heads, dh).permute(2, 0, 3, 1, 4)
        if "rope.cos" in sd:
            # transformers DINOv3ViT apply_rotary_pos_emb (models/patch/dinov3.py -> AutoModel): patch tokens only
            cos, sin = sd["rope.cos"], sd["rope.sin"]
            pre = t - cos.shape[0]

            def rot(x):
                xp = x[:, :, pre:]
                x1, x2 = xp[..., : dh // 2], xp[..., dh // 2:]
                return torch.cat([x[:, :, :pre], xp * cos + torch.cat([-x2, x1], dim=-1) * sin], dim=2)
            q, k = rot(q), rot(k)
        att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(dh), dim=-1)
        ctx = (att @ v).transpose(1, 2).reshape(n, t, d)
        out = ctx @ sd[p + "proj.weight"].T + sd[p + "proj.bias"]
        if p + "ls1" in sd:
            out = out * sd[p + "ls1"]
        tok = tok + out
        h = F.layer_norm(tok, (d,), sd[p + "ln2.weight"], sd[p + "ln2.bias"], eps)
        m = h @ sd[p + "fc1.weight"].T + sd[p + "fc1.bias"]
        if sd[p + "fc1.weight"].shape[0] == 2 * sd[p + "fc2.weight"].shape[1]:
            x1, x2 = m.chunk(2, dim=-1)
            m = F.silu(x1) * x2
        elif act == "quick_gelu":
            m = m * torch.sigmoid(1.702 * m)
        else:
            m = F.gelu(m)
        m = m @ sd[p + "fc2.weight"].T + sd[p + "fc2.bias"]
        if p + "ls2" in sd:
            m = m * sd[p + "ls2"]
        tok = tok + m
    return F.layer_norm(tok, (d,), sd["norm.weight"], sd["norm.bias"], eps)


def transform_resize_crop(patches_u8, *, resize, crop: int, mean=IMAGENET_MEAN, std=IMAGENET_STD) -> torch.Tensor:
    """torchvision ``ImageClassification`` / timm ``create_transform`` on a PIL tile (base.py:42-45 hands the transform a PIL
    image): Resize(shorter side -> size, Pillow filter) unless the shorter side already has it, CenterCrop(crop), ToTensor,
    Normalize.  resize = (size, "bilinear" | "bicubic") or None."""
    from PIL import Image
    arrs = []
    for p in patches_u8:
        img = Image.fromarray(np.asarray(p))
        if resize is not None:
            size, filt = resize
            w, h = img.size
            if min(w, h) != size:
                nw, nh = (size, int(size * h / w)) if w <= h else (int(size * w / h), size)
                img = img.resize((nw, nh), Image.Resampling.BICUBIC if filt == "bicubic" else Image.Resampling.BILINEAR)
        arrs.append(np.asarray(img))
    return preprocess_center_crop(np.stack(arrs, 0), crop=crop, mean=mean, std=std)


@torch.inference_mode()
def canonical_extract(sd: dict, patches_u8, *, heads: int, depth: int, image_size: int, resize=None, eps: float = 1e-6,
                      batch: int = 8, pool: str = "cls", mean=IMAGENET_MEAN, std=IMAGENET_STD, act: str = "gelu") -> np.ndarray:
    """extract_batch of an encoder from canonical parameters: transform -> tokens -> LN -> token 0 (``pool="cls"``) or
    torch.cat([token 0, patch tokens.mean(1)], -1) (``pool="cls_mean"``: models/patch/midnight.py:58-61, virchow.py:58-61;
    register tokens are not patch tokens, virchow.py:111-114)."""
    outs = []
    prefix = 1 + (sd["reg_tokens"].shape[0] if "reg_tokens" in sd else 0)
    for s in range(0, len(patches_u8), batch):
        x = transform_resize_crop(patches_u8[s:s + batch], resize=resize, crop=image_size, mean=mean, std=std)
        tok = vit_tokens_canonical(sd, x, heads=heads, depth=depth, eps=eps, act=act)
        feat = tok[:, 0] if pool == "cls" else torch.cat([tok[:, 0], tok[:, prefix:].mean(1)], dim=-1)
        if "head_proj.weight" in sd:                     # CLIP: pooled @ visual projection (no bias)
            feat = feat @ sd["head_proj.weight"].T
        outs.append(feat)
    return torch.cat(outs, 0).to(torch.float32).numpy()


@torch.inference_mode()
def attentional_pooler(pool: dict, tokens: torch.Tensor, *, n_head: int, eps: float = 1e-5) -> torch.Tensor:
    """open_clip ``AttentionalPooler.forward`` (transformer.py): tokens [n, L, C] -> [n, n_queries, P]."""
    c = tokens.shape[-1]
    p_dim = pool["attn.q_proj_weight"].shape[0]
    x = F.layer_norm(tokens, (c,), pool["ln_k.weight"], pool["ln_k.bias"], eps).permute(1, 0, 2)     # NLD -> LND
    n = x.shape[1]
    q = F.layer_norm(pool["query"], (p_dim,), pool["ln_q.weight"], pool["ln_q.bias"], eps)
    out, _ = F.multi_head_attention_forward(
        q.unsqueeze(1).expand(-1, n, -1), x, x, p_dim, n_head, None, pool["attn.in_proj_bias"], None, None, False, 0.0,
        pool["attn.out_proj.weight"], pool["attn.out_proj.bias"], training=False, need_weights=False,
        use_separate_proj_weight=True, q_proj_weight=pool["attn.q_proj_weight"],
        k_proj_weight=pool["attn.k_proj_weight"], v_proj_weight=pool["attn.v_proj_weight"])
    return out.permute(1, 0, 2)


def conch_preprocess(patches_u8, size: int = 448) -> torch.Tensor:
    """open_clip image_transform(is_train=False): Resize(size, bicubic) + CenterCrop(size) + ToTensor + Normalize."""
    from PIL import Image
    arrs = []
    for p in patches_u8:
        img = Image.fromarray(np.asarray(p))
        w, h = img.size
        if min(w, h) != size:
            nw, nh = (size, int(size * h / w)) if w <= h else (int(size * w / h), size)
            img = img.resize((nw, nh), Image.Resampling.BICUBIC)
        arrs.append(np.asarray(img))
    return preprocess_center_crop(np.stack(arrs, 0), crop=size, mean=OPENAI_CLIP_MEAN, std=OPENAI_CLIP_STD)


@torch.inference_mode()
def conch_encode_image(trunk_sd: dict, pool: dict, patches_u8, *, heads: int = 12, depth: int = 12,
                       pool_heads: int = 8, size: int = 448) -> np.ndarray:
    """``encode_image(x, proj_contrast=False, normalize=False)``: LN(attn_pool_contrast(trunk(x))[:, 0])."""
    x = conch_preprocess(patches_u8, size)
    tokens = vit_tokens_canonical(trunk_sd, x, heads=heads, depth=depth)
    pooled = attentional_pooler(pool, tokens, n_head=pool_heads)[:, 0]
    p_dim = pooled.shape[-1]
    return F.layer_norm(pooled, (p_dim,), pool["ln_out.weight"], pool["ln_out.bias"], 1e-5).numpy()
