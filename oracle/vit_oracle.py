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
