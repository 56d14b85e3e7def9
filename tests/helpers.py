"""Shared test helpers (inputs that reproduce the golden generator's seeds)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_coords_cases():
    arrays = np.load(os.path.join(GOLDEN, "coords_cases.npz"))
    with open(os.path.join(GOLDEN, "coords_cases.json")) as fh:
        meta = json.load(fh)
    cases = {}
    for name, info in meta.items():
        shape = tuple(int(v) for v in arrays[f"{name}__mask_shape"])
        bits = np.unpackbits(arrays[f"{name}__mask_bits"])[: shape[0] * shape[1]]
        mask = bits.reshape(shape).astype(np.float32)
        cases[name] = dict(mask=mask, coords=arrays[f"{name}__coords"], info=info)
    return cases


def load_contour_case(name):
    arrays = np.load(os.path.join(GOLDEN, "contours_cases.npz"))
    lens = arrays[f"{name}__lens"]
    pts = arrays[f"{name}__pts"]
    nholes = arrays[f"{name}__nholes"]
    polys, off = [], 0
    for n in lens:
        polys.append(pts[off:off + n])
        off += n
    n_t = len(nholes)
    tissue = polys[:n_t]
    holes, k = [], n_t
    for nh in nholes:
        holes.append(polys[k:k + nh])
        k += nh
    return tissue, holes


def golden_patches(tag_ns):
    """Patches exactly as tests/golden/gen_golden.py drew them: ONE rng per tag, sequential ns."""
    rng = np.random.default_rng(0)
    out = {}
    for n in tag_ns:
        out[n] = [rng.integers(0, 256, (256, 256, 3), dtype=np.uint8) for _ in range(n)]
    return out


def canonical_to_hf(sd, depth):
    """Canonical (ap_vit_set_param) names -> HF ViTModel names the oracle consumes."""
    hf = {"embeddings.patch_embeddings.projection.weight": sd["patch_embed.weight"],
          "embeddings.patch_embeddings.projection.bias": sd["patch_embed.bias"],
          "embeddings.cls_token": sd["cls_token"].view(1, 1, -1),
          "embeddings.position_embeddings": sd["pos_embed"][None],
          "layernorm.weight": sd["norm.weight"], "layernorm.bias": sd["norm.bias"]}
    for i in range(depth):
        p, b = f"layers.{i}.", f"blocks.{i}."
        q, k, v = sd[b + "qkv.weight"].chunk(3, 0)
        qb, kb, vb = sd[b + "qkv.bias"].chunk(3, 0)
        hf.update({p + "layernorm_before.weight": sd[b + "ln1.weight"], p + "layernorm_before.bias": sd[b + "ln1.bias"],
                   p + "attention.q_proj.weight": q, p + "attention.q_proj.bias": qb,
                   p + "attention.k_proj.weight": k, p + "attention.k_proj.bias": kb,
                   p + "attention.v_proj.weight": v, p + "attention.v_proj.bias": vb,
                   p + "attention.o_proj.weight": sd[b + "proj.weight"], p + "attention.o_proj.bias": sd[b + "proj.bias"],
                   p + "layernorm_after.weight": sd[b + "ln2.weight"], p + "layernorm_after.bias": sd[b + "ln2.bias"],
                   p + "mlp.fc1.weight": sd[b + "fc1.weight"], p + "mlp.fc1.bias": sd[b + "fc1.bias"],
                   p + "mlp.fc2.weight": sd[b + "fc2.weight"], p + "mlp.fc2.bias": sd[b + "fc2.bias"]})
    return hf
