#!/usr/bin/env python3
"""bench.py -- patches/sec embedded (256x256 tiles, ViT-B/16) on N MI355X.

Workload: `process` on one synthetic 100 000 x 100 000 slide per rank, ViT-B/16 -- the slide BASELINE.json's north star
quotes its target on, at EVERY N, so the driver's 1 -> 8 curve is one workload string (configs[3] per rank; configs[1]'s
40 000^2 slide: `--slide 40000` -- the step is the same 2048 HBM-resident tiles, only the tile pool and the coordinate
block differ; its end-to-end rate is `rates.e2e_cli_40k`).  Tiles are the slide's own tissue tiles (coords from the device coordinate
path, pixels rendered into HBM by ap_synth_tiles) -- resident in HBM before the timed region.
A "step" = one pass of the hot path over one device batch of tiles:
    uint8 HWC tiles -> K1 normalise/crop/patch-rows -> ViT-B/16 (12 blocks) -> float32 [B, 768].
Weights: seeded random init of the ViT-B/16 architecture (no checkpoints offline).

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched by
torch.distributed.run, one rank per GPU (RCCL).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

FLOP_PER_PATCH_VIT_B16 = 35.126e9      # 33.695 GEMM (incl. 0.231 patch-embed) + 1.431 attention
OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
# --encoder: the three encoders BASELINE.json's configs name (registered names of the reference: models/patch/vit.py:9-15,
# uni.py:13-60, conch.py:20-64); `batch` = default device batch
ENCODERS = {
    "vit_b_16": {"label": "ViT-B/16", "batch": 2048},
    "uni_v1": {"label": "UNI v1 (ViT-L/16 + LayerScale, Resize(224, bicubic) on the device)", "batch": 2048},
    "conch_v1": {"label": "CONCH v1 visual tower (ViT-B/16 at 448 px = 785 tokens + attentional pooler, Resize(448, bicubic) "
                          "on the device)", "batch": 256, "mean": OPENAI_CLIP_MEAN, "std": OPENAI_CLIP_STD},
    # the rest of the reference's three encoder files (round 4)
    "vit_b_32": {"label": "ViT-B/32 (50 tokens)", "batch": 4096},
    "vit_l_32": {"label": "ViT-L/32 (50 tokens)", "batch": 4096},
    "vit_h_14": {"label": "ViT-H/14 at torchvision's SWAG_E2E 518 px (1370 tokens, 80-wide heads stored 96 wide, Resize(518, "
                          "bicubic) on the device)", "batch": 128},
    "uni_v2": {"label": "UNI2-h (ViT-H/14 at 224 px, 8 register tokens, SwiGLU, LayerScale; Resize(224, bicubic) on the device)",
               "batch": 1024},
    # the transformers-backed encoder files (models/patch/dinov2.py, phikon.py)
    "dinov2_small": {"label": "DINOv2 ViT-S/14 (257 tokens, LayerScale)", "batch": 4096},
    "dinov2_base": {"label": "DINOv2 ViT-B/14 (257 tokens, LayerScale)", "batch": 2048},
    "dinov2_large": {"label": "DINOv2 ViT-L/14 (257 tokens, LayerScale)", "batch": 1024},
    "dinov2_giant": {"label": "DINOv2 ViT-g/14 (257 tokens, 40 blocks, SwiGLU, LayerScale)", "batch": 512},
    "phikon_v1": {"label": "Phikon (HF ViT-B/16, LayerNorm 1e-12; Resize(224, bilinear) on the device)", "batch": 2048},
    "phikon_v2": {"label": "Phikon-v2 (HF DINOv2 ViT-L/16; Resize(224, bicubic) on the device)", "batch": 2048},
    # midnight.py (HF DINOv2 ViT-g/14, class token | mean patch token) and the timm-hub ViTs
    "midnight": {"label": "Midnight (HF DINOv2 ViT-g/14, class token + mean patch token = 3072-d)", "batch": 512,
                 "mean": (0.5, 0.5, 0.5), "std": (0.5, 0.5, 0.5)},
    "h_optimus_0": {"label": "H-optimus-0 (ViT-g/14, 4 register tokens, SwiGLU)", "batch": 512,
                    "mean": (0.707223, 0.578729, 0.703617), "std": (0.211883, 0.230117, 0.177517)},
    "prov_gigapath": {"label": "Prov-GigaPath tile encoder (ViT-g/16, SwiGLU)", "batch": 512},
    "lunit_vit_small_patch16_dino": {"label": "Lunit ViT-S/16 DINO", "batch": 4096},
    "lunit_vit_small_patch8_dino": {"label": "Lunit ViT-S/8 DINO (785 tokens)", "batch": 512},
    "pathorchestra": {"label": "PathOrchestra (ViT-L/16 + LayerScale)", "batch": 2048},
    # CLIP vision towers (clip.py, plip.py, quilt.py): ln_pre, QuickGELU, projection
    "clip_vit_b_32": {"label": "CLIP ViT-B/32 image tower (encode_image, 512-d)", "batch": 4096, "mean": OPENAI_CLIP_MEAN, "std": OPENAI_CLIP_STD},
    "clip_vit_b_16": {"label": "CLIP ViT-B/16 image tower (encode_image, 512-d)", "batch": 2048, "mean": OPENAI_CLIP_MEAN, "std": OPENAI_CLIP_STD},
    "clip_vit_l_14": {"label": "CLIP ViT-L/14 image tower (encode_image, 768-d)", "batch": 1024, "mean": OPENAI_CLIP_MEAN, "std": OPENAI_CLIP_STD},
    "clip_vit_l_14_336": {"label": "CLIP ViT-L/14 at 336 px (577 tokens)", "batch": 256, "mean": OPENAI_CLIP_MEAN, "std": OPENAI_CLIP_STD},
    "plip": {"label": "PLIP (HF CLIP ViT-B/32, get_image_features)", "batch": 4096, "mean": OPENAI_CLIP_MEAN, "std": OPENAI_CLIP_STD},
    "biomedclip": {"label": "BiomedCLIP image tower (timm ViT-B/16 + linear projection, 512-d)", "batch": 2048,
                   "mean": OPENAI_CLIP_MEAN, "std": OPENAI_CLIP_STD},
    "virchow_v1": {"label": "Virchow (ViT-H/14, 80-wide heads and a 3416-wide SwiGLU stored padded, class | mean patch token)", "batch": 512},
    "virchow_v2": {"label": "Virchow2 (as Virchow + 4 register tokens)", "batch": 512},
    # DINOv3 (dinov3.py): rotary embedding on q / k in place after the qkv GEMM
    "dinov3_vits16": {"label": "DINOv3 ViT-S/16 (201 tokens, RoPE)", "batch": 4096},
    "dinov3_vitb16": {"label": "DINOv3 ViT-B/16 (201 tokens, RoPE)", "batch": 2048},
    "dinov3_vitl16": {"label": "DINOv3 ViT-L/16 (201 tokens, RoPE)", "batch": 2048},
    "dinov3_vith16_plus": {"label": "DINOv3 ViT-H+/16 (201 tokens, RoPE, gated MLP)", "batch": 1024},
    # the remaining registered names: same architectures as a line above (other weights / normalisation)
    "vit_l_16": {"label": "ViT-L/16 (torchvision; the architecture of uni_v1 without LayerScale)", "batch": 2048},
    "h_optimus_1": {"label": "H-optimus-1 (as H-optimus-0)", "batch": 512,
                    "mean": (0.707223, 0.578729, 0.703617), "std": (0.211883, 0.230117, 0.177517)},
    "h0_mini": {"label": "H0-mini (ViT-B/14, 4 register tokens, class token + mean patch token)", "batch": 2048,
                "mean": (0.707223, 0.578729, 0.703617), "std": (0.211883, 0.230117, 0.177517)},
    "quilt_b_32": {"label": "QuiltNet-B-32 (open_clip ViT-B/32 image tower)", "batch": 4096, "mean": OPENAI_CLIP_MEAN, "std": OPENAI_CLIP_STD},
    "quilt_b_16": {"label": "QuiltNet-B-16 (open_clip ViT-B/16 image tower)", "batch": 2048, "mean": OPENAI_CLIP_MEAN, "std": OPENAI_CLIP_STD},
    "dinov3_vits16_plus": {"label": "DINOv3 ViT-S+/16 (201 tokens, RoPE, gated MLP)", "batch": 4096},
    "dinov3_vitl16_sat": {"label": "DINOv3 ViT-L/16, satellite weights (as dinov3_vitl16)", "batch": 2048},
    "dinov3_vit7b16_sat": {"label": "DINOv3 ViT-7B/16, satellite weights (as dinov3_vit7b16; run with --no-cpu-baseline)", "batch": 256},
    "dinov3_vit7b16": {"label": "DINOv3 ViT-7B/16 (201 tokens, dim 4096, 40 blocks, RoPE, gated MLP 8192; 6.7 G parameters: run "
                                "with --no-cpu-baseline, the fp32 oracle of this size is not a bounded sample)", "batch": 256},
}


def encoder_geometry(arch):
    """Tokens and algorithmic FLOP per tile (2*M*N*K per linear, 4*T^2*D per attention): `model` = every
    block for every token, `executed` = what this build runs (CLS-pooled encoders: the last block computes K / V for every
    token and everything after that for the CLS row only -- DESIGN.md section 3 -- identical features)."""
    P_ = (arch["image_size"] // arch["patch_size"]) ** 2
    T = 1 + int(arch.get("reg_tokens", 0)) + P_
    D, mlp, L = arch["dim"], arch["mlp_dim"], arch["depth"]
    mlp_w = 3 * D * mlp if arch.get("mlp") == "swiglu" else 2 * D * mlp          # SwiGLU packed: fc1 is [2 mlp, D]
    patch_embed = 2.0 * P_ * D * 3 * arch["patch_size"] ** 2
    block = 2.0 * T * (4 * D * D + mlp_w) + 4.0 * T * T * D                      # algorithmic: true head width
    model = patch_embed + L * block
    executed = model
    if arch.get("pool") == "attn":
        P = arch["pool_dim"]
        model += 2.0 * T * D * 2 * P + 4.0 * T * P + 2.0 * P * P
        executed = model
    elif arch.get("pool") == "cls_mean":
        executed = model                 # class token | mean patch token: every block runs for every token
    else:
        executed = model - block + 2.0 * T * D * 2 * D + 2.0 * (2 * D * D + mlp_w) + 4.0 * T * D
    return {"tokens": T, "model": model, "executed": executed}


MFMA_PEAK = {"f16": 2.5e15, "bf16": 2.5e15, "f32": 157.3e12}   # dense, MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=None,
                    help="tiles per step (device batch; default 2048, conch_v1 256 = its registered maximum)")
    ap.add_argument("--precision", default="float16", choices=["float16", "bfloat16", "float32"])
    ap.add_argument("--slide", type=int, default=None,
                    help="synthetic slide side in pixels (default: 100000 at every N = the north star's slide, BASELINE configs 3 / 4; "
                         "40000 = config 2's slide)")
    ap.add_argument("--encoder", default="vit_b_16", choices=sorted(ENCODERS),
                    help="registered encoder of the forward (vit_b_16 = configs 2 / 4, uni_v1 = config 3, conch_v1 = config 5; "
                         "vit_b_32 / vit_l_32 / vit_h_14 / uni_v2 = the rest of those encoder files; dinov2_* / phikon_* = the "
                         "transformers-backed files)")
    ap.add_argument("--slide-seed", type=int, default=1234, help="rank r embeds the synthetic slide of seed SLIDE_SEED + r")
    ap.add_argument("--dump-features", default=None,
                    help="rank 0 saves the float32 feature matrix of the timed steps (the gathered [N*K*B, D] matrix for N > 1) "
                         "as .npy -- used by the N > 1 rehearsal test")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary rates (ring / end-to-end / boundary / f32 / uni_v1 / conch_v1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0,
                    help="patches timed on the CPU oracle (default: 192 vit_b_16, 64 uni_v1, 32 conch_v1 = 10-20 s)")
    return ap.parse_args()


def slide_tiles(device, side, seed, count):
    """Device coords path + on-device tile synthesis -> uint8 [count, 256, 256, 3] in HBM."""
    import ctypes as C
    from atlaspatch_amd import _lib
    from atlaspatch_amd.core.wsi.synth_pixels import SynthSpec, analytic_mask
    from atlaspatch_amd.services.extraction import coords_from_mask

    spec = SynthSpec(width=side, height=side, seed=seed)
    mask = analytic_mask(spec)
    kw = dict(level0_wh=(side, side), downsamples=list(spec.downsamples), src_mag=spec.mag, tgt_mag=spec.mag,
              patch_size=256, step_size=None, tissue_thresh=0.0)
    t0 = time.perf_counter()
    coords, geom = coords_from_mask(mask, **kw)             # cold: first use of the kernels + allocations in this process
    cold_s = time.perf_counter() - t0
    warm = []
    for _ in range(10):                                     # warm, repeated: what every later slide of a run pays
        t0 = time.perf_counter()
        again, _ = coords_from_mask(mask, **kw)
        warm.append(time.perf_counter() - t0)
    assert np.array_equal(again, coords)
    coords_stats = {"cold_seconds": round(cold_s, 5), "warm_seconds_median": round(float(np.median(warm)), 5),
                    "warm_seconds_min": round(min(warm), 5), "warm_repeats": len(warm)}
    cells = int(math.ceil(side / 256) ** 2)
    n_slide = coords.shape[0]
    reps = int(math.ceil(count / max(1, n_slide)))
    xy = np.tile(coords[:, :2], (reps, 1))[:count].astype(np.int32)
    lib = _lib.load()
    d_xy = torch.from_numpy(np.ascontiguousarray(xy)).to(device)
    d_ell = torch.from_numpy(spec.ellipses()).to(device)
    tiles = torch.empty((count, 256, 256, 3), dtype=torch.uint8, device=device)
    _lib.check(lib.ap_synth_tiles(d_xy.data_ptr(), count, 256, 1, 0, side, side, spec.seed, d_ell.data_ptr(),
                                  d_ell.shape[0], tiles.data_ptr(), _lib.current_stream_ptr(device)))
    torch.cuda.synchronize(device)
    return tiles, n_slide, cells, coords_stats, coords


def cpu_oracle_forward(encoder, seed):
    """The CPU oracle of one registered encoder on the same seeded weights the device path gets (torch fp32 restatement of
    the reference path, oracle/vit_oracle.py): returns f(list of uint8 tiles) -> float32 [n, D]."""
    from PIL import Image
    from atlaspatch_amd.encoders.vit import ARCHS, TRANSFORM_RESIZE, random_attn_pool, random_canonical_state_dict
    from oracle import vit_oracle

    arch = ARCHS[encoder]
    sd = random_canonical_state_dict(arch, seed=seed)
    if encoder == "vit_b_16":             # the HF-keyed forward the reference's own extract_batch outputs pin (golden G1)
        hf = {"embeddings.patch_embeddings.projection.weight": sd["patch_embed.weight"],
              "embeddings.patch_embeddings.projection.bias": sd["patch_embed.bias"],
              "embeddings.cls_token": sd["cls_token"].view(1, 1, -1),
              "embeddings.position_embeddings": sd["pos_embed"][None],
              "layernorm.weight": sd["norm.weight"], "layernorm.bias": sd["norm.bias"]}
        for i in range(arch["depth"]):
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
        return lambda patches: vit_oracle.extract_batch(hf, patches, heads=arch["heads"], batch_size=32)
    if encoder == "conch_v1":
        pool = random_attn_pool(arch, seed)
        trunk = {k: v for k, v in sd.items() if not k.startswith("attn_pool.")}
        return lambda patches: np.concatenate([vit_oracle.conch_encode_image(
            trunk, pool, patches[s:s + 8], heads=arch["heads"], depth=arch["depth"], pool_heads=arch["pool_heads"])
            for s in range(0, len(patches), 8)], 0)
    size, filt = TRANSFORM_RESIZE[encoder]      # timm / torchvision: Resize(size) with Pillow on the PIL tile, then centre crop

    def forward(patches):
        outs = []
        for s in range(0, len(patches), 32):
            arrs = []
            for pch in patches[s:s + 32]:
                img = Image.fromarray(np.asarray(pch))
                w, h = img.size
                if min(w, h) != size:
                    nw, nh = (size, int(size * h / w)) if w <= h else (int(size * w / h), size)
                    img = img.resize((nw, nh), Image.Resampling.BICUBIC if filt == "bicubic" else Image.Resampling.BILINEAR)
                arrs.append(np.asarray(img))
            x = vit_oracle.preprocess_center_crop(np.stack(arrs, 0), crop=arch["image_size"])
            outs.append(vit_oracle.vit_tokens_canonical(sd, x, heads=arch["heads"], depth=arch["depth"])[:, 0].numpy())
        return np.concatenate(outs, 0)
    return forward


def cpu_baseline(encoder, sample, seed):
    """CPU oracle (torch fp32 restatement of the reference path) on a bounded sample."""
    forward = cpu_oracle_forward(encoder, seed)
    rng = np.random.default_rng(0)
    patches = [rng.integers(0, 256, (256, 256, 3), dtype=np.uint8) for _ in range(sample)]
    # torch's default (all physical cores) is far from the best setting for this batch-32 fp32 forward on a 2-socket
    # host (measured on the MI355X box, 2 x EPYC 9575F: 8 thr 13.9, 16 thr 19.1, 32 thr 17.3, 64 thr 15.5, 128 thr
    # 5.5 patches/s): probe a few counts on one batch and time the sample at the fastest, so the baseline is the
    # CPU path at its best, not at its default.
    default_threads = torch.get_num_threads()
    best, best_rate = default_threads, 0.0
    probe = min(32, sample) if encoder == "vit_b_16" else min(8, sample)
    for th in sorted({t for t in (8, 16, 32, 64) if t <= (os.cpu_count() or 1)} | {min(default_threads, 64)}):
        torch.set_num_threads(th)
        if encoder == "vit_b_16":
            forward(patches[:probe])                 # warm-up at this count
        t0 = time.perf_counter()
        forward(patches[:probe])
        r = probe / (time.perf_counter() - t0)
        if r > best_rate:
            best, best_rate = th, r
    torch.set_num_threads(best)
    t0 = time.perf_counter()
    out = forward(patches)
    dt = time.perf_counter() - t0
    torch.set_num_threads(default_threads)
    return out, patches, sample / dt, best


def cpu_coords_baseline(side, seed):
    """The coordinate path's CPU restatement (oracle/coords_oracle.py) on the bench slide's mask: cells/s."""
    from atlaspatch_amd.core.wsi.synth_pixels import SynthSpec, analytic_mask
    from oracle import coords_oracle
    spec = SynthSpec(width=side, height=side, seed=seed)
    mask = analytic_mask(spec)
    t0 = time.perf_counter()
    coords, _ = coords_oracle.coords_from_mask(mask, level0_wh=(side, side), downsamples=list(spec.downsamples),
                                               src_mag=spec.mag, tgt_mag=spec.mag, patch_size=256, step_size=None,
                                               tissue_thresh=0.0)
    return coords, time.perf_counter() - t0


def _timed_forward(ex, tiles, steps, warm=1):
    out = torch.empty((tiles.shape[0], ex.embedding_dim), dtype=torch.float32, device=tiles.device)
    for _ in range(warm):
        ex.forward_device(tiles, out)
    torch.cuda.synchronize(tiles.device)
    t0 = time.perf_counter()
    for _ in range(steps):
        ex.forward_device(tiles, out)
    torch.cuda.synchronize(tiles.device)
    return steps * tiles.shape[0] / (time.perf_counter() - t0)


def jpeg_store_rate(device, tmp):
    """Build a JPEG tile store for the 100 000^2 slide (untimed: pixels from the device generator, Pillow encoders on a
    thread pool), then time `process` on it end to end."""
    import concurrent.futures as futures
    from click.testing import CliRunner
    from PIL import Image
    from atlaspatch_amd import _lib
    from atlaspatch_amd.cli import cli
    from atlaspatch_amd.core.wsi.synth_pixels import SynthSpec, analytic_mask
    from atlaspatch_amd.services.extraction import coords_from_mask
    from atlaspatch_amd.utils.h5 import h5
    side = 100000
    spec = SynthSpec(width=side, height=side, seed=1234)
    coords, _ = coords_from_mask(analytic_mask(spec), level0_wh=(side, side), downsamples=list(spec.downsamples),
                                 src_mag=spec.mag, tgt_mag=spec.mag, patch_size=256, step_size=None, tissue_thresh=0.0)
    store = os.path.join(tmp, "jpeg_tiles")
    os.makedirs(store)
    lib = _lib.load()
    ell = torch.from_numpy(spec.ellipses()).to(device)
    workers = min(64, os.cpu_count() or 8)
    t0 = time.perf_counter()
    with futures.ThreadPoolExecutor(workers) as pool:
        for lo in range(0, len(coords), 4096):
            xy = torch.from_numpy(np.ascontiguousarray(coords[lo:lo + 4096, :2], dtype=np.int32)).to(device)
            dev_tiles = torch.empty((xy.shape[0], 256, 256, 3), dtype=torch.uint8, device=device)
            _lib.check(lib.ap_synth_tiles(xy.data_ptr(), xy.shape[0], 256, 1, 0, side, side, spec.seed, ell.data_ptr(),
                                          ell.shape[0], dev_tiles.data_ptr(), _lib.current_stream_ptr(device)))
            host = dev_tiles.cpu().numpy()
            list(pool.map(lambda i: Image.fromarray(host[i]).save(
                os.path.join(store, f"{coords[lo + i, 0]}_{coords[lo + i, 1]}_256.jpg"), quality=80), range(host.shape[0])))
    build_s = time.perf_counter() - t0
    size_mb = sum(os.path.getsize(os.path.join(store, f)) for f in os.listdir(store)) / 1e6
    slide = os.path.join(tmp, "bigjpeg.synth")
    json.dump({"width": side, "height": side, "seed": 1234, "mag": 20, "mpp": 0.5, "downsamples": [1, 4, 16],
               "jpeg_tiles": "jpeg_tiles"}, open(slide, "w"))
    old = {k: os.environ.get(k) for k in ("ATLASPATCH_WEIGHTS_DIR", "ATLASPATCH_RANDOM_INIT")}
    os.environ["ATLASPATCH_WEIGHTS_DIR"] = tmp
    os.environ.pop("ATLASPATCH_RANDOM_INIT", None)
    try:
        out_dir = os.path.join(tmp, "out_jpeg")
        t0 = time.perf_counter()
        res = CliRunner().invoke(cli, ["process", slide, "-o", out_dir, "--patch-size", "256", "--target-mag", "20",
                                       "--feature-extractors", "vit_b_16", "--feature-precision", "float16",
                                       "--feature-num-workers", str(workers)], catch_exceptions=False)
        dt = time.perf_counter() - t0
        assert res.exit_code == 0 and "failures: 0" in res.output, res.output
        with h5.File(os.path.join(out_dir, "patches", "bigjpeg.h5"), "r") as f:
            n = int(f["coords"].shape[0])
            assert f["features"]["vit_b_16"].shape == (n, 768)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return {"patches_per_s": round(n / dt, 1), "tiles": n, "seconds": round(dt, 3), "host_threads": workers,
            "store_MB": round(size_mb, 1), "store_build_seconds_untimed": round(build_s, 1),
            "what": "`process` on the 100000x100000 slide with its 58938 tiles stored as JPEG files (q80, 4:2:0): file read + "
                    "libjpeg-turbo decode on pinned host threads through the native batched hook (outside the interpreter "
                    "lock) -> pinned ring -> H2D -> K1 + ViT-B/16 f16 -> H5"}


def openslide_stub_rate(device, ex, tmp):
    """Tiles of a 100 000^2 slide served by a stand-in libopenslide (tools/stub_openslide: libopenslide's interface, hash
    pixels with partial alpha; the real library is not in the image) through OpenSlideWSI.read_tiles_into ->
    ap_host_openslide_read_tiles on the ring's pinned threads -> pinned ring -> H2D -> K1 + ViT-B/16 -> pinned features.
    The coords are the synthetic 100 000^2 slide's 58 938 rows (segmentation is the same for every backend and is timed
    elsewhere).  Runs in a subprocess: libopenslide is resolved once per process."""
    import subprocess
    import textwrap
    from tools import stub_openslide as so
    lib = so.build(os.path.join(tmp, "stublib"))
    code = f"""
        import json, os, sys, time, numpy as np, torch
        sys.path.insert(0, {ROOT!r})
        from tools import stub_openslide as so
        from atlaspatch_amd.core.wsi import openslide_wsi
        from atlaspatch_amd.core.wsi.synth_pixels import SynthSpec, analytic_mask
        from atlaspatch_amd.encoders.vit import build_hip_vit_extractor
        from atlaspatch_amd.services.extraction import coords_from_mask
        from atlaspatch_amd.services.tile_ring import TileRing
        openslide_wsi.openslide = so.python_module({lib!r})
        path = so.write_slide({os.path.join(tmp, 'stub100k.svs')!r}, 100000, 100000, seed=9, alpha_period=3)
        spec = SynthSpec(width=100000, height=100000, seed=1234)
        coords, _ = coords_from_mask(analytic_mask(spec), level0_wh=(100000, 100000), downsamples=[1.0, 4.0, 16.0], src_mag=20,
                                     tgt_mag=20, patch_size=256, step_size=None, tissue_thresh=0.0)
        dev = torch.device("cuda:0")
        ex = build_hip_vit_extractor(name="vit_b_16", arch="vit_b_16", device=dev, dtype=torch.float16, random_init_seed=0, max_batch=2048)
        wsi = openslide_wsi.OpenSlideWSI(path)
        wsi._ensure_loaded()
        workers = min(64, os.cpu_count() or 8)
        ring = TileRing(device=dev, batch=2048, patch_size=256, slots=3, workers=workers)
        read = lambda x, y, rw, rh, lv: wsi.extract((x, y), lv=lv, wh=(rw, rh), mode="array")
        fwd = lambda t, o: ex.forward_device(t, o)
        ring.run(coords[:4096], read, fwd, 768, read_chunk=wsi.read_tiles_into)
        t0 = time.perf_counter()
        feats = ring.run(coords, read, fwd, 768, read_chunk=wsi.read_tiles_into)
        dt = time.perf_counter() - t0
        # decode-only rate of the hook (no GPU work): the ring's threads, one slot
        import concurrent.futures as futures
        n = 8192
        host = np.empty((n, 256, 256, 3), np.uint8)
        rows = coords[:n].tolist()
        chunk = 32
        t1 = time.perf_counter()
        with futures.ThreadPoolExecutor(workers) as pool:
            list(pool.map(lambda s: wsi.read_tiles_into(rows[s:s + chunk], host.ctypes.data + s * 196608, 256), range(0, n, chunk)))
        dt_read = time.perf_counter() - t1
        assert np.isfinite(feats).all() and feats.shape == (coords.shape[0], 768)
        print(json.dumps(dict(patches_per_s=round(coords.shape[0] / dt, 1), tiles=int(coords.shape[0]), seconds=round(dt, 3),
                              host_threads=workers, hook_read_only_tiles_per_s=round(n / dt_read, 1))))
        """
    env = dict(os.environ, ATLASPATCH_LIBOPENSLIDE=lib)
    res = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    if res.returncode != 0:
        return {"error": (res.stdout + res.stderr)[-600:]}
    out = json.loads(res.stdout.strip().splitlines()[-1])
    out["what"] = ("58938 tiles of a 100000x100000 slide read by a stand-in libopenslide through the native batched hook "
                   "(ap_host_openslide_read_tiles: premultiplied ARGB -> RGB as openslide-python + PIL do, outside the interpreter "
                   "lock) -> pinned ring -> H2D -> K1 + ViT-B/16 f16 -> features; hook_read_only = the hook alone on the same threads")
    return out


def secondary_rates(device, ex, tiles, B):
    """Rates that are NOT `value` (outside its timed region), measured in the same process so that the driver's line
    carries them: PCIe-inclusive ring, end-to-end CLI on the 100 000^2 slide (device tile source / host ring with the
    native renderer as the decoder), the reference's call shape extract_batch(32 host patches), float32 mode with its
    fc1 roofline fraction, and the other two BASELINE encoders kernel-only (transform included)."""
    import tempfile
    from click.testing import CliRunner
    from atlaspatch_amd.cli import cli
    from atlaspatch_amd.encoders import build_default_registry
    from atlaspatch_amd.encoders.vit import build_hip_vit_extractor
    from atlaspatch_amd.services.tile_ring import TileRing
    from atlaspatch_amd.utils.h5 import h5
    rates = {}
    # ---- (1) PCIe-inclusive: tiles in host memory -> pinned ring -> HBM -> forward -> features back on the host
    n_host = min(8 * B, tiles.shape[0])           # 8 batches: the pipeline's fill (one gather + one H2D before the first forward) is 1/8 of the run
    host = tiles[:n_host].cpu().numpy()
    coords = np.stack([np.arange(n_host), np.zeros(n_host), np.full(n_host, 256), np.full(n_host, 256),
                       np.zeros(n_host)], 1).astype(np.int32)
    workers = min(32, os.cpu_count() or 8)
    ring = TileRing(device=device, batch=B, patch_size=256, slots=3, workers=workers)
    fwd = lambda t, o: ex.forward_device(t, o)
    read = lambda x, y, rw, rh, lv: host[x]
    ring.run(coords[:B], read, fwd, ex.embedding_dim)
    t0 = time.perf_counter()
    ring.run(coords, read, fwd, ex.embedding_dim)
    dt = time.perf_counter() - t0
    ring.close()
    rates["ring_pcie_inclusive"] = {"patches_per_s": round(n_host / dt, 1), "tiles": n_host, "host_threads": workers,
                                    "what": "uint8 tiles in host memory -> pinned 3-slot ring -> H2D on a copy stream -> "
                                            "K1 + ViT-B/16 -> float32 features back in pinned host memory"}
    # ---- (2) the reference's call shape: extract_batch on 32 host patches (models/patch/base.py:76-107)
    patches = [host[i] for i in range(32)]
    ex.extract_batch(patches, batch_size=32)
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        ex.extract_batch(patches, batch_size=32)
    dt = (time.perf_counter() - t0) / reps
    rates["extract_batch_32"] = {"patches_per_s": round(32 / dt, 1), "ms_per_call": round(dt * 1e3, 3),
                                 "what": "FeatureExtractor.extract_batch(32 host uint8 patches) -> float32 [32, 768] on the "
                                         "host: synchronous H2D + forward + D2H, the drop-in boundary as the reference "
                                         "calls it (storage.py:283-294)"}
    del host
    # ---- (3) end-to-end `process` CLI, f16, weights from a file: config 2's own slide (40 000^2), the north-star slide
    #      (100 000^2; device tile source / host ring), eight 100 000^2 slides in one invocation, with the stage breakdown
    from atlaspatch_amd.utils import stages as stage_table
    with tempfile.TemporaryDirectory() as tmp:
        from safetensors.torch import save_file
        from atlaspatch_amd.encoders.vit import ARCHS, random_canonical_state_dict
        save_file(random_canonical_state_dict(ARCHS["vit_b_16"], 0), os.path.join(tmp, "vit_b_16.safetensors"))

        def synth(name, side, seed=1234):
            path = os.path.join(tmp, name)
            os.makedirs(os.path.dirname(path), exist_ok=True)
            json.dump({"width": side, "height": side, "seed": seed, "mag": 20, "mpp": 0.5, "downsamples": [1, 4, 16]}, open(path, "w"))
            return path

        def run_cli(command, target, out_name, env, extra=()):
            env = dict(env, ATLASPATCH_WEIGHTS_DIR=tmp)
            old = {k: os.environ.get(k) for k in list(env) + ["ATLASPATCH_RANDOM_INIT"]}
            os.environ.update(env)
            if "ATLASPATCH_RANDOM_INIT" not in env:
                os.environ.pop("ATLASPATCH_RANDOM_INIT", None)
            try:
                out_dir = os.path.join(tmp, out_name)
                stage_table.snapshot(reset=True)
                t0 = time.perf_counter()
                res = CliRunner().invoke(cli, [command, target, "-o", out_dir, "--patch-size", "256", "--target-mag", "20", *extra],
                                         catch_exceptions=False)
                dt = time.perf_counter() - t0
                assert res.exit_code == 0 and "failures: 0" in res.output, res.output
                return out_dir, dt, {k: v["seconds"] for k, v in stage_table.snapshot(reset=True).items()}
            finally:
                for k, v in old.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v

        feat = ("--feature-extractors", "vit_b_16", "--feature-precision", "float16",
                "--feature-num-workers", str(min(64, os.cpu_count() or 8)))

        def tiles_in(out_dir):
            total = 0
            for name in sorted(os.listdir(os.path.join(out_dir, "patches"))):
                if name.endswith(".h5"):
                    with h5.File(os.path.join(out_dir, "patches", name), "r") as f:
                        n = int(f["coords"].shape[0])
                        assert f["features"]["vit_b_16"].shape == (n, 768)
                        total += n
            return total

        what = ("`process`: checkpoint load (side thread, overlaps phase 1), analytic segmentation, device coords, H5 coords, {}"
                ", K1 + ViT-B/16 f16, H5 features (writer thread); stage seconds are sums per stage, overlapping stages included")
        slide40, slide100 = synth("s40.synth", 40000), synth("big.synth", 100000)
        run_cli("process", slide40, "warm", {}, feat)                    # first invocation of the process: kernels, allocator
        for key, target, env, src in (("e2e_cli_40k", slide40, {}, "tiles from the slide's device tile source"),
                                      ("e2e_cli_100k_device_tile_source", slide100, {}, "tiles from the slide's device tile source"),
                                      ("e2e_cli_100k_host_ring", slide100, {"ATLASPATCH_HOST_TILES": "1"},
                                       "tiles rendered on HOST threads by the native renderer (the stand-in decoder) -> pinned ring -> H2D")):
            out_dir, dt, st = run_cli("process", target, "out_" + key, env, feat)
            n = tiles_in(out_dir)
            rates[key] = {"patches_per_s": round(n / dt, 1), "tiles": n, "seconds": round(dt, 3), "stage_seconds": st,
                          "what": what.format(src)}
        # ---- BASELINE config 3 end to end: the 100 000^2 slide, UNI (ViT-L/16 + LayerScale, 256 -> 224 bicubic on the device),
        #      tiles rendered by host threads through the async tile ring
        save_file(random_canonical_state_dict(ARCHS["uni_v1"], 0), os.path.join(tmp, "uni_v1.safetensors"))
        feat_uni = ("--feature-extractors", "uni_v1", "--feature-precision", "float16",
                    "--feature-num-workers", str(min(64, os.cpu_count() or 8)))
        out_dir, dt, st = run_cli("process", slide100, "out_uni", {"ATLASPATCH_HOST_TILES": "1"}, feat_uni)
        with h5.File(os.path.join(out_dir, "patches", sorted(os.listdir(os.path.join(out_dir, "patches")))[0]), "r") as f:
            n = int(f["coords"].shape[0])
            assert f["features"]["uni_v1"].shape == (n, 1024)
        rates["e2e_cli_100k_uni_v1_host_ring"] = {
            "patches_per_s": round(n / dt, 1), "tiles": n, "seconds": round(dt, 3), "stage_seconds": st,
            "what": "BASELINE config 3: `process` on the 100000x100000 slide with uni_v1 (f16), tiles rendered on host threads -> "
                    "pinned ring -> H2D -> device Resize(224, bicubic) + K1 + ViT-L/16 -> H5"}
        os.remove(os.path.join(tmp, "uni_v1.safetensors"))
        for i in range(8):
            synth(f"eight/s{i}.synth", 100000, seed=300 + i)
        out_dir, dt, st = run_cli("process", os.path.join(tmp, "eight"), "out_eight", {}, feat)
        n = tiles_in(out_dir)
        rates["e2e_cli_8x100k_one_gpu"] = {"patches_per_s": round(n / dt, 1), "tiles": n, "slides": 8, "seconds": round(dt, 3),
                                           "stage_seconds": st, "what": "one `process` invocation over a folder of eight 100000x100000 "
                                           "slides on one GPU: slide k's H5 write overlaps slide k + 1's embedding"}
        # ---- (3a) segment-and-get-coords with the SAM2 segmenter forced on synthetic slides (seeded random SAM2 weights: the
        #      masks are arbitrary, the work is not): thumbnail -> SAM2 -> contours -> grid -> H5, 64 slides of 100 000^2
        from atlaspatch_amd.services.segmentation import random_sam2_state_dict
        torch.save({"model": random_sam2_state_dict(0)}, os.path.join(tmp, "sam2.pt"))
        for i in range(64):
            synth(f"seg/s{i:03d}.synth", 100000, seed=500 + i)
        seg_env = {"ATLASPATCH_SEGMENTER": "sam2"}
        run_cli("segment-and-get-coords", os.path.join(tmp, "eight"), "seg_warm", seg_env)
        out_dir, dt, st = run_cli("segment-and-get-coords", os.path.join(tmp, "seg"), "out_seg", seg_env)
        steady = max(1e-9, st.get("segmentation", 0.0))
        rates["segment_and_get_coords"] = {
            "slides_per_s": round(64 / dt, 2), "slides": 64, "seconds": round(dt, 3), "ms_per_slide": round(dt / 64 * 1e3, 2),
            "stage_seconds": st, "reference_published": "~19 s per 100 WSIs (docs/release-notes/v1.0.0.md:17; hardware unstated)",
            "what": "one `segment-and-get-coords` invocation over 64 synthetic 100000x100000 slides, SAM2 checkpoint loaded from a "
                    "file: level read + cv2 / Pillow thumbnail + SAM2 (hipGraph) + mask resize on the device, contours + grid + "
                    "H5 on worker threads; includes the per-invocation predictor build and graph capture"}
        # ---- (3b) the same slide with its tiles stored as JPEG files (quality 80, 4:2:0), decoded by the ring's pinned host
        #      threads through the native batched decoder (ap_host_decode_jpeg_tiles, libjpeg-turbo outside the interpreter lock)
        rates["e2e_cli_100k_jpeg_store"] = jpeg_store_rate(device, tmp)
        # ---- (3c) tiles served by (stub) OpenSlide through the native batched hook
        rates["e2e_openslide_stub_100k"] = openslide_stub_rate(device, ex, tmp)
    # ---- (4) float32 mode (`--feature-precision float32`; the mode that meets 1e-3 on EVERY element): split-f16 products (the
    #      default: float32 buffers / LayerNorm / softmax / stream, GEMM products as three f16 MFMA passes on hi / lo halves) and the
    #      exact f32 MFMA chain (option split_f16 off).  fc1 priced against the peak of the instruction it runs on: the split form
    #      executes 3 x 2 M N K f16 flop per launch (dense f16 peak), the exact form 2 M N K on v_mfma_f32_32x32x2_f32
    Bf = 1024
    ex32 = build_hip_vit_extractor(name="vit_b_16", arch="vit_b_16", device=device, dtype=torch.float32,
                                   random_init_seed=0, max_batch=Bf)
    out32 = torch.empty((Bf, 768), dtype=torch.float32, device=device)

    def f32_rate(split):
        ex32.vit.set_option("split_f16", split)
        ex32.forward_device(tiles[:Bf], out32)
        ex32.vit.profile(True)
        v = _timed_forward(ex32, tiles[:Bf], 4, warm=0)
        prof = ex32.vit.profile_read()
        ex32.vit.profile(False)
        ms, cnt = prof["gemm_fc1"]
        eff = 2.0 * Bf * 197 * 3072 * 768 / ((ms / max(1, cnt)) * 1e-3) / 1e12 if cnt else 0.0
        peak = (MFMA_PEAK["f16"] if split else MFMA_PEAK["f32"]) / 1e12
        executed = eff * (3 if split else 1)
        return {"patches_per_s": round(v, 1), "device_batch": Bf, "fc1_ms_per_launch": round(ms / max(1, cnt), 4),
                "fc1_TFLOPs": round(eff, 1), "fc1_executed_TFLOPs": round(executed, 1), "fc1_peak_TFLOPs": peak,
                "fc1_frac": round(executed / peak, 4),
                "ms_by_kind": {k: round(t / 4, 3) for k, (t, _) in prof.items()}}

    rates["float32_mode"] = dict(f32_rate(True), what="ViT-B/16 float32, split-f16 products (AP_VIT_OPT_SPLIT_F16, the default of the "
                                 "float32 compute type): x = hi + 2^-11 lo, w a ~= w_hi a_hi + 2^-11 (w_hi a_lo + w_lo a_hi), three "
                                 "v_mfma_f32_32x32x16_f16 passes, f32 accumulation; tiles resident in HBM")
    rates["float32_exact_mode"] = dict(f32_rate(False), what="the same with the exact f32 MFMA chain (v_mfma_f32_32x32x2_f32)")
    ex32.vit.set_option("split_f16", True)
    ex32.cleanup()
    # ---- (4b) SAM2 Hiera-T tissue segmentation of one 1024 x 1024 thumbnail (config 1's hot path; random weights: the
    #      checkpoint is not available offline), float32 MFMA operator set replayed as a hipGraph
    from atlaspatch_amd.services.sam2_hip import Sam2HipPredictor
    from atlaspatch_amd.services.segmentation import random_sam2_state_dict
    pred = Sam2HipPredictor(random_sam2_state_dict(0), device=str(device))
    img = np.random.default_rng(0).integers(0, 256, (1024, 1024, 3), dtype=np.uint8)
    sam2_flop = pred.count_flop()
    pred.predict_image(img)                                   # captures the graph
    torch.cuda.synchronize(device)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(10):
        pred._graph.replay()
    ev1.record()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(5):
        pred.predict_image(img)
    host_ms = (time.perf_counter() - t0) / 5 * 1e3
    # --seg-batch-size 4: four thumbnails per forward (trunk on the stacked batch; every mask bit-equal to its single forward)
    imgs4 = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (4, 1024, 1024, 3), dtype=np.uint8)).to(device)
    with torch.inference_mode():
        for _ in range(2):
            pred._graph_masks_device(imgs4)
        torch.cuda.synchronize(device)
        eb0, eb1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        eb0.record()
        for _ in range(5):
            pred._graph_masks_device(imgs4)
        eb1.record()
        torch.cuda.synchronize(device)
    rates["sam2_segmentation"] = {"ms_per_slide_device": round(ev0.elapsed_time(ev1) / 10, 3), "ms_per_slide_host_to_host": round(host_ms, 3),
                                  "ms_per_slide_device_seg_batch_4": round(eb0.elapsed_time(eb1) / 20, 3),
                                  "slides_per_s": round(1e3 / host_ms, 1),
                                  "gflop_per_slide": round(sam2_flop / 1e9, 1),
                                  "TFLOPs": round(sam2_flop / (ev0.elapsed_time(ev1) / 10 * 1e-3) / 1e12, 1),
                                  "TFLOPs_seg_batch_4": round(sam2_flop / (eb0.elapsed_time(eb1) / 20 * 1e-3) / 1e12, 1),
                                  "peak_TFLOPs": MFMA_PEAK["f32"] / 1e12,
                                  "frac": round(sam2_flop / (ev0.elapsed_time(ev1) / 10 * 1e-3) / MFMA_PEAK["f32"], 4),
                                  "frac_seg_batch_4": round(sam2_flop / (eb0.elapsed_time(eb1) / 20 * 1e-3) / MFMA_PEAK["f32"], 4),
                                  "executed_TFLOPs_f16_mfma": round(3.0 * sam2_flop / (ev0.elapsed_time(ev1) / 10 * 1e-3) / 1e12, 1),
                                  "executed_frac_of_f16_peak": round(3.0 * sam2_flop / (ev0.elapsed_time(ev1) / 10 * 1e-3) / MFMA_PEAK["f16"], 4),
                                  "what": "SAM2.1 Hiera-T image encoder + box-prompted mask decoder on one 1024x1024 thumbnail "
                                          "(services/segmentation.py:120-140), float32 buffers, row-wise layers and attention as split-f16 "
                                          "products (three f16 MFMA passes on hi / lo halves, f32 accumulation; ATLASPATCH_SAM2_EXACT_F32=1 = "
                                          "the exact f32 MFMA chain of rounds 2-5), 184 launches captured in one hipGraph.  `frac` prices the "
                                          "MODEL's multiply-adds against the exact-f32 MFMA peak (the ceiling of an implementation that keeps "
                                          "the reference's float32 arithmetic -- comparable with earlier rounds); `executed_frac_of_f16_peak` "
                                          "prices the 3x flop the split form executes against the dense f16 peak.  host-to-host includes the "
                                          "PIL resize to 1024x1024, H2D and the mask D2H"}
    pred.close()
    # ---- (5) the other two BASELINE encoders, kernel-only incl. their transform (device resize), f16
    os.environ["ATLASPATCH_RANDOM_INIT"] = "0"
    try:
        reg = build_default_registry(device=device, dtype=torch.float16)
        for name, nb in (("uni_v1", 2048), ("conch_v1", 256)):
            enc = reg.create(name)
            nb = min(nb, enc.max_batch, tiles.shape[0])
            rates[f"{name}_kernel_only"] = {"patches_per_s": round(_timed_forward(enc, tiles[:nb], 3), 1),
                                            "device_batch": nb, "dtype": "f16",
                                            "what": "registered encoder incl. its Resize on the device, tiles resident in HBM"}
            enc.cleanup()
    finally:
        os.environ.pop("ATLASPATCH_RANDOM_INIT", None)
    return rates


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this same command under torch.distributed.run
    (one process per GPU, RCCL).  On a box with fewer than N devices the ranks share cuda:0 and the collectives go through
    gloo -- a REHEARSAL of the N > 1 code path, marked as such in the line, never a measurement."""
    import socket
    import subprocess
    env = dict(os.environ)
    have = torch.cuda.device_count()
    if have < args.gpus:
        env["AP_BENCH_BACKEND"] = "gloo"
        env["AP_BENCH_ONE_GPU"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device: the product has no CPU fallback")
        respawn_under_torchrun(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.slide is None:
        args.slide = 100000                 # one workload string for N = 1, 2, 4, 8 (round 4 spliced 40 000 / 100 000)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product has no CPU fallback")
    # AP_BENCH_BACKEND=gloo + AP_BENCH_ONE_GPU=1: rehearsal of the N > 1 path on a one-GPU box (every rank on cuda:0,
    # collectives through gloo); the line it prints is marked and is not a measurement
    rehearsal = os.environ.get("AP_BENCH_BACKEND", "nccl") != "nccl"
    device = torch.device("cuda", 0 if os.environ.get("AP_BENCH_ONE_GPU") else local_rank)
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend=os.environ["AP_BENCH_BACKEND"])
        else:
            dist.init_process_group(backend="nccl", device_id=device)
        world = dist.get_world_size()             # the ranks the process group actually has
        rank = dist.get_rank()
    if world != args.gpus:
        # `value` is "the units all ranks processed / time": a line whose n_gpus disagrees with the ranks that ran would be read as
        # a scaling point it is not (launch with --nproc-per-node = --gpus, or let bench.py re-spawn itself: `python bench.py --gpus N`)
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {world} rank(s) (WORLD_SIZE={os.environ.get('WORLD_SIZE', 'unset')}); "
                         f"refusing to print a line for a configuration that did not run")

    from atlaspatch_amd.encoders.vit import ARCHS, TRANSFORM_RESIZE, build_hip_vit_extractor

    dtype = {"float16": torch.float16, "bfloat16": torch.bfloat16, "float32": torch.float32}[args.precision]
    short = {"float16": "f16", "bfloat16": "bf16", "float32": "f32"}[args.precision]
    enc = ENCODERS[args.encoder]
    arch = ARCHS[args.encoder]
    if args.batch is None:
        args.batch = enc["batch"]
    geo = encoder_geometry(arch)
    T, D, MLP = geo["tokens"], arch["dim"], arch["mlp_dim"] * (2 if arch.get("mlp") == "swiglu" else 1)   # fc1 rows
    ex = build_hip_vit_extractor(name=args.encoder, arch=args.encoder, device=device, dtype=dtype,
                                 random_init_seed=0, max_batch=args.batch, resize=TRANSFORM_RESIZE[args.encoder],
                                 expect_size=None, mean=enc.get("mean"), std=enc.get("std"))
    B, K, W = args.batch, args.steps, args.warmup
    pool_steps = min(K, 12)                       # distinct tile batches cycled through the timed steps
    tiles, n_slide, cells, coords_stats, dev_coords = slide_tiles(device, args.slide, args.slide_seed + rank, pool_steps * B)
    feats = torch.empty((K * B, ex.embedding_dim), dtype=torch.float32, device=device)

    def step(i, dst):
        s = (i % pool_steps) * B
        ex.forward_device(tiles[s:s + B], dst)    # the transform's Resize (uni_v1 / conch_v1) + K1 + encoder

    gathered = None
    gather_algo = None
    if dist is not None:
        from atlaspatch_amd.orchestration.dispatch import gather_algorithm, gather_feature_matrix
        gather_algo = gather_algorithm()          # ATLASPATCH_GATHER_ALGO: "allgather" (default) | "pairs" -- the product's two exchanges
        gathered = torch.empty((world * K * B, ex.embedding_dim), dtype=torch.float32, device=device)

    def reassemble():
        if gather_algo == "pairs":                # every rank's exact block to every peer, one point-to-point transfer per link
            return gather_feature_matrix(feats, algorithm="pairs")
        dist.all_gather_into_tensor(gathered, feats)
        return gathered

    for i in range(W):
        step(i, feats[:B])
    if dist is not None and W > 0:                # warm the collective's channels too (untimed, like the W steps)
        reassemble()
    torch.cuda.synchronize(device)
    ex.vit.profile(True)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    # what the chip ran at: shader-clock stamps on the stream just outside the timed region + package power polled by a host
    # thread while it runs (atlaspatch_amd/utils/telemetry.py).  Both stay OUTSIDE the barrier / synchronize brackets.
    from atlaspatch_amd.utils.telemetry import ClockProbe, PowerSampler
    clock_probe = ClockProbe(device)
    power = PowerSampler(device)
    clock_probe.start()
    power.__enter__()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(K):
        step(i, feats[i * B:(i + 1) * B])
    ev[1].record()
    assembled = None
    if dist is not None:        # reassemble the feature matrix on every rank (north star): one all-gather (or the all-pairs exchange)
        assembled = reassemble()
    ev[2].record()
    torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    power.__exit__(None, None, None)
    clock_probe.stop()
    torch.cuda.synchronize(device)
    clock_info, power_info = clock_probe.read(), power.summary()
    prof = ex.vit.profile_read()
    ex.vit.profile(False)
    per_rank = None
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = {"rank": rank, "device": str(device), "slide_seed": args.slide_seed + rank,
                "forward_ms": round(ev[0].elapsed_time(ev[1]), 3),
                "patches_per_s": round(K * B / (ev[0].elapsed_time(ev[1]) * 1e-3), 1),
                "all_gather_ms": round(ev[1].elapsed_time(ev[2]), 3)}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    if args.dump_features and rank == 0:
        whole = feats if dist is None else (torch.cat(assembled, 0) if isinstance(assembled, list) else assembled)
        np.save(args.dump_features, whole.cpu().numpy())

    # Secondary, untimed-for-`value` measurement: the same K steps with the last block computed for every token
    # (option full_last_block), so the line shows what the CLS-only tail is worth.
    full_value = None
    full_last = bool(os.environ.get("AP_VIT_FULL_LAST_BLOCK"))      # the default the encoder object was created with
    cls_tail_possible = geo["executed"] != geo["model"]
    if world == 1 and not full_last and cls_tail_possible:
        ex.vit.set_option("full_last_block", True)
        try:
            step(0, feats[:B])
            torch.cuda.synchronize(device)
            t1 = time.perf_counter()
            for i in range(K):
                step(i, feats[i * B:(i + 1) * B])
            torch.cuda.synchronize(device)
            full_value = K * B / (time.perf_counter() - t1)
        finally:
            ex.vit.set_option("full_last_block", False)

    # ... and with the float32 residual stream + standalone add+LayerNorm launches (round 1's dataflow, option f32_stream)
    f32s = None
    fused = short != "f32" and not os.environ.get("AP_VIT_F32_STREAM")
    if world == 1 and fused:
        ex.vit.set_option("f32_stream", True)
        try:
            step(0, feats[:B])
            torch.cuda.synchronize(device)
            ex.vit.profile(True)
            t1 = time.perf_counter()
            for i in range(K):
                step(i, feats[i * B:(i + 1) * B])
            torch.cuda.synchronize(device)
            dt1 = time.perf_counter() - t1
            p1 = ex.vit.profile_read()
            ex.vit.profile(False)
            f32s = {"patches_per_s": round(K * B / dt1, 1), "kernel_ms_per_step": {k: round(v[0] / K, 4) for k, v in p1.items()}}
        finally:
            ex.vit.set_option("f32_stream", False)

    # ... and without the class rows' float32 side stream (option exact_cls off: the plain 16-bit stream of rounds 2-3)
    plain16 = None
    exact_cls_on = fused and not os.environ.get("AP_VIT_NO_EXACT_CLS") and arch.get("pool", "cls") in ("cls", "cls_mean")
    if world == 1 and exact_cls_on:
        ex.vit.set_option("exact_cls", False)
        try:
            step(0, feats[:B])
            torch.cuda.synchronize(device)
            t1 = time.perf_counter()
            for i in range(K):
                step(i, feats[i * B:(i + 1) * B])
            torch.cuda.synchronize(device)
            plain16 = {"patches_per_s": round(K * B / (time.perf_counter() - t1), 1)}
        finally:
            ex.vit.set_option("exact_cls", True)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    value = world * K * B / elapsed
    # ---- roofline of the dominant kernel: the fc1 GEMM (gemm256_kernel<T, EPI_NORM_GELU>: LayerNorm statistics + GELU in
    #      the epilogue; <T, EPI_BIAS_GELU> with the f32-stream dataflow), one shape per launch
    fc1_tag = "Li5E" if fused else "Li1E"
    M = B * T
    fc1_ms, fc1_n = prof["gemm_fc1"]
    flop_launch = 2.0 * M * MLP * D
    fc1_avg_s = (fc1_ms / max(1, fc1_n)) * 1e-3
    achieved = flop_launch / fc1_avg_s / 1e12 if fc1_n else 0.0
    peak = MFMA_PEAK[short] / 1e12
    # --precision float32 runs the split-f16 products unless AP_VIT_EXACT_F32 is set: three f16 MFMA passes per product, priced
    # against the dense f16 peak on the flop the kernel EXECUTES (3 x 2 M N K)
    f32_split = short == "f32" and not os.environ.get("AP_VIT_EXACT_F32")
    if f32_split:
        achieved, peak = 3.0 * achieved, MFMA_PEAK["f16"] / 1e12
    # HBM bytes of that kernel from the PMC counters (separate rocprofv3 --pmc passes of this same command,
    # summarised by tools/pmc_traffic.py into profiles/; FETCH_SIZE x2 on gfx950 per MI355X_MICROARCH.md)
    traffic = None
    import glob
    tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))      # the latest round's PMC summary
    tpath = tfiles[-1] if tfiles else ""
    pmc_shape = short != "f32" and B == 2048 and args.encoder == "vit_b_16"      # the shape the committed PMC passes ran
    if pmc_shape and tpath and os.path.exists(tpath):
        with open(tpath) as fh:
            for kname, rec in json.load(fh).items():
                if "gemm256_kernel" in kname and fc1_tag in kname and ("DF16_" in kname) == (short == "f16"):
                    traffic = rec["hbm_bytes_per_launch"]
    # MFMA-pipe utilisation and effective shader clock of the same kernel from the PMC pass of the same command
    # (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE; tools/pmc_mfma.py -> profiles/): frac ~= mfma_util * clock / 2.4 GHz
    pmc_mfma = None
    mfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_mfma_util.json")))
    if pmc_shape and mfiles:
        with open(mfiles[-1]) as fh:
            for kname, rec in json.load(fh).items():
                if "gemm256_kernel" in kname and fc1_tag in kname and ("DF16_" in kname) == (short == "f16"):
                    pmc_mfma = {"mfma_util": round(rec["mfma_util"], 4),
                                "effective_clock_GHz": round(rec.get("effective_clock_GHz", 0.0), 3),
                                "source": os.path.basename(mfiles[-1])}
    kernel_ms = {k: round(v[0] / K, 4) for k, v in prof.items()}
    # the HBM-bound kernels of the path, same HIP-event timers, algorithmic bytes:
    #   preprocess: reads the S x S x 3 u8 centre crop, writes the [T - 1, 3 * 16 * 16] patch matrix in the compute dtype
    #   add+LayerNorm: per block LN1 (f32 stream + one pending branch in, normalised rows out; the first block has no
    #   pending branch) and LN2 (stream + two branches in, stream + normalised rows out); the final LN touches CLS rows only
    eb = 2.0 if short != "f32" else 4.0
    L = arch["depth"]
    pre_bytes = B * 3.0 * arch["image_size"] ** 2 * (1.0 + eb)
    tail = cls_tail_possible and not full_last
    ln_blocks = L - 1 if tail else L             # the last block's LN2 runs on the CLS rows only (timed under cls_tail)
    ln_bytes = M * float(D) * (ln_blocks * ((4 + eb + eb) + (4 + 2 * eb + 4 + eb)) + (4 + eb + eb if tail else 0) - eb)
    if fused:
        # fused dataflow: no add+LayerNorm pass and no stream initialisation pass (the patch-embed GEMM writes the T stream and
        # its partial sums).  What is timed under "layernorm" are the statistic finalisations -- one after the patch
        # embedding, two per fused block minus the last ([M, D/64, 2] f32 partial sums in, [M, 2] out) -- and the class-token
        # rows; the stream itself moves inside the proj / fc2 epilogues (2 B read + 2 B written per element)
        ln_bytes = (2 * L - 1 if tail else 2 * L) * M * (D / 64 * 8 + 8.0) + B * float(D) * (8 + eb)
    hbm_kernels = {}
    for kind, nbytes in (("preproc", pre_bytes), ("layernorm", ln_bytes)):
        ms = prof[kind][0] / K
        if ms > 0:
            hbm_kernels[kind] = {"algorithmic_bytes_per_step": nbytes, "ms_per_step": round(ms, 4),
                                 "achieved_GBps": round(nbytes / (ms * 1e-3) / 1e9, 1), "peak_GBps": 8000.0,
                                 "frac": round(nbytes / (ms * 1e-3) / 8e12, 4)}
    shape = f"[B*{T},{D}]x[{D},{MLP}]"
    flop_exec = geo["model"] if full_last else geo["executed"]
    line = {
        "metric": f"patches/sec embedded (256x256, {'ViT-B/16' if args.encoder == 'vit_b_16' else args.encoder})",
        "value": round(value, 1), "unit": "patches/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(elapsed / K * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": short,
        "data": "synthetic" if not rehearsal else "synthetic -- REHEARSAL of the N > 1 code path (ranks share one GPU, gloo): not a measurement",
        "config": {"workload": f"process: one synthetic {args.slide}x{args.slide} slide per rank, 256x256 tiles "
                               f"resident in HBM, {enc['label']} (random-init), device batch {B}",
                   "encoder": args.encoder,
                   "tiles_per_step": B, "slide_tissue_tiles": int(n_slide), "grid_cells": cells,
                   "parallelism": f"slide-per-rank x{world}" + (" + RCCL all-gather of features" if world > 1 else ""),
                   "gpus_requested": args.gpus, "ranks_in_process_group": world},
        "roofline": {"bound": "mfma", "kernel": ((f"gemm256_kernel<T,EPI_NORM_GELU> (fc1 with the LayerNorm statistics applied "
                                                          f"in the epilogue: {shape}, " if fused else
                                                          f"gemm256_kernel<T,EPI_BIAS_GELU> (fc1: {shape}, ") +
                                                         "persistent 256x256-tile MFMA GEMM)") if short != "f32" else
                                                        ((f"gemm_kernel<float,EPI_BIAS_GELU,split> (fc1: {shape}, 128x128-tile split-f16 GEMM: "
                                                          "3 x v_mfma_f32_32x32x16_f16 per product, `achieved` = executed flop)") if f32_split else
                                                         (f"gemm_kernel<float,EPI_BIAS_GELU> (fc1: {shape}, "
                                                          "128x128-tile v_mfma_f32_32x32x2_f32 GEMM)")),
                     "achieved": round(achieved, 1), "peak": round(peak, 1), "unit": "TFLOP/s",
                     "frac": round(achieved / peak, 4), "traffic": traffic,
                     "traffic_source": (f"profiles/{os.path.basename(tpath)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                        "this command, committed; not measured in this run)") if traffic is not None else None,
                     "pmc": pmc_mfma,
                     "algorithmic_bytes_per_launch": (M * D + MLP * D + M * MLP) * (4.0 if short == "f32" else 2.0) +
                                                     (M * 8.0 + MLP * 8.0 if fused else 0.0),
                     "avg_launch_ms": round(fc1_avg_s * 1e3, 4), "launches": fc1_n,
                     "algorithmic_flop_per_launch": flop_launch,
                     "executed_flop_per_launch": flop_launch * (3.0 if f32_split else 1.0)},
        "clock": clock_info,
        "power": {**power_info,
                  "joules_per_step": (None if not power_info.get("package_W_mean") else
                                      round(power_info["package_W_mean"] * elapsed / K, 2)),
                  "what": "rank 0's package while the K timed steps ran; the MFMA kernels sit at the 1400 W cap, so value tracks "
                          "clock.shader_clock_GHz: compare that across runs before comparing value"},
        "end_to_end_model_tflops": round(value * flop_exec / 1e12 / world, 1),
        "flop_per_patch": {"model": geo["model"], "executed": flop_exec,
                           "note": "last block: K/V for all tokens, the rest for the CLS row only (identical features); "
                                   "option full_last_block (AP_VIT_FULL_LAST_BLOCK=1 at start) computes it for every token"
                                   if cls_tail_possible else "every block for every token (attentional pooler reads all tokens)",
                           "value_with_full_last_block": None if full_value is None else round(full_value, 1)},
        "dataflow": ("fused_layernorm: residual stream in the compute type, LayerNorm folded into the qkv / fc1 GEMMs, residual "
                     "add + row sums in the proj / fc2 epilogues" + ("; the class rows' stream also carried in float32 (exact_cls)"
                                                                     if exact_cls_on else "")
                     if fused else "f32 residual stream + add+LayerNorm launches"),
        "plain_16bit_stream_dataflow": plain16,
        "f32_stream_dataflow": f32s,
        "kernel_ms_per_step": kernel_ms, "hbm_kernels": hbm_kernels,
        "coords": {"cells": cells, "rows": int(n_slide), **coords_stats,
                   "cells_per_s": round(cells / coords_stats["warm_seconds_median"], 1),
                   "what": "mask -> contours -> grid scan -> rows on the host, one slide; warm = median of 10 repeats after the "
                           "first (cold) call"},
    }
    if per_rank is not None:
        line["per_rank"] = per_rank
        line["all_gather"] = {"bytes_per_rank": int(K * B * ex.embedding_dim * 4), "ms_max_over_ranks": max(r["all_gather_ms"] for r in per_rank),
                              "algorithm": gather_algo,
                              "what": ("one all_gather_into_tensor of every rank's float32 [K*B, D] block" if gather_algo != "pairs" else
                                       "all-pairs batch_isend_irecv of every rank's float32 [K*B, D] block (ATLASPATCH_GATHER_ALGO=pairs)")
                                      + ", inside the timed region"}
    if not args.no_cpu_baseline and world == 1:          # the CPU leg runs at N = 1 only
        sample = args.cpu_sample if args.cpu_sample else {"vit_b_16": 192, "uni_v1": 64, "conch_v1": 32, "vit_b_32": 256,
                                                          "vit_l_32": 128, "vit_h_14": 4, "uni_v2": 24}[args.encoder]
        out_cpu, patches, cpu_rate, cpu_threads = cpu_baseline(args.encoder, sample, seed=0)
        # parity of the measured path against the CPU oracle on the same sample (reported, not timed)
        relf = lambda got: float(np.linalg.norm(got.astype(np.float64) - out_cpu) / np.linalg.norm(out_cpu))
        rel = relf(ex.extract_batch(patches, batch_size=32))
        line["cpu_baseline"] = {"value": round(cpu_rate, 2), "unit": "patches/s",
                                "cores": int(cpu_threads), "kind": "port",
                                "sample": f"{sample} random 256x256 tiles, {enc['label']} fp32, torch CPU oracle "
                                          f"(oracle/vit_oracle.py), at the fastest of 8/16/32/64 torch threads "
                                          f"(host has {os.cpu_count()} hardware threads)",
                                "rel_err_gpu_vs_cpu": rel}
        # ... and of the other modes of the same encoder on the same sample: the f32-stream dataflow, the exact-f32 mode
        errs = {("fused_layernorm" if fused else "f32_stream") + "_" + short: rel}
        if fused:
            ex.vit.set_option("f32_stream", True)
            try:
                errs["f32_stream_" + short] = relf(ex.extract_batch(patches, batch_size=32))
            finally:
                ex.vit.set_option("f32_stream", False)
            if exact_cls_on:
                ex.vit.set_option("exact_cls", False)
                try:
                    errs["plain_16bit_stream_" + short] = relf(ex.extract_batch(patches, batch_size=32))
                finally:
                    ex.vit.set_option("exact_cls", True)
        f32_ok = arch.get("pool") != "attn" and geo["tokens"] <= 288 and arch["dim"] // arch["heads"] == 64
        if short != "f32" and f32_ok:          # conch_v1 / vit_h_14: f16 / bf16 only (the f32 attention kernel: 288 tokens, 64 wide)
            ex32 = build_hip_vit_extractor(name=args.encoder, arch=args.encoder, device=device, dtype=torch.float32,
                                           random_init_seed=0, max_batch=min(B, 256), resize=TRANSFORM_RESIZE[args.encoder],
                                           expect_size=None, mean=enc.get("mean"), std=enc.get("std"))
            elemf = lambda got: float((np.abs(got.astype(np.float64) - out_cpu) / (np.abs(out_cpu) + 0.05 * np.abs(out_cpu).max())).max())
            got32 = ex32.extract_batch(patches, batch_size=32)
            errs["float32_mode"] = relf(got32)
            errs["float32_mode_elementwise_max"] = elemf(got32)
            ex32.vit.set_option("split_f16", False)
            got32 = ex32.extract_batch(patches, batch_size=32)
            errs["float32_exact_mode"] = relf(got32)
            errs["float32_exact_mode_elementwise_max"] = elemf(got32)
            ex32.cleanup()
        line["cpu_baseline"]["rel_err_by_mode"] = errs
        cpu_coords, cpu_coords_s = cpu_coords_baseline(args.slide, args.slide_seed + rank)
        line["cpu_baseline"]["coords"] = {"cells_per_s": round(cells / cpu_coords_s, 1), "seconds": round(cpu_coords_s, 3),
                                          "cores": 1, "kind": "port",
                                          "sample": "the bench slide's mask through oracle/coords_oracle.py (NumPy)",
                                          "rows_equal_device_path": bool(np.array_equal(np.asarray(cpu_coords), np.asarray(dev_coords)))}
    if not args.no_extras and world == 1 and args.encoder == "vit_b_16":
        line["rates"] = {"kernel_only": {"patches_per_s": round(value, 1), "what": "= value"}}
        line["rates"].update(secondary_rates(device, ex, tiles, B))
        seg = line["rates"].get("sam2_segmentation")
        if seg:
            # config 1's hot kernel chain priced like `roofline`: the forward's multiply-add work over its replay time against
            # the exact-f32 MFMA peak (the reference runs SAM2 in float32: services/segmentation.py:120-180 has no autocast)
            line["sam2"] = {"bound": "mfma", "dtype": "f32", "gflop_per_slide": seg["gflop_per_slide"], "ms_per_slide": seg["ms_per_slide_device"],
                            "ms_per_slide_seg_batch_4": seg["ms_per_slide_device_seg_batch_4"], "achieved": seg["TFLOPs"],
                            "achieved_seg_batch_4": seg["TFLOPs_seg_batch_4"], "peak": seg["peak_TFLOPs"], "unit": "TFLOP/s",
                            "frac": seg["frac"], "frac_seg_batch_4": seg["frac_seg_batch_4"]}
    # every arithmetic mode next to its error vs the CPU fp32 path: `value` is the reference CLI's default precision (f16,
    # cli.py:175-181), the north star's 1e-3 is met by the modes marked so
    errs = (line.get("cpu_baseline") or {}).get("rel_err_by_mode") or {}
    modes = {}
    main_key = ("fused_layernorm" if fused else "f32_stream") + "_" + short
    modes[main_key] = {"patches_per_s": round(value, 1), "rel_err_vs_cpu_fp32": errs.get(main_key), "is_value": True}
    if f32s is not None:
        modes["f32_stream_" + short] = {"patches_per_s": f32s["patches_per_s"], "rel_err_vs_cpu_fp32": errs.get("f32_stream_" + short)}
    if plain16 is not None:
        modes["plain_16bit_stream_" + short] = {"patches_per_s": plain16["patches_per_s"],
                                                "rel_err_vs_cpu_fp32": errs.get("plain_16bit_stream_" + short)}
    for key in ("float32_mode", "float32_exact_mode"):
        f32m = (line.get("rates") or {}).get(key)
        if f32m:
            modes[key] = {"patches_per_s": f32m["patches_per_s"], "rel_err_vs_cpu_fp32": errs.get(key),
                          "elementwise_max_vs_cpu_fp32": errs.get(key + "_elementwise_max"),
                          "meets_1e-3_elementwise": (errs.get(key + "_elementwise_max") is not None and errs[key + "_elementwise_max"] <= 1e-3)}
    for m in modes.values():
        m["meets_1e-3"] = (m["rel_err_vs_cpu_fp32"] is not None and m["rel_err_vs_cpu_fp32"] <= 1e-3) if m["rel_err_vs_cpu_fp32"] is not None else None
    line["parity_modes"] = modes
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
