#!/usr/bin/env python
"""Generate tests/golden/hf_sam2.npz: an INDEPENDENT pin of the SAM2.1 Hiera-T image path.

The reference's segmentation arithmetic lives in the ``sam2`` package (facebookresearch/sam2, absent from this image;
/root/reference services/segmentation.py:15,64-68,127-136,160-172, configs/sam2.1_hiera_t.yaml:10-27).  The image does
ship Hugging Face ``transformers`` whose ``transformers.models.sam2.Sam2Model`` is a separate implementation of the same
network: ``Sam2Config()`` is Hiera-T (embed 96, stages [1, 2, 7, 2], global blocks [5, 7, 9], window spec [8, 4, 14, 7],
FPN 256).  This script

  1. builds ``Sam2Model(Sam2Config())`` with ``dynamic_multimask_via_stability=False`` — the reference instantiates the
     yaml with hydra directly (segmentation.py:62-64), not through ``sam2.build_sam.build_sam2`` whose extra overrides
     switch that fallback on, so with ``multimask_output=False`` the decoder returns mask token 0;
  2. loads the seeded facebook-layout weights of ``oracle.sam2_oracle.random_state_dict(seed)`` into it through
     ``atlaspatch_amd.services.sam2_keys.facebook_to_hf`` (strict: every HF tensor the image path reads must be hit);
  3. runs it the way ``SAM2ImagePredictor`` is driven by the reference: thumbnail -> PIL BILINEAR 1024 x 1024 ->
     ToTensor / Normalize(ImageNet) -> box prompt [0, 0, 1024, 1024] -> 256 x 256 logits of the single mask;
  4. stores the thumbnail (PNG bytes), strided samples + Frobenius norms of the three feature levels, the four Hiera
     stage outputs' norms, the full logits, and the key map it used.

Nothing of ``transformers`` travels: only inputs / outputs.  Run from the repo root:  python tests/golden/gen_golden_hf_sam2.py
"""
from __future__ import annotations

import io
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "hf_sam2.npz")
SEED = 3
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def make_thumbnail(h: int, w: int, seed: int) -> np.ndarray:
    """A slide-like RGB thumbnail: bright background, a few textured pink / purple blobs."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.full((h, w, 3), 236.0, np.float32) + rng.normal(0, 3, (h, w, 3)).astype(np.float32)
    for _ in range(5):
        cy, cx = rng.uniform(0.15, 0.85) * h, rng.uniform(0.15, 0.85) * w
        ry, rx = rng.uniform(0.08, 0.25) * h, rng.uniform(0.08, 0.25) * w
        inside = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1.0
        colour = np.array([rng.uniform(150, 215), rng.uniform(70, 150), rng.uniform(140, 200)], np.float32)
        tex = rng.normal(0, 18, (h, w, 1)).astype(np.float32)
        img = np.where(inside[..., None], colour + tex, img)
    return np.clip(img, 0, 255).astype(np.uint8)


def build_hf_model(sd_facebook: dict):
    from transformers.models.sam2 import Sam2Config, Sam2Model

    from atlaspatch_amd.services.sam2_keys import facebook_to_hf

    cfg = Sam2Config()
    cfg.mask_decoder_config.dynamic_multimask_via_stability = False
    torch.manual_seed(0)
    model = Sam2Model(cfg).eval().float()
    hf_sd = facebook_to_hf(sd_facebook)
    own = model.state_dict()
    unknown = sorted(set(hf_sd) - set(own))
    assert not unknown, f"key map produced names Sam2Model does not have: {unknown[:8]}"
    for k, v in hf_sd.items():
        assert tuple(own[k].shape) == tuple(v.shape), (k, tuple(own[k].shape), tuple(v.shape))
    # tensors the map did not fill must be ones the box-prompted single-mask path never reads
    untouched = sorted(set(own) - set(hf_sd))
    allowed = ("prompt_encoder.mask_embed.", "mask_decoder.pred_obj_score_head.")
    assert all(k.startswith(allowed) for k in untouched), [k for k in untouched if not k.startswith(allowed)][:8]
    model.load_state_dict(hf_sd, strict=False)
    return model, hf_sd


@torch.inference_mode()
def main() -> None:
    from PIL import Image

    from oracle import sam2_oracle as so

    sd = so.random_state_dict(SEED)
    model, hf_sd = build_hf_model(sd)

    thumb = make_thumbnail(367, 512, seed=11)                                   # 733 x 1024 (CMU-1) halved: keeps the fixture small
    img = np.asarray(Image.fromarray(thumb).resize((1024, 1024), Image.Resampling.BILINEAR))
    x = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).float().div(255)[None]
    x = (x - torch.tensor(MEAN).view(1, 3, 1, 1)) / torch.tensor(STD).view(1, 3, 1, 1)

    stages = model.vision_encoder.backbone(x).intermediate_hidden_states        # 4 x [1, H, W, C]
    s0, s1, embed = model.get_image_embeddings(x)                               # [1,32,256,256] [1,64,128,128] [1,256,64,64]
    box = torch.tensor([[[0.0, 0.0, 1024.0, 1024.0]]])
    out = model(image_embeddings=[s0, s1, embed], input_boxes=box, multimask_output=False)
    logits = out.pred_masks[0, 0, 0].float().numpy()                            # [256, 256]
    assert logits.shape == (256, 256)
    # the HF forward from pixel_values must agree with the two-step call above
    out2 = model(pixel_values=x, input_boxes=box, multimask_output=False)
    assert torch.equal(out2.pred_masks, out.pred_masks)

    png = io.BytesIO()
    Image.fromarray(thumb).save(png, format="PNG", optimize=True)
    from atlaspatch_amd.services.sam2_keys import facebook_key_to_hf
    keymap = {k: facebook_key_to_hf(k) for k in sd}
    np.savez_compressed(
        OUT,
        thumbnail_png=np.frombuffer(png.getvalue(), np.uint8),
        seed=np.int64(SEED),
        logits=logits.astype(np.float32),
        embed_sample=embed[0, :, ::4, ::4].numpy().astype(np.float32),           # [256, 16, 16]
        s0_sample=s0[0, :, ::16, ::16].numpy().astype(np.float32),               # [32, 16, 16]
        s1_sample=s1[0, :, ::8, ::8].numpy().astype(np.float32),                 # [64, 16, 16]
        norms=np.array([float(embed.norm()), float(s0.norm()), float(s1.norm())], np.float64),
        stage_norms=np.array([float(s.norm()) for s in stages], np.float64),
        stage_samples=np.stack([s[0, ::max(1, s.shape[1] // 8), ::max(1, s.shape[2] // 8), :8].numpy()[:8, :8] for s in stages]).astype(np.float32),
        positive_fraction=np.float64((logits > 0).mean()),
        weights_checksum=np.float64(sum(float(v.double().sum()) for v in sd.values())),   # detects generator drift
        meta=np.frombuffer(json.dumps({
            "generator": "tests/golden/gen_golden_hf_sam2.py",
            "transformers": __import__("transformers").__version__, "torch": torch.__version__,
            "model": "Sam2Model(Sam2Config()), dynamic_multimask_via_stability=False, float32, eager attention",
            "weights": f"oracle.sam2_oracle.random_state_dict({SEED}) through sam2_keys.facebook_to_hf",
            "prompt": "input_boxes [[0, 0, 1024, 1024]], multimask_output=False",
            "keymap": keymap,
        }).encode(), np.uint8),
    )
    print(f"wrote {OUT}: {os.path.getsize(OUT)} bytes; logits range [{logits.min():.3f}, {logits.max():.3f}], "
          f"positive fraction {(logits > 0).mean():.4f}")
    # informational: how the restatement compares (the test asserts this)
    want = so.predict_logits(sd, img)
    rel = float(np.linalg.norm(want - logits) / np.linalg.norm(logits))
    print(f"oracle/sam2_oracle.py vs HF: logits norm-wise {rel:.3e}")


if __name__ == "__main__":
    main()
