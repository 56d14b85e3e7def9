"""CPU rate of the REFERENCE's own PatchFeatureExtractor.extract_batch (CPU baseline (1)), measured
in the build container (the only place /root/reference exists), next to this repository's CPU oracle on the same
model and inputs.  Run once:

    python tests/golden/time_reference_cpu.py        -> tests/golden/reference_cpu_timing.json

The reference runs through the same stub-import harness as gen_golden.py (nothing of it is copied); model =
seeded HF ViTModel with ViT-B/16 shape (12 layers), fp32, batch 32, num_workers=0, 8 torch threads.
"""
from __future__ import annotations

import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent.parent))
import _ref_harness  # noqa: E402

ref = _ref_harness.install()
from transformers import ViTConfig, ViTModel  # noqa: E402

N = 64
torch.set_num_threads(8)
mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)


def preprocess(pil):
    arr = np.asarray(pil, dtype=np.uint8)[16:240, 16:240, :]
    x = torch.from_numpy(arr.copy()).permute(2, 0, 1).to(torch.float32).div(255)
    return x.sub(mean).div(std)


torch.manual_seed(0)
cfg = ViTConfig(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                image_size=224, patch_size=16, layer_norm_eps=1e-6, hidden_act="gelu")
model = ViTModel(cfg, add_pooling_layer=False).eval()


def loader(device, dtype):
    return ref.custom.CustomEncoderComponents(model=model, preprocess=preprocess,
                                              forward_fn=lambda x: model(pixel_values=x).last_hidden_state[:, 0])


reg = ref.registry.PatchFeatureExtractorRegistry()
ref.custom.register_custom_encoder(registry=reg, name="hfvit", embedding_dim=768, loader=loader,
                                   device=torch.device("cpu"), dtype=torch.float32, num_workers=0)
ex = reg.create("hfvit")
rng = np.random.default_rng(0)
patches = [rng.integers(0, 256, (256, 256, 3), dtype=np.uint8) for _ in range(N)]
ex.extract_batch(patches[:32], batch_size=32)
t0 = time.perf_counter()
want = ex.extract_batch(patches, batch_size=32)
t_ref = time.perf_counter() - t0

from oracle import vit_oracle  # noqa: E402
sd = {k: v.detach() for k, v in model.state_dict().items()}      # the oracle reads HF ViTModel key names directly
try:
    vit_oracle.extract_batch(sd, patches[:32], heads=12, batch_size=32)
    t0 = time.perf_counter()
    got = vit_oracle.extract_batch(sd, patches, heads=12, batch_size=32)
    t_or = time.perf_counter() - t0
    rel = float(np.linalg.norm(got - want) / np.linalg.norm(want))
except Exception as exc:  # noqa: BLE001  (key mapping differs: report the reference side only)
    t_or, rel = None, repr(exc)
out = {"host": {"cpus": os.cpu_count(), "torch_threads": 8}, "patches": N, "batch": 32,
       "reference_extract_batch": {"seconds": round(t_ref, 3), "patches_per_s": round(N / t_ref, 2)},
       "oracle_extract_batch": None if t_or is None else {"seconds": round(t_or, 3), "patches_per_s": round(N / t_or, 2)},
       "oracle_vs_reference_rel_err": rel}
print(json.dumps(out, indent=1))
(HERE / "reference_cpu_timing.json").write_text(json.dumps(out, indent=1) + "\n")
