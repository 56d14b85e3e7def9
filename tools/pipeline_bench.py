#!/usr/bin/env python3
"""usage: pipeline_bench.py [encoder=vit_b_16] [slide_side=40000]
The three rates of SURVEY.md 8(d) on one MI355X (f16, registered encoder incl. its transform, synthetic slide, random-init weights):
   kernel-only       tiles resident in HBM (what bench.py reports as `value`)
   device-pipeline   tiles in pinned host memory -> tile ring (H2D over PCIe) -> features back on the host
   end-to-end        `process` CLI: segmentation + device coords + host tile synthesis ("decode") -> ring -> H5
"""
import json, os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch

os.environ.setdefault("ATLASPATCH_RANDOM_INIT", "0")
from atlaspatch_amd.encoders import build_default_registry
from atlaspatch_amd.services.tile_ring import TileRing

dev = torch.device("cuda:0")
arch = sys.argv[1] if len(sys.argv) > 1 else "vit_b_16"
side = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
ex = build_default_registry(device=dev, dtype=torch.float16).create(arch)
B = min(2048, ex.max_batch)
rng = np.random.default_rng(0)
N = 8 * B
host = rng.integers(0, 256, (N, 256, 256, 3), dtype=np.uint8)
out = {}
# kernel-only
d = torch.from_numpy(host[:B]).to(dev)
o = torch.empty((B, ex.embedding_dim), dtype=torch.float32, device=dev)
for _ in range(2):
    ex.forward_device(d, o)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(6):
    ex.forward_device(d, o)
torch.cuda.synchronize()
out["kernel_only_tiles_per_s"] = round(6 * B / (time.perf_counter() - t0), 1)
# device pipeline
coords = np.stack([np.arange(N), np.zeros(N), np.full(N, 256), np.full(N, 256), np.zeros(N)], 1).astype(np.int32)
for workers in (8, 32):
    ring = TileRing(device=dev, batch=B, patch_size=256, slots=2, workers=workers)
    ring.run(coords[:2 * B], lambda x, y, rw, rh, lv: host[x], lambda t, o_: ex.forward_device(t, o_), ex.embedding_dim)
    t0 = time.perf_counter()
    feats = ring.run(coords, lambda x, y, rw, rh, lv: host[x], lambda t, o_: ex.forward_device(t, o_), ex.embedding_dim)
    dt = time.perf_counter() - t0
    ring.close()
    out[f"device_pipeline_tiles_per_s_workers{workers}"] = round(N / dt, 1)
ex.cleanup()
# end to end through the CLI
from click.testing import CliRunner
from atlaspatch_amd.cli import cli
with tempfile.TemporaryDirectory() as tmp:
    slide = os.path.join(tmp, "big.synth")
    json.dump({"width": side, "height": side, "seed": 1234, "mag": 20, "mpp": 0.5, "downsamples": [1, 4, 16]}, open(slide, "w"))
    t0 = time.perf_counter()
    res = CliRunner().invoke(cli, ["process", slide, "-o", os.path.join(tmp, "out"), "--patch-size", "256", "--target-mag", "20",
                                   "--feature-extractors", arch, "--feature-precision", "float16", "--feature-num-workers", "32",
                                   "--feature-batch-size", str(B)], catch_exceptions=False)
    dt = time.perf_counter() - t0
    assert res.exit_code == 0, res.output
    from atlaspatch_amd.utils.h5 import h5
    with h5.File(os.path.join(tmp, "out", "patches", "big.h5"), "r") as f:
        n = f["coords"].shape[0]
    out["end_to_end_tiles"] = int(n)
    out["end_to_end_seconds"] = round(dt, 2)
    out["end_to_end_tiles_per_s"] = round(n / dt, 1)
out["host_threads"] = os.cpu_count()
print(json.dumps(out))
