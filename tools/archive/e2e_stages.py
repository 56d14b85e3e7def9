#!/usr/bin/env python3
"""usage: e2e_stages.py [side=40000,100000] [host_tiles=0|1] [repeat=2]
Stage breakdown (utils/stages.py) of `process` on a synthetic slide, weights loaded from a safetensors file like a real
checkpoint.  One JSON line per run: tiles, seconds, tiles/s, stages."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sides = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "40000,100000").split(",")]
host = len(sys.argv) > 2 and sys.argv[2] == "1"
repeat = int(sys.argv[3]) if len(sys.argv) > 3 else 2
import torch
from click.testing import CliRunner
from safetensors.torch import save_file
from atlaspatch_amd.cli import cli
from atlaspatch_amd.encoders.vit import ARCHS, random_canonical_state_dict
from atlaspatch_amd.utils import stages
from atlaspatch_amd.utils.h5 import h5
torch.zeros(1, device="cuda")
with tempfile.TemporaryDirectory() as tmp:
    save_file(random_canonical_state_dict(ARCHS["vit_b_16"], 0), os.path.join(tmp, "vit_b_16.safetensors"))
    os.environ["ATLASPATCH_WEIGHTS_DIR"] = tmp
    os.environ.pop("ATLASPATCH_RANDOM_INIT", None)
    if host:
        os.environ["ATLASPATCH_HOST_TILES"] = "1"
    for side in sides:
        slide = os.path.join(tmp, f"s{side}.synth")
        json.dump({"width": side, "height": side, "seed": 1234, "mag": 20, "mpp": 0.5, "downsamples": [1, 4, 16]}, open(slide, "w"))
        for rep in range(repeat):
            out = os.path.join(tmp, f"out{side}_{rep}")
            stages.snapshot(reset=True)
            t0 = time.perf_counter()
            res = CliRunner().invoke(cli, ["process", slide, "-o", out, "--patch-size", "256", "--target-mag", "20",
                                           "--feature-extractors", "vit_b_16", "--feature-precision", "float16",
                                           "--feature-num-workers", str(min(64, os.cpu_count() or 8))], catch_exceptions=False)
            dt = time.perf_counter() - t0
            assert res.exit_code == 0 and "failures: 0" in res.output, res.output
            with h5.File(os.path.join(out, "patches", f"s{side}.h5"), "r") as f:
                n = int(f["coords"].shape[0])
            print(json.dumps({"side": side, "host_tiles": host, "rep": rep, "tiles": n, "seconds": round(dt, 3),
                              "tiles_per_s": round(n / dt, 1), "stages": stages.snapshot(reset=True)}), flush=True)
