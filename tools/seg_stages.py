#!/usr/bin/env python3
"""usage: seg_stages.py [n_slides=16] [side=100000] [seg_batch=1]
`segment-and-get-coords` on n synthetic slides with the SAM2 segmenter forced (seeded random weights: arbitrary masks):
slides/s and the stage breakdown (utils/stages.py)."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
side = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
segb = int(sys.argv[3]) if len(sys.argv) > 3 else 1
os.environ["ATLASPATCH_SEGMENTER"] = "sam2"
import torch
if os.environ.get('AP_SWITCH'): sys.setswitchinterval(float(os.environ['AP_SWITCH']))
from click.testing import CliRunner
from atlaspatch_amd.cli import cli
from atlaspatch_amd.utils import stages
torch.zeros(1, device="cuda")
with tempfile.TemporaryDirectory() as tmp:
    # the SAM2 weights come from a checkpoint file, as in bench.py (generating 38 M random parameters on the host takes ~1.7 s)
    from atlaspatch_amd.services.segmentation import random_sam2_state_dict
    torch.save({"model": random_sam2_state_dict(0)}, os.path.join(tmp, "sam2.pt"))
    os.environ["ATLASPATCH_WEIGHTS_DIR"] = tmp
    os.makedirs(os.path.join(tmp, "slides"))
    for i in range(n):
        json.dump({"width": side, "height": side, "seed": 100 + i, "mag": 20, "mpp": 0.5, "downsamples": [1, 4, 16]},
                  open(os.path.join(tmp, "slides", f"s{i:03d}.synth"), "w"))
    for rep in range(2):
        stages.snapshot(reset=True)
        t0 = time.perf_counter()
        res = CliRunner().invoke(cli, ["segment-and-get-coords", os.path.join(tmp, "slides"), "-o", os.path.join(tmp, f"out{rep}"),
                                       "--patch-size", "256", "--target-mag", "20", "--seg-batch-size", str(segb)], catch_exceptions=False)
        dt = time.perf_counter() - t0
        assert res.exit_code == 0 and "failures: 0" in res.output, res.output
        print(json.dumps({"slides": n, "side": side, "seg_batch": segb, "rep": rep, "seconds": round(dt, 3), "slides_per_s": round(n / dt, 2),
                          "ms_per_slide": round(dt / n * 1e3, 2), "stages": stages.snapshot(reset=True)}), flush=True)
