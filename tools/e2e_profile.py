#!/usr/bin/env python3
"""usage: e2e_profile.py [encoder=vit_b_16] [slide_side=100000] [host_tiles=0] [weights=random|file]
weights=file: the seeded weights are first written to <tmp>/<encoder>.safetensors (untimed) and the run loads them through
ATLASPATCH_WEIGHTS_DIR like a real checkpoint, instead of generating 86-303 M random parameters on the host inside the run.
Where the wall time of one `process` CLI run goes (cProfile, cumulative, top functions of this package)."""
import cProfile, io, json, os, pstats, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ATLASPATCH_RANDOM_INIT", "0")
arch = sys.argv[1] if len(sys.argv) > 1 else "vit_b_16"
side = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
if len(sys.argv) > 3 and sys.argv[3] == "1":
    os.environ["ATLASPATCH_HOST_TILES"] = "1"
import torch
from click.testing import CliRunner
from atlaspatch_amd.cli import cli
torch.zeros(1, device="cuda")            # context creation is not the pipeline's cost
with tempfile.TemporaryDirectory() as tmp:
    if len(sys.argv) > 4 and sys.argv[4] == "file":
        from safetensors.torch import save_file
        from atlaspatch_amd.encoders.vit import ARCHS, random_canonical_state_dict
        save_file(random_canonical_state_dict(ARCHS[arch], 0), os.path.join(tmp, f"{arch}.safetensors"))
        os.environ["ATLASPATCH_WEIGHTS_DIR"] = tmp
        os.environ.pop("ATLASPATCH_RANDOM_INIT", None)
    slide = os.path.join(tmp, "big.synth")
    json.dump({"width": side, "height": side, "seed": 1234, "mag": 20, "mpp": 0.5, "downsamples": [1, 4, 16]}, open(slide, "w"))
    args = ["process", slide, "-o", os.path.join(tmp, "out"), "--patch-size", "256", "--target-mag", "20",
            "--feature-extractors", arch, "--feature-precision", "float16", "--feature-num-workers", "32",
            "--feature-batch-size", "1024"]
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    res = CliRunner().invoke(cli, args, catch_exceptions=False)
    pr.disable()
    dt = time.perf_counter() - t0
    assert res.exit_code == 0, res.output
    from atlaspatch_amd.utils.h5 import h5
    with h5.File(os.path.join(tmp, "out", "patches", "big.h5"), "r") as f:
        n = f["coords"].shape[0]
buf = io.StringIO()
pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(r"atlaspatch_amd|torch/nn/init|synchronize|\.cpu|to\b", 45)
print(buf.getvalue()[:9000])
print(json.dumps({"encoder": arch, "side": side, "tiles": int(n), "seconds": round(dt, 2), "tiles_per_s": round(n / dt, 1)}))
