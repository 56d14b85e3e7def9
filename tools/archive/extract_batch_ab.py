import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from atlaspatch_amd.encoders.vit import build_hip_vit_extractor
dev = torch.device("cuda:0")
ex = build_hip_vit_extractor(name="vit_b_16", arch="vit_b_16", device=dev, dtype=torch.float16, random_init_seed=0, max_batch=256)
rng = np.random.default_rng(0)
tiles = [rng.integers(0, 256, (256, 256, 3), dtype=np.uint8) for _ in range(32)]
res = {True: [], False: []}
for rep in range(6):
    for on in (True, False):
        ex.vit.set_option("exact_cls", on)
        for _ in range(20): ex.extract_batch(tiles)
        t0 = time.perf_counter()
        for _ in range(200): ex.extract_batch(tiles)
        res[on].append((time.perf_counter() - t0) / 200 * 1e3)
for on in (True, False):
    ms = sorted(res[on])[2]
    print(f"exact_cls={on}: {ms:.3f} ms per call, {32 / ms * 1e3:.0f} patches/s   all {[round(x, 3) for x in res[on]]}")
