#!/usr/bin/env python3
"""The class rows' exact float32 side stream (AP_VIT_OPT_EXACT_CLS, default on) against the plain 16-bit stream and the f32-stream
dataflow: error vs the fp32 CPU oracle, full-last-block agreement, batch-cut invariance, repeatability, and interleaved timing.
usage: exact_cls_check.py [batch=2048] [dtype=float16]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from atlaspatch_amd.encoders.vit import build_hip_vit_extractor
from oracle import vit_oracle

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dtype = getattr(torch, sys.argv[2]) if len(sys.argv) > 2 else torch.float16
dev = torch.device("cuda:0")
model = vit_oracle.make_hf_vit(layers=12)
sd = dict(model.state_dict())
ex = build_hip_vit_extractor(name="hfvit_L12", arch="vit_b_16", depth=12, state_dict=sd, device=dev, dtype=dtype, source="hf")
rng = np.random.default_rng(11)
tiles = [rng.integers(0, 256, (256, 256, 3), dtype=np.uint8) for _ in range(19)]
torch.set_num_threads(min(32, torch.get_num_threads()))
want = vit_oracle.extract_batch(sd, tiles, heads=12, batch_size=32)
def rel(a, b): return float(np.linalg.norm(a - b) / np.linalg.norm(b))
def elem(a, b): return float((np.abs(a - b) / (np.abs(b) + 0.05 * np.abs(b).max())).max())
MODES = {"exact_cls": (False, True), "plain_16bit_stream": (False, False), "f32_stream": (True, False)}
def set_mode(mode):
    f32, exact = MODES[mode]
    ex.vit.set_option("f32_stream", f32)
    ex.vit.set_option("exact_cls", exact)
res = {}
for mode in MODES:
    set_mode(mode)
    got = ex.extract_batch(tiles, batch_size=32)
    res[mode] = {"norm": rel(got, want), "elem": elem(got, want)}
    ex.vit.set_option("full_last_block", True)
    got2 = ex.extract_batch(tiles, batch_size=32)
    ex.vit.set_option("full_last_block", False)
    res[mode]["full_last_block_vs_tail"] = rel(got2, got)
    res[mode]["full_last_block_vs_oracle"] = rel(got2, want)
print(json.dumps({"dtype": str(dtype), "errors_vs_fp32_oracle": res}), flush=True)

set_mode("exact_cls")
n = 600
t = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (n, 256, 256, 3), dtype=np.uint8)).to(dev)
ref = torch.empty((n, ex.embedding_dim), dtype=torch.float32, device=dev)
ex.forward_device(t, ref)
bad = 0
for chunk in (1, 7, 255, 257):
    out = torch.empty_like(ref)
    lim = n if chunk > 1 else 40
    for lo in range(0, lim, chunk):
        ex.forward_device(t[lo:lo + chunk], out[lo:lo + chunk])
    torch.cuda.synchronize()
    same = torch.equal(out[:lim], ref[:lim])
    bad += not same
    print(f"chunks of {chunk}: {'bit-identical' if same else 'DIFFERS max ' + str(float((out[:lim] - ref[:lim]).abs().max()))}", flush=True)
for it in range(20):
    out = torch.empty_like(ref)
    ex.forward_device(t, out)
    torch.cuda.synchronize()
    if not torch.equal(out, ref):
        bad += 1
        print("repeat", it, "DIFFERS", int((out != ref).any(1).sum()), "rows")
print("finite:", bool(torch.isfinite(ref).all()), "bad:", bad, flush=True)

tb = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 256, 256, 3), dtype=np.uint8)).to(dev)
ob = torch.empty((B, ex.embedding_dim), dtype=torch.float32, device=dev)
times = {m: [] for m in MODES}
for rep in range(4):
    for mode in MODES:
        set_mode(mode)
        for _ in range(2):
            ex.forward_device(tb, ob)
        torch.cuda.synchronize()
        K = 8
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(K):
            ex.forward_device(tb, ob)
        e1.record(); torch.cuda.synchronize()
        times[mode].append(e0.elapsed_time(e1) / K)
for mode in MODES:
    ms = sorted(times[mode])[1]
    print(json.dumps({"mode": mode, "batch": B, "ms": round(ms, 3), "tiles_per_s": round(B / ms * 1e3, 1), "all_ms": [round(x, 2) for x in times[mode]]}), flush=True)
set_mode("exact_cls")
ex.vit.profile(True)
for _ in range(5):
    ex.forward_device(tb, ob)
torch.cuda.synchronize()
print(json.dumps({k: round(v[0] / 5, 3) for k, v in ex.vit.profile_read().items()}))
sys.exit(1 if bad else 0)
