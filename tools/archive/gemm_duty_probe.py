#!/usr/bin/env python3
"""Is the fc1 GEMM power-bound?  The SAME kernel and operands (ViT-B fc1 with the LayerNorm + GELU epilogue,
M = 2048 tiles) timed per launch with HIP events (a) back to back and (b) with the GPU left idle for `gap` ms before
every launch (host sleep after a synchronize).  Algorithm, traffic and launch are identical; only the duty cycle
-- hence package power and the clock the governor grants -- changes.  usage: gemm_duty_probe.py [gaps_ms ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atlaspatch_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0"); sp = _lib.current_stream_ptr
M, N, K = 403456, 3072, 768
g = torch.Generator().manual_seed(0)
A = torch.randn(M, K, generator=g).half().to(dev)
W = (torch.randn(N, K, generator=g) * 0.05).half().to(dev)
bias = torch.randn(N, generator=g).to(dev); cs = W.float().sum(-1).contiguous()
O = torch.zeros((M, N), dtype=torch.float16, device=dev); rs = torch.ones((M, 2), device=dev)
def launch():
    _lib.check(lib.ap_gemm_fused(1, 5, A.data_ptr(), K, W.data_ptr(), K, M, N, K, bias.data_ptr(), cs.data_ptr(), rs.data_ptr(),
                                 None, O.data_ptr(), N, 0, sp()), "fc1")
flop = 2.0 * M * N * K
def run(gap_ms, iters):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        if gap_ms > 0:
            torch.cuda.synchronize(); time.sleep(gap_ms * 1e-3)
        a.record(); launch(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    med = ms[len(ms) // 2]
    return {"gap_ms": gap_ms, "launch_ms_median": round(med, 4), "launch_ms_min": round(ms[0], 4), "TFLOPs_median": round(flop / med / 1e9, 1),
            "frac_of_2.5PF": round(flop / med / 1e9 / 2500.0, 4)}
for _ in range(30): launch()                      # bring the package to its steady state first
torch.cuda.synchronize()
gaps = [float(a) for a in sys.argv[1:]] or [0, 1, 2, 5, 10, 20]
res = [run(0.0, 100)] + [run(gp, 40) for gp in gaps if gp > 0] + [run(0.0, 100)]
print(json.dumps({"kernel": "gemm256_kernel<f16, EPI_NORM_GELU>", "shape": [M, N, K], "runs": res}))
