#!/usr/bin/env python3
"""How much of attention (VALU-bound) and LayerNorm (HBM-bound) overlaps when they run on two streams at once?
(Feasibility number for the attention || add+LN pairing listed in DESIGN.md section 7.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atlaspatch_amd import _lib
dev = torch.device("cuda:0"); lib = _lib.load()
n, T, H, D = 1024, 197, 12, 768
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn((n * T, 3 * D), device=dev, generator=g).half()
att = torch.empty((n * T, D), device=dev, dtype=torch.float16)
x = torch.randn((n * T, D), device=dev, generator=g)
gamma = torch.ones(D, device=dev); beta = torch.zeros(D, device=dev)
xn = torch.empty((n * T, D), device=dev, dtype=torch.float16)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def attn(s): _lib.check(lib.ap_attention(1, qkv.data_ptr(), att.data_ptr(), n, T, H, 64, s.cuda_stream))
def ln(s): _lib.check(lib.ap_layernorm(1, x.data_ptr(), D, n * T, D, gamma.data_ptr(), beta.data_ptr(), 1e-6, xn.data_ptr(), s.cuda_stream))
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
def both():
    attn(s1); ln(s2); ln(s2)
def serial():
    attn(s1); s1.synchronize(); ln(s1); ln(s1)
import time
def wall(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
ta = wall(lambda: attn(s1)); tl = wall(lambda: (ln(s2), ln(s2)))
tb = wall(both)
print(f"attention alone {ta:.3f} ms, 2 x layernorm alone {tl:.3f} ms, sum {ta + tl:.3f} ms, both streams at once {tb:.3f} ms "
      f"-> {100 * (ta + tl - tb) / (ta + tl):.0f} % of the sum saved")
