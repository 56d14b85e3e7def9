#!/usr/bin/env python3
"""Operator-level check of the fused-LayerNorm GEMM epilogues against torch (same rounded operands), and run-to-run determinism."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from atlaspatch_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
sp = _lib.current_stream_ptr
def run(M, N, K, dt, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    tdt = torch.float16 if dt == _lib.AP_F16 else torch.bfloat16
    A = (torch.randn(M, K, generator=g) * 1.0).to(tdt).to(dev)
    W = (torch.randn(N, K, generator=g) * 0.05).to(tdt).to(dev)
    bias = (torch.randn(N, generator=g) * 0.1).to(dev)
    X0 = (torch.randn(M, N, generator=g) * 2.0 + 0.3).to(tdt).to(dev)
    bad = 0
    # --- RESID_STATS
    outs = []
    for rep in range(4):
        X = X0.clone(); part = torch.full((M, N // 64, 2), float("nan"), device=dev)
        _lib.check(lib.ap_gemm_fused(dt, 6, A.data_ptr(), K, W.data_ptr(), K, M, N, K, bias.data_ptr(), None, None, part.data_ptr(),
                                     X.data_ptr(), N, 0, sp()), "resid")
        torch.cuda.synchronize(); outs.append((X, part))
    for X, part in outs[1:]:
        if not (torch.equal(X, outs[0][0]) and torch.equal(part, outs[0][1])):
            bad += 1; print("RESID nondeterministic", M, N, K, int((X != outs[0][0]).sum()), int((part != outs[0][1]).sum()))
    X, part = outs[0]
    delta = (A.float() @ W.float().T + bias).to(tdt)
    want = (X0.float() + delta.float()).to(tdt)
    d = (X.float() - want.float()).abs().max().item()
    nbad = int((X != want).sum())
    ps = X.float().view(M, N // 64, 64).sum(-1); pq = (X.float() ** 2).view(M, N // 64, 64).sum(-1)
    es = (part[..., 0] - ps).abs().max().item(); eq = ((part[..., 1] - pq).abs() / pq.abs().clamp_min(1)).max().item()
    print(f"RESID {M}x{N}x{K}: max|x-want| {d:.3e} ({nbad} of {M*N} differ by rounding of acc), stats err sum {es:.2e} sumsq rel {eq:.2e}", flush=True)
    # --- NORM
    x = X
    rs = torch.empty((M, 2), device=dev)
    _lib.check(lib.ap_rowstats_finalize(part.data_ptr(), M, N // 64, N, 1e-6, rs.data_ptr(), sp()), "fin")
    xf = x.float(); mean = xf.mean(-1); var = xf.var(-1, unbiased=False); rstd = torch.rsqrt(var + 1e-6)
    e1 = ((rs[:, 0] - rstd).abs() / rstd).max().item(); e2 = (rs[:, 1] + mean * rstd).abs().max().item()
    print(f"  rowstats: rstd rel {e1:.2e}, -mean*rstd abs {e2:.2e}")
    N2 = 768
    W2 = (torch.randn(N2, N, generator=g) * 0.05).to(tdt).to(dev)
    cs = W2.float().sum(-1).contiguous(); b2 = (torch.randn(N2, generator=g) * 0.1).to(dev)
    for epi, name in ((4, "NORM"), (5, "NORM_GELU")):
        outs = []
        for rep in range(4):
            O = torch.empty((M, N2), dtype=tdt, device=dev)
            _lib.check(lib.ap_gemm_fused(dt, epi, x.data_ptr(), N, W2.data_ptr(), N, M, N2, N, b2.data_ptr(), cs.data_ptr(), rs.data_ptr(), None,
                                         O.data_ptr(), N2, 0, sp()), name)
            torch.cuda.synchronize(); outs.append(O)
        for O in outs[1:]:
            if not torch.equal(O, outs[0]):
                bad += 1; print(name, "nondeterministic", int((O != outs[0]).sum()))
        y = ((xf - mean[:, None]) * rstd[:, None]) @ W2.float().T + b2
        if epi == 5: y = torch.nn.functional.gelu(y)
        err = ((outs[0].float() - y).abs() / (y.abs() + 0.05 * y.abs().max())).max().item()
        print(f"  {name} {M}x{N2}x{N}: element-wise err vs f32 LN+Linear {err:.3e}", flush=True)
    return bad
bad = 0
for dt in (_lib.AP_F16, _lib.AP_BF16):
    for (M, N, K) in ((197, 768, 768), (1000, 768, 3072), (256 * 40 + 77, 1024, 1024), (118200, 768, 768)):
        bad += run(M, N, K, dt)
print("bad:", bad)
sys.exit(1 if bad else 0)
