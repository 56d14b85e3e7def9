#!/usr/bin/env python3
"""ap_sgemm (exact f32 MFMA, the SAM2 operator set's GEMM) against ap_gemm_split_f16 (three f16 MFMA passes on hi / lo halves,
128 x 128 tiles) on every row-wise layer shape of one SAM2 Hiera-T forward: microseconds each, and both errors against float64."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atlaspatch_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0"); st = _lib.current_stream_ptr(dev)
# (M, N, K, epi): epi 0 bias, 1 bias + GELU, 2 bias + residual
shapes = [(65536, 288, 96, 0), (65536, 96, 96, 0), (65536, 384, 96, 1), (65536, 96, 384, 2), (65536, 192, 96, 0), (65536, 576, 96, 0),
          (16384, 192, 192, 0), (16384, 768, 192, 1), (16384, 192, 768, 2), (16384, 576, 192, 0), (16384, 384, 192, 0), (16384, 1152, 192, 0),
          (4096, 384, 384, 2), (4900, 384, 384, 0), (4096, 1152, 384, 0), (4900, 1152, 384, 0), (4096, 1536, 384, 1), (4096, 384, 1536, 2),
          (4096, 768, 384, 0), (4900, 2304, 384, 0), (1024, 768, 768, 2), (1225, 768, 768, 0), (1225, 2304, 768, 0), (1024, 2304, 768, 0),
          (1024, 3072, 768, 1), (1024, 768, 3072, 2), (65536, 256, 96, 0), (16384, 256, 192, 0), (4096, 256, 384, 0), (1024, 256, 768, 0),
          (4096, 128, 256, 0), (4096, 256, 128, 2), (4096, 256, 256, 0), (16384, 128, 64, 0), (65536, 32, 256, 0), (16384, 64, 256, 0)]
for (M, N, K, epi) in shapes:
    A = torch.randn((M, K), device=dev); W = torch.randn((N, K), device=dev) * K ** -0.5; bias = torch.randn(N, device=dev)
    res = torch.randn((M, N), device=dev)
    Ws = torch.empty_like(W)
    _lib.check(lib.ap_split_f16_weights(W.data_ptr(), Ws.data_ptr(), W.numel(), st))
    o1 = torch.empty((M, N), device=dev); o2 = torch.empty((M, N), device=dev)
    def s():
        _lib.check(lib.ap_sgemm(A.data_ptr(), K, 0, W.data_ptr(), K, 0, 0, 1, M, N, K, C.c_float(1.0), bias.data_ptr(), 1 if epi == 1 else 0,
                                res.data_ptr() if epi == 2 else None, N, 0, o1.data_ptr(), N, 0, st))
    def g():
        _lib.check(lib.ap_gemm_split_f16(A.data_ptr(), K, Ws.data_ptr(), M, N, K, bias.data_ptr(), 1 if epi == 1 else 0,
                                         res.data_ptr() if epi == 2 else None, N, o2.data_ptr(), N, st))
    def t(fn):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20 * 1e3
    us_s, us_g = t(s), t(g)
    ref = A.double() @ W.double().t() + bias.double()
    ref = torch.nn.functional.gelu(ref) if epi == 1 else (ref + res.double() if epi == 2 else ref)
    e1 = float((o1.double() - ref).norm() / ref.norm()); e2 = float((o2.double() - ref).norm() / ref.norm())
    print(f"M={M:6d} N={N:5d} K={K:5d} epi{epi}: sgemm {us_s:7.1f} us {2.0*M*N*K/us_s/1e6:6.1f} TF/s err {e1:.1e} | split {us_g:7.1f} us "
          f"{2.0*M*N*K/us_g/1e6:6.1f} TF/s err {e2:.1e}  ratio {us_g/us_s:.2f}", flush=True)
