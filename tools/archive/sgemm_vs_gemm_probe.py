#!/usr/bin/env python3
"""ap_sgemm (the SAM2 operator set's exact-f32 MFMA GEMM) against ap_gemm in float32 (the encoder's 128 x 128-tile kernel, same
v_mfma_f32_32x32x2_f32 arithmetic) on the shapes of one SAM2 Hiera-T forward where ap_gemm's constraints hold
(N % 128 == 0, K % 32 == 0): microseconds and TF/s against the 157.3 TF/s f32 MFMA peak."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atlaspatch_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0"); st = _lib.current_stream_ptr(dev)
shapes = [(4096, 1536, 384, 1), (4096, 384, 1536, 2), (16384, 768, 192, 1), (4900, 1152, 384, 0), (4096, 1152, 384, 0), (1024, 768, 3072, 2),
          (4900, 2304, 384, 0), (1024, 3072, 768, 1), (16384, 192, 768, 2), (65536, 576, 96, 0), (16384, 1152, 192, 0), (4096, 128, 256, 0),
          (65536, 384, 96, 1), (65536, 288, 96, 0), (65536, 96, 384, 2), (16384, 576, 192, 0), (4900, 384, 384, 0), (4096, 384, 384, 2),
          (1225, 768, 768, 0), (1225, 2304, 768, 0), (65536, 256, 96, 0), (16384, 384, 192, 0), (4096, 768, 384, 0)]
for (M, N, K, epi) in shapes:          # epi: 0 bias, 1 bias + GELU, 2 bias + residual
    A = torch.randn((M, K), device=dev); W = torch.randn((N, K), device=dev) * K ** -0.5; bias = torch.randn(N, device=dev)
    res = torch.randn((M, N), device=dev)
    o1 = torch.empty((M, N), device=dev); o2 = res.clone() if epi == 2 else torch.empty((M, N), device=dev)
    def s():
        _lib.check(lib.ap_sgemm(A.data_ptr(), K, 0, W.data_ptr(), K, 0, 0, 1, M, N, K, C.c_float(1.0), bias.data_ptr(), 1 if epi == 1 else 0,
                                res.data_ptr() if epi == 2 else None, N, 0, o1.data_ptr(), N, 0, st))
    ok = N % 128 == 0 and K % 32 == 0
    def g():
        _lib.check(lib.ap_gemm(_lib.AP_F32, epi, A.data_ptr(), K, W.data_ptr(), K, M, N, K, bias.data_ptr(), None, o2.data_ptr(), N, 128, 0, st))
    def t(fn):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20 * 1e3
    us_s = t(s)
    line = f"M={M:6d} N={N:5d} K={K:5d} epi{epi}: sgemm {us_s:7.1f} us {2.0*M*N*K/us_s/1e6:6.1f} TF/s"
    if ok:
        us_g = t(g)
        if epi != 2:
            s(); g(); torch.cuda.synchronize()
            line += f" | gemm128 {us_g:7.1f} us {2.0*M*N*K/us_g/1e6:6.1f} TF/s  ratio {us_g/us_s:.2f}  max|diff| {float((o1-o2).abs().max()):.2e}"
        else:
            line += f" | gemm128 {us_g:7.1f} us {2.0*M*N*K/us_g/1e6:6.1f} TF/s  ratio {us_g/us_s:.2f}"
    print(line, flush=True)
