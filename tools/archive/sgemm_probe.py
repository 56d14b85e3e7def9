#!/usr/bin/env python3
"""Time ap_sgemm (exact-f32 MFMA) on SAM2's mid-size shapes; TF/s against the 157.3 TF/s f32 MFMA peak."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atlaspatch_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0"); st = _lib.current_stream_ptr(dev)
shapes = [(1, 4096, 1536, 384, False), (1, 4096, 384, 1536, False), (1, 4096, 384, 384, False), (1, 65536, 96, 384, False),
          (1, 16384, 768, 192, False), (4, 4096, 4096, 96, False), (4, 4096, 96, 4096, True), (1024, 64, 64, 96, False)]
for (b, M, N, K, kn) in shapes:
    A = torch.randn((b, M, K), device=dev); W = torch.randn((b, K, N) if kn else (b, N, K), device=dev)
    out = torch.empty((b, M, N), device=dev)
    def run():
        _lib.check(lib.ap_sgemm(A.data_ptr(), K, M * K, W.data_ptr(), N if kn else K, N * K, 1 if kn else 0, b, M, N, K,
                                C.c_float(1.0), None, 0, None, N, M * N, out.data_ptr(), N, M * N, st))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"b={b} M={M} N={N} K={K} kn={int(kn)}: {us:7.1f} us  {2.0*b*M*N*K/us/1e6:6.1f} TF/s ({2.0*b*M*N*K/us/1e6/157.3*100:4.1f} %)")
