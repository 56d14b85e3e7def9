#!/usr/bin/env python3
"""Timing ablations of the pipelined attention kernel: tools/attn_exp/libattn_abl<mask>.so (tools/attn_exp/build.sh <masks>) timed
interleaved in one process on the 197- and 785-token shapes.  Masks (AP_PIPE_ABL, results invalid when set): 1 no exponentials,
2 no Q K^T MFMAs, 4 no P V MFMAs, 8 no K fragment reads, 16 no V fragment reads, 32 no range check / redo vote, 64 no LDS-DMA
staging, 128 no tile barriers.  AP_ATTN_IMPL=flash in the environment times the tiled kernel through the same wrapper."""
import ctypes, glob, os, re, sys
import torch
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "attn_exp")
libs = {}
for path in sorted(glob.glob(os.path.join(here, "libattn_abl*.so")), key=lambda p: int(re.findall(r"abl(\d+)", p)[0])):
    lib = ctypes.CDLL(path)
    lib.attn_exp.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    libs[int(re.findall(r"abl(\d+)", path)[0])] = lib
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream(dev).cuda_stream
g = torch.Generator(device=dev).manual_seed(0)
for (n, T, H) in ((1024, 197, 12), (256, 785, 12)):
    qkv = torch.randn((n * T, 3 * H * 64), device=dev, generator=g).half()
    out = torch.empty((n * T, H * 64), device=dev, dtype=torch.float16)
    times = {m: [] for m in libs}
    for rep in range(7):
        for m, lib in libs.items():
            for _ in range(2):
                lib.attn_exp(1, qkv.data_ptr(), out.data_ptr(), n, T, H, stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                lib.attn_exp(1, qkv.data_ptr(), out.data_ptr(), n, T, H, stream)
            e1.record(); torch.cuda.synchronize()
            times[m].append(e0.elapsed_time(e1) / 5)
    base = sorted(times[0])[3] if 0 in times else None
    for m in libs:
        ms = sorted(times[m])[3]
        print(f"T={T} n={n} mask={m:3d}: {ms:.4f} ms" + (f"  ({(ms / base - 1) * 100:+.1f} %)" if base else ""), flush=True)
