#!/usr/bin/env python3
"""Per-tile timeline of the persistent GEMM (ap_gemm_trace): where a tile's time goes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from atlaspatch_amd import _lib
dev = torch.device("cuda:0"); lib = _lib.load(); stream = _lib.current_stream_ptr(dev)
g = torch.Generator(device=dev).manual_seed(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024 * 197
CASES = (("qkv", 2304, 768, 0, 0), ("fc1", 3072, 768, 1, 0), ("fc1-store-epilogue", 3072, 768, 0, 0), ("fc1-rowmajor-walk", 3072, 768, 1, 12 << 16),
         ("fc1-N2304", 2304, 768, 1, 0), ("fc2", 768, 3072, 0, 0))
for name, N, K, epi, variant in CASES:
    A = (torch.rand((M, K), device=dev, generator=g) * 2 - 1).half()
    W = ((torch.rand((N, K), device=dev, generator=g) * 2 - 1) * (2.0 / K ** 0.5)).half()
    bias = torch.rand(N, device=dev, generator=g) - 0.5
    out = torch.zeros((M, N), device=dev, dtype=torch.float16)
    T = 40
    buf = torch.zeros((256, T, 8), dtype=torch.int64, device=dev)
    for it in range(3):
        if it == 2:
            lib.ap_gemm_trace(buf.data_ptr(), T)
        _lib.check(lib.ap_gemm(1, epi, A.data_ptr(), K, W.data_ptr(), K, M, N, K, bias.data_ptr(), None, out.data_ptr(), N, 257, variant, stream))
    torch.cuda.synchronize()
    lib.ap_gemm_trace(None, 0)
    t = buf.cpu().numpy().astype(np.float64) * 0.01      # us
    ntile = (t[:, :, 0] > 0).sum(1)
    print(f"== {name} N={N} K={K} epi={epi} variant={variant}: tiles/WG min {ntile.min()} max {ntile.max()}")
    start = t[:, 0, 0].min()
    for wg in (0, 1, 8, 100, 255):
        n = ntile[wg]
        main = t[wg, :n, 1] - t[wg, :n, 0]
        drain = t[wg, :n, 2] - t[wg, :n, 1]
        bias_t = t[wg, :n, 3] - t[wg, :n, 2]
        epi_t = t[wg, :n, 4] - t[wg, :n, 3]
        gap = t[wg, 1:n, 0] - t[wg, :n - 1, 4]
        print(f" wg{wg:3d}: start+{t[wg,0,0]-start:6.2f}us  mainloop {main[1:].mean():6.2f} (first {main[0]:6.2f})  drain {drain.mean():5.2f}  bias {bias_t.mean():5.2f}  epilogue {epi_t.mean():5.2f}  gap {gap.mean():5.2f}  total/tile {(t[wg,n-1,4]-t[wg,0,0])/n:6.2f}")
    allmain = np.concatenate([(t[w, 1:ntile[w], 1] - t[w, 1:ntile[w], 0]) for w in range(256)])
    alld = np.concatenate([(t[w, :ntile[w], 2] - t[w, :ntile[w], 1]) for w in range(256)])
    allb = np.concatenate([(t[w, :ntile[w], 3] - t[w, :ntile[w], 2]) for w in range(256)])
    alle = np.concatenate([(t[w, :ntile[w], 4] - t[w, :ntile[w], 3]) for w in range(256)])
    clk = np.concatenate([(t[w, 1:ntile[w], 6] - t[w, 1:ntile[w], 5]) / 0.01 / np.maximum(t[w, 1:ntile[w], 1] - t[w, 1:ntile[w], 0], 1e-9) for w in range(256)])
    print(f" shader clock in the main loop (s_memtime ticks per us): mean {clk.mean():.0f}  p10 {np.percentile(clk,10):.0f}  p90 {np.percentile(clk,90):.0f}")
    print(f" all WGs: mainloop {allmain.mean():.2f} us/tile ({allmain.mean()/(K/64):.3f} us/K-tile)  drain {alld.mean():.2f}  bias {allb.mean():.2f}  epilogue {alle.mean():.2f}; kernel span {t[:,:,4].max()-start:.1f} us")
