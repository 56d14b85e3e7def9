#!/usr/bin/env python3
"""The gemm256 twin (impl 257: `make ALT_FLAGS="-DAP_G256_ALT -DAP_EXP_..."`, the product code plus one experiment) against the product kernel (impl 256): bit-equality of every epilogue
on ragged / multi-tile problems, repeatability, and interleaved A/B timing on the ViT-B shapes with the epilogues the
forward uses (NORM qkv, RESID_STATS proj, NORM_GELU fc1, RESID_STATS fc2).

    python tools/gemm_twin_ab.py [--tiles 2048] [--timing-only] [--quick]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atlaspatch_amd import _lib

lib = _lib.load()
dev = torch.device("cuda:0")
sp = lambda: _lib.current_stream_ptr(dev)


def plain(dt, epi, A, W, bias, gamma, out, impl, variant=0):
    M, K = A.shape
    _lib.check(lib.ap_gemm(_lib.torch_dtype_code(dt), epi, A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0), M, W.shape[0], K,
                           bias.data_ptr(), gamma.data_ptr() if gamma is not None else None, out.data_ptr(), out.stride(0),
                           impl, variant, sp()), "ap_gemm")


def fused(dt, epi, A, W, bias, cs, rs, part, out, impl, variant=0):
    M, K = A.shape
    _lib.check(lib.ap_gemm_fused(_lib.torch_dtype_code(dt), epi, A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0), M, W.shape[0], K,
                                 bias.data_ptr(), cs.data_ptr() if cs is not None else None, rs.data_ptr() if rs is not None else None,
                                 part.data_ptr() if part is not None else None, out.data_ptr(), out.stride(0),
                                 impl | (variant << 12), sp()), "ap_gemm_fused")


def operands(M, N, K, dt, g):
    A = (torch.rand((M, K), device=dev, generator=g) * 2 - 1).to(dt)
    W = ((torch.rand((N, K), device=dev, generator=g) * 2 - 1) * (2.0 / K ** 0.5)).to(dt)
    bias = torch.rand(N, device=dev, generator=g) - 0.5
    cs = W.float().sum(-1).contiguous()
    rs = torch.stack([torch.rand(M, device=dev, generator=g) + 0.5, torch.rand(M, device=dev, generator=g) - 0.5], -1).contiguous()
    X0 = (torch.rand((M, N), device=dev, generator=g) * 4 - 2).to(dt)
    return A, W, bias, cs, rs, X0


def run_epi(name, dt, ops, impl, variant=0):
    A, W, bias, cs, rs, X0 = ops
    M, N = A.shape[0], W.shape[0]
    if name == "bias":
        out = torch.full((M, N), float("nan"), device=dev, dtype=dt); plain(dt, 0, A, W, bias, None, out, impl, variant); return (out,)
    if name == "bias_gamma":
        out = torch.full((M, N), float("nan"), device=dev, dtype=dt); plain(dt, 0, A, W, bias, cs * 0 + rs[:1, 0] + 0.25, out, impl, variant); return (out,)
    if name == "gelu":
        out = torch.full((M, N), float("nan"), device=dev, dtype=dt); plain(dt, 1, A, W, bias, None, out, impl, variant); return (out,)
    if name == "resid_f32":
        out = X0.float().clone(); plain(dt, 2, A, W, bias, None, out, impl, variant); return (out,)
    if name == "norm":
        out = torch.full((M, N), float("nan"), device=dev, dtype=dt); fused(dt, 4, A, W, bias, cs, rs, None, out, impl, variant); return (out,)
    if name == "norm_gelu":
        out = torch.full((M, N), float("nan"), device=dev, dtype=dt); fused(dt, 5, A, W, bias, cs, rs, None, out, impl, variant); return (out,)
    if name == "resid_stats":
        out = X0.clone(); part = torch.full((M, N // 64, 2), float("nan"), device=dev)
        fused(dt, 6, A, W, bias, None, None, part, out, impl, variant); return (out, part)
    raise ValueError(name)


def same(a, b):
    return all(torch.equal(x.view(torch.uint8) if x.dtype != torch.float32 else x, y.view(torch.uint8) if y.dtype != torch.float32 else y)
               or (torch.equal(torch.isnan(x), torch.isnan(y)) and torch.equal(torch.nan_to_num(x.float()), torch.nan_to_num(y.float())))
               for x, y in zip(a, b))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=2048)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--timing-only", action="store_true")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--ablate", default="", help="comma list of ablation masks for the gemm256 twin (impl 257), e.g. 0,8")
    args = ap.parse_args()
    g = torch.Generator(device=dev).manual_seed(0)
    bad = 0
    if not args.timing_only:
        cases = [(300, 256, 128), (1182, 768, 768), (256, 512, 256), (100, 256, 384), (5000, 2304, 768), (3941, 768, 3072),
                 (70000, 768, 768), (40000, 3072, 768), (256 * 40 + 77, 1024, 1024), (150000, 256, 256)]
        for dt in (torch.float16, torch.bfloat16):
            for (M, N, K) in cases:
                ops = operands(M, N, K, dt, g)
                for name in ("bias", "bias_gamma", "gelu", "resid_f32", "norm", "norm_gelu", "resid_stats"):
                    ref = run_epi(name, dt, ops, 256)
                    outs = [run_epi(name, dt, ops, 257, v) for v in (0, 0, 0)]
                    torch.cuda.synchronize()
                    ok = all(same(ref, o) for o in outs)
                    if not ok:
                        bad += 1
                        d = [int((torch.nan_to_num(r.float()) != torch.nan_to_num(o.float())).sum()) for r, o in zip(ref, outs[0])]
                        print(f"FAIL {str(dt)[6:]:9s} {name:12s} M={M} N={N} K={K}: differing elements {d}", flush=True)
            print(f"{str(dt)[6:]}: bit-equality vs impl 256 on {len(cases)} shapes x 7 epilogues: {'FAIL' if bad else 'ok'}", flush=True)
        # repeatability under load: the same launch 20x
        ops = operands(40000, 2304, 768, torch.float16, g)
        first = run_epi("norm", torch.float16, ops, 257)
        rep_bad = sum(0 if same(first, run_epi("norm", torch.float16, ops, 257)) else 1 for _ in range(20))
        ops = operands(40000, 768, 3072, torch.float16, g)
        first = run_epi("resid_stats", torch.float16, ops, 257)
        rep_bad += sum(0 if same(first, run_epi("resid_stats", torch.float16, ops, 257)) else 1 for _ in range(20))
        print("repeatability (40 launches):", "ok" if rep_bad == 0 else f"{rep_bad} DIFFER", flush=True)
        bad += rep_bad
    if args.quick:
        sys.exit(1 if bad else 0)
    M = args.tiles * 197
    impls = [(256, 0), (257, 0)]
    if args.ablate:
        impls += [(257, int(a)) for a in args.ablate.split(",")]
    res = []
    for gname, N, K, epi in (("qkv", 2304, 768, "norm"), ("proj", 768, 768, "resid_stats"), ("fc1", 3072, 768, "norm_gelu"),
                             ("fc2", 768, 3072, "resid_stats"), ("fc1_plain_gelu", 3072, 768, "gelu"), ("qkv_plain", 2304, 768, "bias")):
        ops = operands(M, N, K, torch.float16, g)
        for impl, v in impls:
            for _ in range(3):
                run_epi(epi, torch.float16, ops, impl, v)
        torch.cuda.synchronize()
        # run_epi allocates its outputs: time the launch alone with preallocated buffers
        A, W, bias, cs, rs, X0 = ops
        out = torch.empty((M, N), device=dev, dtype=torch.float16)
        part = torch.empty((M, N // 64, 2), device=dev)
        def launch(impl, v):
            if epi == "norm": fused(torch.float16, 4, A, W, bias, cs, rs, None, out, impl, v)
            elif epi == "norm_gelu": fused(torch.float16, 5, A, W, bias, cs, rs, None, out, impl, v)
            elif epi == "resid_stats": fused(torch.float16, 6, A, W, bias, None, None, part, out, impl, v)
            elif epi == "gelu": plain(torch.float16, 1, A, W, bias, None, out, impl, v)
            else: plain(torch.float16, 0, A, W, bias, None, out, impl, v)
        best = {}
        for rnd in range(args.rounds):
            for impl, v in impls:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    launch(impl, v)
                e1.record()
                torch.cuda.synchronize()
                best.setdefault((impl, v), []).append(e0.elapsed_time(e1) / args.iters)
        for (impl, v), t in best.items():
            ms = sorted(t)[len(t) // 2]
            res.append({"gemm": gname, "impl": impl, "ablate": v if impl == 257 else 0, "ms_median": round(ms, 4),
                        "ms_min": round(min(t), 4), "TF": round(2.0 * M * N * K / ms / 1e9, 1)})
            print(res[-1], flush=True)
    print(json.dumps(res))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
