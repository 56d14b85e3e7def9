#!/usr/bin/env python3
"""Correctness + throughput of ap_gemm (128x128 vs persistent 256x256 kernel) on one MI355X.

    python tools/gemm_check.py [--quick] [--iters 20]

Reference = torch fp32 matmul of the same (f16 / bf16-rounded) operands on the GPU.
Operands are uniform random in [-1, 1) (never zeros: DVFS inflates zero-filled numbers)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atlaspatch_amd import _lib

EPI = {"bias": 0, "gelu": 1, "resid": 2}


def run(lib, dt, epi, A, W, bias, gamma, out, impl, variant, stream):
    M, K = A.shape
    N = W.shape[0]
    _lib.check(lib.ap_gemm(_lib.torch_dtype_code(dt), EPI[epi], A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0),
                           M, N, K, bias.data_ptr(), gamma.data_ptr() if gamma is not None else None,
                           out.data_ptr(), out.stride(0), impl, variant, stream), "ap_gemm")


def reference(A, W, bias, gamma, epi, resid):
    C = A.float() @ W.float().t() + bias
    if epi == "gelu":
        C = torch.nn.functional.gelu(C)
    if epi != "gelu" and gamma is not None:
        C = C * gamma
    if epi == "resid":
        C = resid + C
    return C


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--impls", default="128:0,256:0,257:0")
    ap.add_argument("--timing-only", action="store_true")
    ap.add_argument("--tiles", type=int, default=1024, help="device batch the timed shapes correspond to")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    stream = _lib.current_stream_ptr(dev)
    g = torch.Generator(device=dev).manual_seed(0)
    impls = [tuple(int(v) for v in s.split(":")) for s in args.impls.split(",")]
    # ---- correctness: ragged / small / multi-tile problems
    ok = True
    cases = [(300, 256, 128), (1182, 768, 768), (256, 512, 256), (100, 256, 384), (5000, 2304, 768),
             (3941, 768, 3072), (70000, 768, 768)]
    for dt in (() if args.timing_only else (torch.float16, torch.bfloat16)):
        for epi in ("bias", "gelu", "resid"):
            for (M, N, K) in cases:
                A = (torch.rand((M, K), device=dev, generator=g) * 2 - 1).to(dt)
                W = ((torch.rand((N, K), device=dev, generator=g) * 2 - 1) * (2.0 / K ** 0.5)).to(dt)
                bias = torch.rand(N, device=dev, generator=g) - 0.5
                gamma = torch.rand(N, device=dev, generator=g) + 0.5 if epi != "gelu" and M % 2 else None
                resid = torch.rand((M, N), device=dev, generator=g) if epi == "resid" else None
                ref = reference(A, W, bias, gamma, epi, resid)
                for impl, variant in impls:
                    if impl == 258 and epi == "resid":
                        continue
                    out = resid.clone() if epi == "resid" else torch.full((M, N), float("nan"), device=dev, dtype=dt)
                    run(lib, dt, epi, A, W, bias, gamma, out, impl, variant, stream)
                    torch.cuda.synchronize()
                    err = (out.float() - ref).abs().max().item()
                    tol = (2e-2 if dt == torch.bfloat16 else 3e-3) if epi != "resid" else 2e-4
                    bad = not (err <= tol)
                    ok &= not bad
                    if bad:
                        print(f"{'FAIL' if bad else 'ok  '} {str(dt)[6:]:9s} {epi:5s} M={M} N={N} K={K} impl={impl}/{variant} max_abs_err={err:.3e}", flush=True)
    print("correctness:", "PASS" if ok else "FAIL", flush=True)
    # ---- repeatability (race screen): same launch 5x must be bit-identical
    RM = 20480 if args.timing_only else 20000        # a multiple of 256 in timing-only mode (twin experiments on full tiles)
    A = (torch.rand((RM, 768), device=dev, generator=g) * 2 - 1).half()
    W = ((torch.rand((2304, 768), device=dev, generator=g) * 2 - 1) * 0.07).half()
    bias = torch.rand(2304, device=dev, generator=g)
    for impl, variant in impls:
        outs = []
        for _ in range(5):
            out = torch.empty((RM, 2304), device=dev, dtype=torch.float16)
            run(lib, torch.float16, "bias", A, W, bias, None, out, impl, variant, stream)
            outs.append(out)
        torch.cuda.synchronize()
        same = all(torch.equal(outs[0], o) for o in outs[1:])
        print(f"repeat impl={impl}/{variant}: {'bit-identical' if same else 'DIFFERS'}", flush=True)
        if impl == impls[0][0] and variant == impls[0][1]:
            first = outs[0]
        else:
            print(f"   vs impl {impls[0]}: {'bit-identical' if torch.equal(first, outs[0]) else 'differs'}", flush=True)
    if args.quick:
        return
    # ---- throughput on the ViT-B/16 shapes of a 1024-tile batch
    M = args.tiles * 197
    res = []
    for name, N, K, epi in (("qkv", 2304, 768, "bias"), ("proj", 768, 768, "bias"), ("fc1", 3072, 768, "gelu"),
                            ("fc2", 768, 3072, "bias")):
        A = (torch.rand((M, K), device=dev, generator=g) * 2 - 1).half()
        W = ((torch.rand((N, K), device=dev, generator=g) * 2 - 1) * (2.0 / K ** 0.5)).half()
        bias = torch.rand(N, device=dev, generator=g) - 0.5
        out = torch.zeros((M, N), device=dev, dtype=torch.float32 if epi == "resid" else torch.float16)
        best = {}
        for impl, variant in impls:
            for _ in range(3):
                run(lib, torch.float16, epi, A, W, bias, None, out, impl, variant, stream)
        torch.cuda.synchronize()
        for rnd in range(args.rounds):          # interleaved rounds: A B A B ... (run-to-run drift cancels)
            for impl, variant in impls:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    run(lib, torch.float16, epi, A, W, bias, None, out, impl, variant, stream)
                e1.record()
                torch.cuda.synchronize()
                best.setdefault((impl, variant), []).append(e0.elapsed_time(e1) / args.iters)
        for (impl, variant), v in best.items():
            ms = sorted(v)[len(v) // 2]
            tf = 2.0 * M * N * K / ms / 1e9
            res.append({"gemm": name, "impl": impl, "variant": variant, "ms_median": round(ms, 4), "ms_min": round(min(v), 4), "TF": round(tf, 1)})
            print(res[-1], flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
