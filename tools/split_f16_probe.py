#!/usr/bin/env python3
"""The float32-accurate fast product (ap_gemm impl 129, AP_VIT_OPT_SPLIT_F16) on one MI355X.

    python tools/split_f16_probe.py [--tiles 2048] [--skip-encoder]

1. GEMM accuracy against a float64 matmul of the same float32 operands: exact f32 MFMA (impl 128) vs split-f16
   (impl 129) on the ViT-B shapes, with operands that include tiny and large magnitudes.
2. GEMM time of both on the 2048-tile shapes.
3. The depth-12 ViT-B/16: features of float32 / split_f16 against the fp32 CPU oracle (three statistics) and tiles/s."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from atlaspatch_amd import _lib


def gemm(lib, A, W, bias, out, impl, epi=0):
    M, K = A.shape
    N = out.shape[1]
    _lib.check(lib.ap_gemm(_lib.AP_F32, epi, A.data_ptr(), A.stride(0), W.data_ptr(), K, M, N, K, bias.data_ptr(), None,
                           out.data_ptr(), out.stride(0), impl, 0, _lib.current_stream_ptr(A.device)), "ap_gemm")


def split_w(lib, W):
    out = torch.empty_like(W)
    _lib.check(lib.ap_split_f16_weights(W.data_ptr(), out.data_ptr(), W.numel(), _lib.current_stream_ptr(W.device)), "ap_split_f16_weights")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=2048)
    ap.add_argument("--skip-encoder", action="store_true")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(0)
    res = {"gemm": [], "time": []}
    # ---- 1. accuracy
    for name, (M, N, K), ascale, wscale in [("qkv-like", (3941, 2304, 768), 1.0, 0.03), ("fc2-like small acts", (3941, 768, 3072), 0.02, 0.02),
                                            ("tiny weights", (1000, 768, 768), 1.0, 1e-5), ("large acts", (1000, 768, 768), 300.0, 0.03),
                                            ("mixed magnitudes", (2000, 1024, 1024), None, 0.03)]:
        A = torch.randn((M, K), device=dev, generator=g)
        A = A * ascale if ascale is not None else A * torch.exp2(torch.randint(-12, 6, (M, K), device=dev, generator=g).float())
        W = torch.randn((N, K), device=dev, generator=g) * wscale
        bias = torch.randn(N, device=dev, generator=g) * 0.1
        ref = (A.double() @ W.double().t() + bias.double())
        row = {"case": name, "M": M, "N": N, "K": K}
        Ws = split_w(lib, W)
        for tag, impl, w in (("f32", 128, W), ("split", 129, Ws)):
            out = torch.full((M, N), float("nan"), device=dev)
            gemm(lib, A, w, bias, out, impl)
            torch.cuda.synchronize()
            d = (out.double() - ref)
            row[tag + "_norm"] = float(d.norm() / ref.norm())
            row[tag + "_max_rel_to_rms"] = float(d.abs().max() / ref.pow(2).mean().sqrt())
        res["gemm"].append(row)
        print("ACC", json.dumps(row), flush=True)
    # ---- 2. time
    Mb = args.tiles * 197
    for name, (N, K), epi in [("qkv", (2304, 768), 0), ("proj", (768, 768), 0), ("fc1+gelu", (3072, 768), 1), ("fc2", (768, 3072), 0)]:
        A = torch.randn((Mb, K), device=dev, generator=g)
        W = torch.randn((N, K), device=dev, generator=g) * 0.03
        Ws = split_w(lib, W)
        bias = torch.randn(N, device=dev, generator=g) * 0.1
        out = torch.empty((Mb, N), device=dev)
        row = {"gemm": name, "M": Mb, "N": N, "K": K}
        for tag, impl, w in (("f32", 128, W), ("split", 129, Ws)):
            for _ in range(2):
                gemm(lib, A, w, bias, out, impl, epi)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                gemm(lib, A, w, bias, out, impl, epi)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            row[tag + "_ms"] = ms
            row[tag + "_tflops"] = 2.0 * Mb * N * K / ms / 1e9
        res["time"].append(row)
        print("TIME", json.dumps(row), flush=True)
        del A, out
    # ---- 3. encoder
    if not args.skip_encoder:
        from atlaspatch_amd.encoders.vit import build_hip_vit_extractor
        from oracle import vit_oracle
        sd = dict(vit_oracle.make_hf_vit(layers=12).state_dict())
        rng = np.random.default_rng(11)
        tiles = [rng.integers(0, 256, (256, 256, 3), dtype=np.uint8) for _ in range(19)]
        want = vit_oracle.extract_batch(sd, tiles, heads=12, batch_size=32).astype(np.float64)
        ex = build_hip_vit_extractor(name="hfvit_L12", arch="vit_b_16", depth=12, state_dict=sd, device=dev, dtype=torch.float32, source="hf")
        big = torch.randint(0, 256, (args.tiles, 256, 256, 3), dtype=torch.uint8, device=dev)
        out = torch.empty((args.tiles, 768), device=dev)
        mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
        enc = []
        for mode in (False, True):
            ex.vit.set_option("split_f16", mode)
            got = ex.extract_batch(tiles, batch_size=32).astype(np.float64)
            d = np.abs(got - want)
            el = d / (np.abs(want) + 0.05 * np.abs(want).max())
            ex.vit.forward_u8(big, mean, std, out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                ex.vit.forward_u8(big, mean, std, out)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            ex.vit.profile(True)
            ex.vit.forward_u8(big, mean, std, out)
            torch.cuda.synchronize()
            prof = ex.vit.profile_read()
            ex.vit.profile(False)
            row = {"split_f16": mode, "norm": float(np.linalg.norm(got - want) / np.linalg.norm(want)), "elem_max": float(el.max()),
                   "elem_q999": float(np.quantile(el, 0.999)), "tiles_per_s": args.tiles / dt, "ms_per_step": dt * 1e3,
                   "ms_by_kind": {k: round(v[0], 2) for k, v in prof.items()}}
            enc.append(row)
            print("ENC", json.dumps(row), flush=True)
        res["encoder"] = enc
        ex.cleanup()
    if args.json:
        with open(args.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
