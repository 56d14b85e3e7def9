"""The drop-in boundary from plain C ("a C-ABI shared library ... plain pointers and sizes, no torch types").

``examples/c_abi_demo.c`` drives the library with nothing but the C header and the HIP runtime's allocator.
CPU: it compiles and links with gcc -std=c99 -Wall -Wextra -Werror against the built library (the header is C, not C++).
GPU: its coordinate rows equal the coords ORACLE's for the same mask, and its features equal -- bit for bit -- what the Python host side
(HipViT) computes from the same seeded parameters and tiles: one code path, two host languages.
"""
from __future__ import annotations

import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def _build(tmp_path) -> str:
    lib = os.path.join(ROOT, "atlaspatch_amd", "libatlaspatch_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__ as g
        g.build()
    exe = str(tmp_path / "c_abi_demo")
    cmd = ["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", f"-I{ROOT}/include", f"-I{ROCM}/include", "-D__HIP_PLATFORM_AMD__",
           os.path.join(ROOT, "examples", "c_abi_demo.c"), f"-L{ROOT}/atlaspatch_amd", "-latlaspatch_hip", f"-L{ROCM}/lib", "-lamdhip64",
           f"-Wl,-rpath,{ROOT}/atlaspatch_amd", f"-Wl,-rpath,{ROCM}/lib", "-o", exe]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return exe


def test_the_c_demo_compiles_as_c99_against_the_header_and_the_library(tmp_path):
    assert os.path.getsize(_build(tmp_path)) > 0


def _lcg_stream(seed: int, count: int) -> np.ndarray:
    """x_i of x <- 1664525 x + 1013904223 (mod 2^32), i = 1 .. count, vectorised: x_i = a^i x_0 + c (1 + a + ... + a^(i-1))."""
    with np.errstate(over="ignore"):
        a = np.cumprod(np.full(count, 1664525, dtype=np.uint32), dtype=np.uint32)           # a^1 .. a^count
        geo = np.cumsum(np.concatenate([np.ones(1, np.uint32), a[:-1]]), dtype=np.uint32)    # 1 + a + ... + a^(i-1)
        return a * np.uint32(seed) + np.uint32(1013904223) * geo


@pytest.mark.gpu
def test_the_c_demo_equals_the_oracle_rows_and_the_python_host_side_bit_for_bit(tmp_path):
    import torch
    from atlaspatch_amd.encoders.vit import ARCHS, HipViT, IMAGENET_MEAN, IMAGENET_STD
    from oracle import coords_oracle

    exe, dump, n = _build(tmp_path), str(tmp_path / "feat.bin"), 5
    res = subprocess.run([exe, str(n), dump], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    lines = dict(l.split(" ", 1) for l in res.stdout.strip().splitlines())

    # ---- coordinates vs the oracle (reference utils/contours.py + services/extraction.py restated)
    yy, xx = np.mgrid[0:64, 0:64]
    mask = ((((xx - 30) ** 2 + (yy - 34) ** 2) <= 24 * 24) & ~(((xx - 36) ** 2 + (yy - 30) ** 2) <= 7 * 7)).astype(np.float32)
    want, _ = coords_oracle.coords_from_mask(mask, level0_wh=(16000, 16000), downsamples=[1.0], src_mag=20, tgt_mag=20, patch_size=256,
                                             tissue_thresh=0.0)
    want = np.asarray(want)
    f = lines["rows"].split()
    assert int(f[0]) == len(want) > 0 and (int(f[2]), int(f[3])) == (int(want[0][0]), int(want[0][1])) and f[5] == "1" and f[7] == "1"

    # ---- features vs the Python host side on the same seeded parameters and tiles
    dim, mlp, tokens, depth = 768, 3072, 197, 2
    plan = [("patch_embed.weight", dim * 3 * 16 * 16, 0.08, 0.0), ("patch_embed.bias", dim, 0.04, 0.0), ("cls_token", dim, 0.04, 0.0),
            ("pos_embed", tokens * dim, 0.04, 0.0), ("norm.weight", dim, 0.2, 1.0), ("norm.bias", dim, 0.04, 0.0)]
    for b in range(depth):
        p = f"blocks.{b}."
        plan += [(p + "ln1.weight", dim, 0.2, 1.0), (p + "ln1.bias", dim, 0.04, 0.0), (p + "qkv.weight", 3 * dim * dim, 0.08, 0.0),
                 (p + "qkv.bias", 3 * dim, 0.04, 0.0), (p + "proj.weight", dim * dim, 0.08, 0.0), (p + "proj.bias", dim, 0.04, 0.0),
                 (p + "ln2.weight", dim, 0.2, 1.0), (p + "ln2.bias", dim, 0.04, 0.0), (p + "fc1.weight", mlp * dim, 0.08, 0.0),
                 (p + "fc1.bias", mlp, 0.04, 0.0), (p + "fc2.weight", dim * mlp, 0.08, 0.0), (p + "fc2.bias", dim, 0.04, 0.0)]
    stream = _lcg_stream(12345, sum(c for _, c, _, _ in plan))
    unit = (stream >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0) - np.float32(0.5)
    shapes = {"patch_embed.weight": (dim, 3, 16, 16), "pos_embed": (tokens, dim)}
    state, at = {}, 0
    for name, count, scale, offset in plan:
        vals = np.float32(offset) + np.float32(scale) * unit[at:at + count]
        at += count
        base = name.split(".")[-2] + "." + name.split(".")[-1] if name.startswith("blocks.") else name
        shape = shapes.get(name) or ((3 * dim, dim) if base == "qkv.weight" else (dim, dim) if base == "proj.weight" else
                                     (mlp, dim) if base == "fc1.weight" else (dim, mlp) if base == "fc2.weight" else (count,))
        state[name] = torch.from_numpy(vals.reshape(shape).copy())
    tiles = (_lcg_stream(777, n * 256 * 256 * 3) >> np.uint32(24)).astype(np.uint8).reshape(n, 256, 256, 3)
    dev = torch.device("cuda:0")
    vit = HipViT(dict(ARCHS["vit_b_16"], depth=depth), state, device=dev, dtype=torch.float16)
    out = torch.empty((n, dim), dtype=torch.float32, device=dev)
    vit.forward_u8(torch.from_numpy(tiles).to(dev), IMAGENET_MEAN, IMAGENET_STD, out)
    torch.cuda.synchronize()
    py = out.cpu().numpy()
    vit.release()
    c = np.fromfile(dump, dtype=np.float32).reshape(n, dim)
    assert np.isfinite(c).all() and np.abs(c).max() > 0.1
    assert np.array_equal(c, py)
    ff = lines["feat"].split()
    assert (int(ff[0]), int(ff[1])) == (n, dim) and float.fromhex(ff[3]) == float(c[0, 0]) and float.fromhex(ff[4]) == float(c[-1, -1])
