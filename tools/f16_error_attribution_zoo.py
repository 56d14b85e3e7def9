#!/usr/bin/env python3
"""CPU emulation of the float16 device dataflow for ANY canonical ViT of the zoo (tools/f16_error_attribution.py is the ViT-B/16
original): every rounding the HIP path performs -- folded weights (LayerNorm gain into qkv / fc1, LayerScale into proj / fc2, ONE
rounding of the product), the 16-bit residual stream, the class row as a GEMM operand, qkv, softmax weights, attention output,
the MLP's hidden activation (GELU or the SwiGLU gate), branch outputs -- restated in torch on seeded random weights, against the
all-float32 forward.  Prints the norm-wise error with the class rows exact (the build's default), each source switched off alone /
on alone, and what an exact class row for further quantities would return.

    python tools/f16_error_attribution_zoo.py uni_v2 [--images 2] [--depth N] [--ls 1e-5] [--wscale 0.02]

Round 5 used it on uni_v2 (device: 2.05e-3 norm-wise against the fp32 oracle in float16, bound 1e-3 for ViT-B/16)."""
import argparse, math, os, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atlaspatch_amd.encoders.vit import ARCHS, random_canonical_state_dict

ap = argparse.ArgumentParser()
ap.add_argument("arch")
ap.add_argument("--images", type=int, default=2)
ap.add_argument("--depth", type=int, default=None)
ap.add_argument("--seed", type=int, default=21)
ap.add_argument("--ls", type=float, default=None, help="override every LayerScale entry (the tests draw 1e-5 +- 1e-6: timm's init_values)")
ap.add_argument("--ls-test", action="store_true", help="LayerScale drawn in [0.2, 0.7] with seed 42, weights seed 41: the GPU parity test's model "
                                                     "(tests/test_encoder_zoo.py::test_encoder_at_real_size_vs_fp32_oracle)")
ap.add_argument("--only", default="", help="comma list: run only these single-source lines")
args = ap.parse_args()
torch.set_num_threads(8)
arch = dict(ARCHS[args.arch])
if args.depth:
    arch["depth"] = args.depth
sd = random_canonical_state_dict(arch, seed=41 if args.ls_test else args.seed)
if args.ls_test and arch.get("layer_scale"):
    g42 = torch.Generator().manual_seed(42)
    for i in range(arch["depth"]):
        sd[f"blocks.{i}.ls1"] = torch.rand(arch["dim"], generator=g42) * 0.5 + 0.2
        sd[f"blocks.{i}.ls2"] = torch.rand(arch["dim"], generator=g42) * 0.5 + 0.2
if args.ls is not None:
    for k in sd:
        if k.endswith(".ls1") or k.endswith(".ls2"):
            sd[k] = torch.full_like(sd[k], args.ls)
d, heads, depth, eps = arch["dim"], arch["heads"], arch["depth"], float(arch["ln_eps"])
dh = d // heads
swiglu = arch.get("mlp") == "swiglu"
n = args.images
x = torch.randn(n, 3, arch["image_size"], arch["image_size"], generator=torch.Generator().manual_seed(5))
H = lambda t: t.half().float()
ALL = ("w", "xin", "a_in", "qkv", "p", "ctx", "hid", "branch", "stream")


@torch.inference_mode()
def fwd(on, cls_only_exact=(), exact_cls=True):
    def R(t, src, rowdim=None):
        if src not in on:
            return t
        r = H(t)
        if src in cls_only_exact and rowdim is not None:
            idx = [slice(None)] * t.dim(); idx[rowdim] = 0
            r[tuple(idx)] = t[tuple(idx)]
        return r
    w, b = R(sd["patch_embed.weight"], "w"), sd["patch_embed.bias"]
    pe = F.conv2d(R(x, "xin"), w, b, stride=w.shape[-1]).flatten(2).transpose(1, 2)
    prefix = [sd["cls_token"].view(1, 1, d).expand(n, -1, -1)]
    if "reg_tokens" in sd:
        prefix.append(sd["reg_tokens"].view(1, -1, d).expand(n, -1, -1))
    pos = sd["pos_embed"]
    tok = torch.cat(prefix + [pe + pos[None]], 1) if pos.shape[0] == pe.shape[1] else torch.cat(prefix + [pe], 1) + pos[None]

    def settle(new):                       # the 16-bit stream; class rows exact (cls32) when the option is on
        if "stream" not in on:
            return new
        r = H(new)
        if exact_cls:
            r[:, 0] = new[:, 0]
        return r
    tok = settle(tok)

    def ln_gemm(xin, lnw, lnb, wk, bk):    # LayerNorm folded into the GEMM: W' = T(W * g), statistics of the operand in f32
        xa = R(xin, "a_in", 1)
        Wf = R(sd[wk] * sd[lnw], "w")
        mu = xa.mean(-1, keepdim=True); var = xa.var(-1, unbiased=False, keepdim=True); rstd = (var + eps).rsqrt()
        return ((xa - mu) * rstd) @ Wf.T + (sd[bk] + sd[wk] @ sd[lnb])

    def ls_gemm(a, wk, bk, lsk):           # LayerScale folded: W' = T(W * ls[:, None]), bias' = bias * ls
        ls = sd[lsk] if lsk in sd else None
        Wf = R(sd[wk] * ls[:, None] if ls is not None else sd[wk], "w")
        return a @ Wf.T + (sd[bk] * ls if ls is not None else sd[bk])
    for i in range(depth):
        p = f"blocks.{i}."
        qkv = R(ln_gemm(tok, p + "ln1.weight", p + "ln1.bias", p + "qkv.weight", p + "qkv.bias"), "qkv", 1)
        t = tok.shape[1]
        q, k, v = qkv.view(n, t, 3, heads, dh).permute(2, 0, 3, 1, 4)
        s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
        pexp = torch.exp(s - s.max(-1, keepdim=True).values)
        ctx = (R(pexp, "p", 2) @ v) / pexp.sum(-1, keepdim=True)
        ctx = R(ctx.transpose(1, 2).reshape(n, t, d), "ctx", 1)
        tok = settle(tok + R(ls_gemm(ctx, p + "proj.weight", p + "proj.bias", p + "ls1"), "branch", 1))
        m = ln_gemm(tok, p + "ln2.weight", p + "ln2.bias", p + "fc1.weight", p + "fc1.bias")
        if swiglu:
            x1, x2 = m.chunk(2, -1)
            m = F.silu(x1) * x2
        else:
            m = F.gelu(m)
        m = R(m, "hid", 1)
        tok = settle(tok + R(ls_gemm(m, p + "fc2.weight", p + "fc2.bias", p + "ls2"), "branch", 1))
    return F.layer_norm(tok, (d,), sd["norm.weight"], sd["norm.bias"], eps)[:, 0]


t0 = time.time()
ref = fwd(set())
print(f"{args.arch}: dim {d}, depth {depth}, {n} images, one forward {time.time() - t0:.1f} s", flush=True)
rel = lambda o: float((o - ref).norm() / ref.norm())
E = ("branch",)
print(f"all roundings, plain 16-bit stream (exact_cls off): {rel(fwd(set(ALL), exact_cls=False)):.3e}", flush=True)
print(f"all roundings, class stream + class branch exact (the default): {rel(fwd(set(ALL), cls_only_exact=E)):.3e}", flush=True)
only = [s for s in args.only.split(",") if s]
for src in (only or ALL):
    a = rel(fwd(set(ALL) - {src}, cls_only_exact=E)); b = rel(fwd({src}, cls_only_exact=E))
    print(f"  without '{src}': {a:.3e}    only '{src}': {b:.3e}", flush=True)
if not only:
    for src in ("a_in", "qkv", "ctx", "hid", "p"):
        print(f"  class row exact for '{src}' too: {rel(fwd(set(ALL), cls_only_exact=E + (src,))):.3e}", flush=True)
