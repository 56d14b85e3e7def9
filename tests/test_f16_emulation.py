"""CPU: the mechanism behind the exact class rows, checked without a GPU.

tools/f16_error_attribution.py restates every rounding of the float16 device dataflow in torch.  On the seeded depth-12 ViT-B/16 of
the parity tests the emulation of the PLAIN 16-bit stream lands on the error the device measures (1.26e-3 norm-wise against the
all-float32 forward; tests/test_gpu_parity.py bounds the device at 1.256e-3 x 1.2), and keeping only the class rows' stream and
branch exact removes more than a third of it -- the patch rows' stream roundings reach the features only through attention."""
import numpy as np
import torch


def test_the_class_rows_own_roundings_carry_most_of_the_16_bit_streams_error():
    import math
    import torch.nn.functional as F
    from oracle import vit_oracle as vo
    torch.set_num_threads(min(8, torch.get_num_threads()))
    sd = {k: v.detach().float() for k, v in vo.make_hf_vit().state_dict().items()}
    x = vo.preprocess_center_crop(np.random.default_rng(0).integers(0, 256, (2, 256, 256, 3), dtype=np.uint8))
    H = lambda t: t.half().float()

    @torch.inference_mode()
    def fwd(dev, cls_exact):
        R = H if dev else (lambda t: t)
        w = R(sd["embeddings.patch_embeddings.projection.weight"])
        d, heads, n = w.shape[0], 12, x.shape[0]
        dh = d // heads
        pe = F.conv2d(R(x), w, sd["embeddings.patch_embeddings.projection.bias"], stride=16).flatten(2).transpose(1, 2)
        tok = torch.cat([sd["embeddings.cls_token"].expand(n, -1, -1), pe], dim=1) + sd["embeddings.position_embeddings"]

        def upd(t, branch):                      # the proj / fc2 epilogue: T(stream + T(branch)); class rows in f32 when exact
            if not dev:
                return t + branch
            br = R(branch)
            new = t + br
            r = R(new)
            if cls_exact:
                r[:, 0] = t[:, 0] + branch[:, 0]
            return r
        if dev:
            t0 = R(tok)
            if cls_exact:
                t0[:, 0] = tok[:, 0]
            tok = t0
        layer = 0
        while True:
            names = vo._layer_keys(sd, layer)
            if names is None:
                break
            p, q_, k_, v_, o_, f1, f2 = names

            def ln_gemm(xin, lnw, lnb, wk, bk):  # LayerNorm folded into the GEMM: A = the rounded stream, W' = T(W * gamma)
                xa = R(xin)
                wf = R(sd[wk] * sd[lnw])
                mu = xa.mean(-1, keepdim=True)
                rstd = (xa.var(-1, unbiased=False, keepdim=True) + 1e-6).rsqrt()
                return ((xa - mu) * rstd) @ wf.T + (sd[bk] + sd[wk] @ sd[lnb])
            q, k, v = [R(ln_gemm(tok, p + "layernorm_before.weight", p + "layernorm_before.bias", nm + ".weight", nm + ".bias"))
                       .view(n, -1, heads, dh).transpose(1, 2) for nm in (q_, k_, v_)]
            s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
            pexp = torch.exp(s - s.max(-1, keepdim=True).values)
            ctx = R(((R(pexp) @ v) / pexp.sum(-1, keepdim=True)).transpose(1, 2).reshape(n, -1, d))
            tok = upd(tok, ctx @ R(sd[o_ + ".weight"]).T + sd[o_ + ".bias"])
            hid = R(F.gelu(ln_gemm(tok, p + "layernorm_after.weight", p + "layernorm_after.bias", f1 + ".weight", f1 + ".bias")))
            tok = upd(tok, hid @ R(sd[f2 + ".weight"]).T + sd[f2 + ".bias"])
            layer += 1
        return F.layer_norm(tok, (d,), sd["layernorm.weight"], sd["layernorm.bias"], 1e-6)[:, 0]

    ref = fwd(False, False)
    rel = lambda o: float((o - ref).norm() / ref.norm())
    plain, exact = rel(fwd(True, False)), rel(fwd(True, True))
    print(f"emulated float16 dataflow vs float32: plain 16-bit stream {plain:.3e}, class rows exact {exact:.3e}")
    assert 1.0e-3 <= plain <= 1.5e-3            # the device measures 1.256e-3 (19 tiles) / 1.260e-3 (G1) on this model
    assert exact <= 0.72 * plain and exact <= 1e-3
