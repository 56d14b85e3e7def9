"""SAM2.1 Hiera-T image path on the HIP float32 operator set (reference: services/segmentation.py:25-180).

``Sam2HipPredictor`` mirrors ``_SAM2Predictor``: ``predict_image(thumbnail)`` = PIL BILINEAR resize to 1024 x 1024,
``set_image`` + ``predict(box=[0, 0, w, h], multimask_output=False, return_logits=False)``, PIL NEAREST resize of the
mask back to the thumbnail.  The network itself (Hiera trunk, FpnNeck, prompt encoder, two-way mask decoder, see
oracle/sam2_oracle.py for the structure and its sources) runs as a chain of the C-ABI kernels in
``csrc/sam2_ops.hip`` plus ``ap_layernorm``: torch only owns the device buffers.  Everything that depends on the
weights alone is folded on the host once: positional embedding (bicubic background + tiled window), the box-prompt
tokens of the constant box, the dense positional encoding, ``no_mask_embed`` / ``no_mem_embed`` sums.

There is no CPU fallback: without the library or a HIP device construction raises.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional

import numpy as np
import torch

from .. import _lib

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)
STAGES = (1, 2, 7, 2)
EMBED = 96
WINDOW_SPEC = (8, 4, 14, 7)
GLOBAL_BLOCKS = (5, 7, 9)


def block_plan():
    """[(dim_in, dim_out, heads, window, q_pool)] per block + stage ends (sam2 hieradet.py Hiera.__init__)."""
    stage_ends = [sum(STAGES[:i]) - 1 for i in range(1, len(STAGES) + 1)]
    q_pool_blocks = [x + 1 for x in stage_ends[:-1]][:3]
    plan, dim, heads, cur = [], EMBED, 1, 1
    for i in range(sum(STAGES)):
        dim_out, window = dim, WINDOW_SPEC[cur - 1]              # the window size lags by one block
        if i in GLOBAL_BLOCKS:
            window = 0
        if i - 1 in stage_ends:
            dim_out, heads, cur = dim * 2, heads * 2, cur + 1
        plan.append((dim, dim_out, heads, window, i in q_pool_blocks))
        dim = dim_out
    return plan, stage_ends


class Sam2HipPredictor:
    MAX_BATCH = 32          # thumbnails per forward (larger --seg-batch-size values run as several forwards)

    def __init__(self, state_dict: dict, *, device="cuda", mask_threshold: float = 0.0) -> None:
        self.device = torch.device(device)
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise _lib.HipLibraryError("the SAM2 segmenter needs a HIP device ('cuda' on PyTorch-ROCm); there is no CPU fallback")
        self.lib = _lib.load()
        self.mask_threshold = float(mask_threshold)
        self.input_size = 1024
        self.plan, self.stage_ends = block_plan()
        self._graph = None
        self._flop = None                    # set to 0.0 by count_flop(): _gemm / _attention then add their multiply-adds
        self._batch_graphs: dict = {}
        self.fused_attention = os.environ.get("ATLASPATCH_SAM2_UNFUSED_ATTENTION") in (None, "", "0")
        self.wide_gemm = os.environ.get("ATLASPATCH_SAM2_NO_WIDE_GEMM") in (None, "", "0")      # A/B switch of _gemm's kernel choice
        # row-wise layers as split-f16 products (ap_gemm_split_f16: three f16 MFMA passes on hi / lo halves, f32 accumulation;
        # float32-accurate -- measured closer to float64 than the exact f32 MFMA chain on every layer shape -- at 1.3-3x its
        # rate); ATLASPATCH_SAM2_EXACT_F32=1 keeps the exact chain everywhere (rounds 2-5's arithmetic)
        self.split_gemm = os.environ.get("ATLASPATCH_SAM2_EXACT_F32") in (None, "", "0")
        self._split: dict = {}               # data_ptr of a weight matrix -> its [hi | lo] f16 rows (ap_split_f16_weights)
        self._static_img = self._static_mask = None
        self._resamplers: dict = {}
        f = lambda t: t.detach().to(torch.float32).contiguous()
        sd = {k: f(v) for k, v in state_dict.items()}
        dev = lambda t: t.to(self.device).contiguous()
        self.w = {}
        t = "image_encoder.trunk."
        self.w["pe.w"] = dev(sd[t + "patch_embed.proj.weight"].reshape(EMBED, 147))
        self.w["pe.b"] = dev(sd[t + "patch_embed.proj.bias"])
        # positional embedding: weights only -> folded once (hieradet.py _get_pos_embed)
        pos = torch.nn.functional.interpolate(sd[t + "pos_embed"], size=(256, 256), mode="bicubic")
        win = sd[t + "pos_embed_window"]
        pos = pos + win.tile([x // y for x, y in zip(pos.shape, win.shape)])
        self.w["pos"] = dev(pos.permute(0, 2, 3, 1).reshape(256 * 256, EMBED))
        for i, (din, dout, heads, window, qpool) in enumerate(self.plan):
            b = f"{t}blocks.{i}."
            for name in ("norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight",
                         "attn.proj.bias", "norm2.weight", "norm2.bias", "mlp.layers.0.weight", "mlp.layers.0.bias",
                         "mlp.layers.1.weight", "mlp.layers.1.bias"):
                self.w[f"b{i}.{name}"] = dev(sd[b + name])
            if din != dout:
                self.w[f"b{i}.proj.weight"] = dev(sd[b + "proj.weight"]); self.w[f"b{i}.proj.bias"] = dev(sd[b + "proj.bias"])
        for n in range(4):
            self.w[f"neck{n}.w"] = dev(sd[f"image_encoder.neck.convs.{n}.conv.weight"].reshape(256, -1))
            self.w[f"neck{n}.b"] = dev(sd[f"image_encoder.neck.convs.{n}.conv.bias"])
        d = "sam_mask_decoder."
        self.w["s0.w"] = dev(sd[d + "conv_s0.weight"].reshape(32, 256)); self.w["s0.b"] = dev(sd[d + "conv_s0.bias"])
        self.w["s1.w"] = dev(sd[d + "conv_s1.weight"].reshape(64, 256)); self.w["s1.b"] = dev(sd[d + "conv_s1.bias"])
        # image embedding gets no_mem_embed (directly_add_no_mem_embed) and, inside the decoder, no_mask_embed: one vector
        self.w["embed_add"] = dev(sd["no_mem_embed"].reshape(256) + sd["sam_prompt_encoder.no_mask_embed.weight"].reshape(256))
        sparse, dense_pe = self._prompt_constants(sd)
        tokens = torch.cat([sd[d + "obj_score_token.weight"], sd[d + "iou_token.weight"], sd[d + "mask_tokens.weight"], sparse], 0)
        self.w["tokens"] = dev(tokens)                       # [9, 256]
        self.w["image_pe"] = dev(dense_pe)                   # [4096, 256]
        for l in range(2):
            b = f"{d}transformer.layers.{l}."
            for a in ("self_attn", "cross_attn_token_to_image", "cross_attn_image_to_token"):
                for pj in ("q_proj", "k_proj", "v_proj", "out_proj"):
                    self.w[f"d{l}.{a}.{pj}.w"] = dev(sd[b + f"{a}.{pj}.weight"]); self.w[f"d{l}.{a}.{pj}.b"] = dev(sd[b + f"{a}.{pj}.bias"])
            for k in range(1, 5):
                self.w[f"d{l}.norm{k}.w"] = dev(sd[b + f"norm{k}.weight"]); self.w[f"d{l}.norm{k}.b"] = dev(sd[b + f"norm{k}.bias"])
            for k in range(2):
                self.w[f"d{l}.mlp{k}.w"] = dev(sd[b + f"mlp.layers.{k}.weight"]); self.w[f"d{l}.mlp{k}.b"] = dev(sd[b + f"mlp.layers.{k}.bias"])
        for pj in ("q_proj", "k_proj", "v_proj", "out_proj"):
            self.w[f"dfin.{pj}.w"] = dev(sd[d + f"transformer.final_attn_token_to_image.{pj}.weight"])
            self.w[f"dfin.{pj}.b"] = dev(sd[d + f"transformer.final_attn_token_to_image.{pj}.bias"])
        self.w["dfin.norm.w"] = dev(sd[d + "transformer.norm_final_attn.weight"]); self.w["dfin.norm.b"] = dev(sd[d + "transformer.norm_final_attn.bias"])
        # ConvTranspose2d weight [Cin, Cout, 2, 2] -> GEMM weight [Cout * 4, Cin] (row co * 4 + dy * 2 + dx)
        self.w["up0.w"] = dev(sd[d + "output_upscaling.0.weight"].permute(1, 2, 3, 0).reshape(64 * 4, 256))
        self.w["up0.b"] = dev(sd[d + "output_upscaling.0.bias"])
        self.w["up1.w"] = dev(sd[d + "output_upscaling.1.weight"]); self.w["up1.b"] = dev(sd[d + "output_upscaling.1.bias"])
        self.w["up3.w"] = dev(sd[d + "output_upscaling.3.weight"].permute(1, 2, 3, 0).reshape(32 * 4, 64))
        self.w["up3.b"] = dev(sd[d + "output_upscaling.3.bias"])
        for k in range(3):
            self.w[f"hyper{k}.w"] = dev(sd[d + f"output_hypernetworks_mlps.0.layers.{k}.weight"])
            self.w[f"hyper{k}.b"] = dev(sd[d + f"output_hypernetworks_mlps.0.layers.{k}.bias"])

        if self.split_gemm:
            with torch.cuda.device(self.device):
                for name, t in self.w.items():
                    if t.dim() == 2 and t.shape[0] % 32 == 0 and t.shape[1] % 32 == 0 and t.data_ptr() % 16 == 0:
                        out = torch.empty_like(t)
                        _lib.check(self.lib.ap_split_f16_weights(t.data_ptr(), out.data_ptr(), t.numel(), self._stream()),
                                   "ap_split_f16_weights")
                        self._split[t.data_ptr()] = out
                torch.cuda.synchronize(self.device)

    # ------------------------------------------------------------------ constants of the box prompt
    @staticmethod
    def _prompt_constants(sd: dict, size: int = 1024):
        p = "sam_prompt_encoder."
        g = sd[p + "pe_layer.positional_encoding_gaussian_matrix"]

        def pe(coords01):
            c = 2 * math.pi * ((2 * coords01 - 1) @ g)
            return torch.cat([torch.sin(c), torch.cos(c)], dim=-1)

        # box [0, 0, size, size] -> corners + 0.5 (labels 2, 3) and one padding point (label -1, encoding zeroed)
        pts = torch.tensor([[0.5, 0.5], [size + 0.5, size + 0.5], [0.0, 0.0]])
        emb = pe(pts / float(size))
        emb[2] = sd[p + "not_a_point_embed.weight"][0]
        emb[0] += sd[p + "point_embeddings.2.weight"][0]
        emb[1] += sd[p + "point_embeddings.3.weight"][0]
        grid = (torch.arange(64, dtype=torch.float32) + 0.5) / 64
        yy, xx = torch.meshgrid(grid, grid, indexing="ij")
        return emb, pe(torch.stack([xx, yy], dim=-1)).reshape(4096, 256)

    # ------------------------------------------------------------------ thin op wrappers (device pointers only)
    def _buf(self, *shape) -> torch.Tensor:
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    def _stream(self):
        return _lib.current_stream_ptr(self.device)

    def _gemm(self, a, w, n, k, *, bias=None, act=0, resid=None, out=None, m=None, lda=None, ldw=None, ldo=None, ldr=None,
              batch=1, sa=0, sw=0, so=0, sr=0, w_kn=False, alpha=1.0, stack=1):
        m = a.shape[0] if m is None else m
        out = self._buf(m, n) if out is None else out
        if self._flop is not None:
            self._flop += 2.0 * batch * m * n * k
        # Row-wise layers with >= 90 tiles of 128 x 128 per image go to the 128 x 128-tile kernel of the encoder path: as split-f16
        # products (default; tools/sgemm_vs_split_probe.py: 0.33-0.79 x ap_sgemm's time on the shapes this selects, 0.95-1.9 x on
        # the ones it does not -- profiles/r06d_sgemm_vs_split.txt), or -- ATLASPATCH_SAM2_EXACT_F32=1 -- the five widest on its
        # exact f32 form (round 5).  The choice looks at ONE image's rows, never at the batch, and neither kernel splits K: a
        # row's result does not depend on what it is stacked with.
        plain = batch == 1 and not w_kn and alpha == 1.0 and act in (0, 1) and (lda is None or lda == k) and (ldw is None or ldw == k)
        tiles = -(-(m // stack) // 128) * -(-n // 128)
        if (self.wide_gemm and self.split_gemm and plain and tiles >= 90 and n % 32 == 0 and k % 32 == 0 and w.data_ptr() in self._split
                and a.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0 and (ldo is None or ldo % 4 == 0)
                and (resid is None or (resid.data_ptr() % 16 == 0 and (ldr is None or ldr % 4 == 0)))):
            _lib.check(self.lib.ap_gemm_split_f16(a.data_ptr(), k, self._split[w.data_ptr()].data_ptr(), m, n, k,
                                                  bias.data_ptr() if bias is not None else None, act,
                                                  resid.data_ptr() if resid is not None else None, ldr if ldr is not None else n,
                                                  out.data_ptr(), ldo if ldo is not None else n, self._stream()), "ap_gemm_split_f16")
            return out
        if (self.wide_gemm and not self.split_gemm and plain and resid is None and bias is not None
                and n % 128 == 0 and k % 32 == 0 and k <= 768 and -(-(m // stack) // 128) * (n // 128) >= 512
                and (ldo is None or ldo == n) and a.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0):
            _lib.check(self.lib.ap_gemm(_lib.AP_F32, act, a.data_ptr(), k, w.data_ptr(), k, m, n, k, bias.data_ptr(), None,
                                        out.data_ptr(), n, 128, 0, self._stream()), "ap_gemm")
            return out
        if stack > 1:
            # `stack` images through one row-wise layer: planned like a single image (ap_sgemm_stacked), so a row's result
            # does not depend on the batch size
            assert batch == 1 and not w_kn and alpha == 1.0
            _lib.check(self.lib.ap_sgemm_stacked(a.data_ptr(), lda if lda is not None else k, w.data_ptr(),
                                                 ldw if ldw is not None else k, stack, m, n, k,
                                                 bias.data_ptr() if bias is not None else None, act,
                                                 resid.data_ptr() if resid is not None else None, ldr if ldr is not None else n,
                                                 out.data_ptr(), ldo if ldo is not None else n, self._stream()), "ap_sgemm_stacked")
            return out
        _lib.check(self.lib.ap_sgemm(a.data_ptr(), lda if lda is not None else k, sa, w.data_ptr(),
                                     ldw if ldw is not None else (n if w_kn else k), sw, 1 if w_kn else 0, batch, m, n, k,
                                     C.c_float(alpha), bias.data_ptr() if bias is not None else None, act,
                                     resid.data_ptr() if resid is not None else None, ldr if ldr is not None else n, sr,
                                     out.data_ptr(), ldo if ldo is not None else n, so, self._stream()), "ap_sgemm")
        return out

    def _gemm_windows(self, a, w, n, k, *, mode, B, H, W, window, bias, resid=None):
        """A windowed block's qkv (mode 1: `a` in image order -> window-order rows) or proj + residual (mode 2: window-order
        `a` -> image-order rows added to `resid`) with the (un)partition folded into the split-f16 GEMM; None when that kernel
        does not take the layer (the caller then partitions / un-partitions with the stand-alone passes)."""
        nwy, nwx = -(-H // window), -(-W // window)
        m = B * nwy * nwx * window * window
        tiles = -(-(m // B) // 128) * -(-n // 128)
        if not (self.wide_gemm and self.split_gemm and tiles >= 90 and n % 32 == 0 and k % 32 == 0 and w.data_ptr() in self._split
                and a.data_ptr() % 16 == 0):
            return None
        out = self._buf(m if mode == 1 else B * H * W, n)
        if self._flop is not None:
            self._flop += 2.0 * m * n * k
        _lib.check(self.lib.ap_gemm_split_f16_windows(a.data_ptr(), k, self._split[w.data_ptr()].data_ptr(), m, n, k, bias.data_ptr(), 0,
                                                      resid.data_ptr() if resid is not None else None, n, out.data_ptr(), n,
                                                      mode, B, H, W, window, self._stream()), "ap_gemm_split_f16_windows")
        return out

    def _ln(self, x, rows, dim, w, b, eps, out=None):
        out = self._buf(rows, dim) if out is None else out
        _lib.check(self.lib.ap_layernorm(_lib.AP_F32, x.data_ptr(), dim, rows, dim, w.data_ptr(), b.data_ptr(), C.c_float(eps),
                                         out.data_ptr(), self._stream()), "ap_layernorm")
        return out

    def _attention(self, q, k, v, *, nb, heads, tq, tk, d, ldq, ldk, ldv):
        """q [nb*tq, ldq], k / v [nb*tk, ld*] with head h at column h*d  ->  [nb*tq, heads*d]."""
        out = self._buf(nb * tq, heads * d)
        scale = 1.0 / math.sqrt(d)
        if self._flop is not None:
            self._flop += 4.0 * nb * heads * tq * tk * d
        if (self.fused_attention and d in (32, 64, 96) and nb <= 65535 and ldq % 4 == 0 and ldk % 4 == 0
                and q.data_ptr() % 16 == 0 and k.data_ptr() % 16 == 0):
            # one fused kernel for all windows and heads: the [nb, heads, tq, tk] score matrix never reaches memory
            fn = self.lib.ap_sattention_split_f16 if self.split_gemm else self.lib.ap_sattention_f32
            _lib.check(fn(q.data_ptr(), ldq, k.data_ptr(), ldk, v.data_ptr(), ldv, nb, heads, tq, tk, d,
                          scale, out.data_ptr(), heads * d, self._stream()), "ap_sattention")
            return out
        scores = self._buf(nb * heads, tq, tk)
        if nb == 1:                     # one image-wide attention: the heads are the batch (head h = column offset h * d)
            self._gemm(q, k, tk, d, m=tq, lda=ldq, ldw=ldk, out=scores, ldo=tk, batch=heads, sa=d, sw=d, so=tq * tk, alpha=scale)
            _lib.check(self.lib.ap_softmax_rows(scores.data_ptr(), tk, heads * tq, tk, self._stream()), "ap_softmax_rows")
            self._gemm(scores, v, d, tk, m=tq, lda=tk, ldw=ldv, out=out, ldo=heads * d, batch=heads, sa=tq * tk, sw=d, so=d,
                       w_kn=True)
            return out
        for h in range(heads):          # batched over windows; heads are column offsets
            qh, kh, vh = q[:, h * d:], k[:, h * d:], v[:, h * d:]
            sh = scores[h * nb:]
            self._gemm(qh, kh, tk, d, m=tq, lda=ldq, ldw=ldk, out=sh, ldo=tk, batch=nb, sa=tq * ldq, sw=tk * ldk, so=tq * tk,
                       alpha=scale)
            _lib.check(self.lib.ap_softmax_rows(sh.data_ptr(), tk, nb * tq, tk, self._stream()), "ap_softmax_rows")
            self._gemm(sh, vh, d, tk, m=tq, lda=tk, ldw=ldv, out=out[:, h * d:], ldo=heads * d, batch=nb, sa=tq * tk,
                       sw=tk * ldv, so=tq * heads * d, w_kn=True)
        return out

    # ------------------------------------------------------------------ network
    def _trunk(self, images_u8: torch.Tensor):
        """uint8 [B, 1024, 1024, 3] (or [1024, 1024, 3]) -> per stage (x [B * H * W, C], H, W, C).  Every layer is row-wise,
        per window or per image, so B images are simply stacked along the rows (GEMMs planned like one image:
        ``ap_sgemm_stacked``): an image's activations do not depend on what it is batched with."""
        lib, st = self.lib, self._stream()
        if images_u8.dim() == 3:
            images_u8 = images_u8[None]
        B = int(images_u8.shape[0])
        cols = self._buf(B * 256 * 256, 147)
        for b in range(B):
            _lib.check(lib.ap_sam2_patchify(images_u8[b].data_ptr(), 1024, 1024, _lib.f3(MEAN), _lib.f3(STD),
                                            cols[b * 65536:].data_ptr(), st))
        # the positional embedding is the same for every image: batch stride 0 on the residual
        x = self._gemm(cols, self.w["pe.w"], EMBED, 147, bias=self.w["pe.b"], resid=self.w["pos"], m=65536, batch=B,
                       sa=65536 * 147, so=65536 * EMBED, sr=0, out=self._buf(B * 65536, EMBED))             # [B * 65536, 96]
        H = W = 256
        feats = []
        for i, (din, dout, heads, window, qpool) in enumerate(self.plan):
            g = lambda n: self.w[f"b{i}.{n}"]
            rows = B * H * W
            xn = self._ln(x, rows, din, g("norm1.weight"), g("norm1.bias"), 1e-6)
            shortcut = x
            if din != dout:
                shortcut = self._gemm(xn, g("proj.weight"), dout, din, bias=g("proj.bias"), stack=B)
                if qpool:
                    pooled = self._buf(B * (H // 2) * (W // 2), dout)
                    _lib.check(lib.ap_maxpool2x2(shortcut.data_ptr(), dout, B, H, W, dout, pooled.data_ptr(), st))
                    shortcut = pooled
            qkv = None
            if window > 0:
                nwy, nwx = -(-H // window), -(-W // window)
                nb, hh, ww = B * nwy * nwx, window, window
                # the window partition rides on the qkv GEMM's operand addressing where the split-f16 kernel takes the layer
                qkv = self._gemm_windows(xn, g("attn.qkv.weight"), 3 * dout, din, mode=1, B=B, H=H, W=W, window=window,
                                         bias=g("attn.qkv.bias"))
                if qkv is None:
                    win = self._buf(nb * hh * ww, din)
                    _lib.check(lib.ap_window_partition(xn.data_ptr(), B, H, W, din, window, win.data_ptr(), st))
                    xn = win
            else:
                nb, hh, ww = B, H, W
            t_k = hh * ww
            if qkv is None:
                qkv = self._gemm(xn, g("attn.qkv.weight"), 3 * dout, din, bias=g("attn.qkv.bias"), stack=B)     # [nb*t_k, 3*dout]
            d = dout // heads
            q, ldq, t_q = qkv, 3 * dout, t_k
            if qpool:
                qp = self._buf(nb * (hh // 2) * (ww // 2), dout)
                _lib.check(lib.ap_maxpool2x2(qkv.data_ptr(), 3 * dout, nb, hh, ww, dout, qp.data_ptr(), st))
                q, ldq, hh, ww = qp, dout, hh // 2, ww // 2
                t_q = hh * ww
            a = self._attention(q, qkv[:, dout:], qkv[:, 2 * dout:], nb=nb, heads=heads, tq=t_q, tk=t_k, d=d,
                                ldq=ldq, ldk=3 * dout, ldv=3 * dout)
            if qpool:
                H, W = H // 2, W // 2
                window = window // 2
            # x = shortcut + attn: the residual add rides on the projection GEMM's epilogue (image-wide blocks) or on the
            # window un-partition pass (windowed blocks)
            if window > 0:
                x2 = self._gemm_windows(a, g("attn.proj.weight"), dout, dout, mode=2, B=B, H=H, W=W, window=window,
                                        bias=g("attn.proj.bias"), resid=shortcut)
                if x2 is None:
                    a = self._gemm(a, g("attn.proj.weight"), dout, dout, bias=g("attn.proj.bias"), stack=B)
                    x2 = self._buf(B * H * W, dout)
                    _lib.check(lib.ap_window_unpartition_add(a.data_ptr(), shortcut.data_ptr(), B, H, W, dout, window, x2.data_ptr(), st))
            else:
                x2 = self._gemm(a, g("attn.proj.weight"), dout, dout, bias=g("attn.proj.bias"), resid=shortcut, stack=B)
            xn2 = self._ln(x2, B * H * W, dout, g("norm2.weight"), g("norm2.bias"), 1e-6)
            hid = self._gemm(xn2, g("mlp.layers.0.weight"), 4 * dout, dout, bias=g("mlp.layers.0.bias"), act=1, stack=B)
            x = self._gemm(hid, g("mlp.layers.1.weight"), dout, 4 * dout, bias=g("mlp.layers.1.bias"), resid=x2, stack=B)
            if i in self.stage_ends:
                feats.append((x, H, W, dout))
        return feats

    def image_features_batch(self, images_u8: torch.Tensor):
        """[B, 1024, 1024, 3] -> [(embed, feat_s0, feat_s1)] per image: the trunk runs on the stacked batch, neck and
        decoder inputs per image (their shapes are an image's)."""
        feats = self._trunk(images_u8)
        B = 1 if images_u8.dim() == 3 else int(images_u8.shape[0])
        out = []
        for b in range(B):
            per = [(x[b * h * w:(b + 1) * h * w], h, w, c) for (x, h, w, c) in feats]
            out.append(self._neck(per))
        return out

    def image_features(self, image_u8: torch.Tensor):
        """set_image: uint8 [1024, 1024, 3] on the device -> (embed [4096, 256] incl. no_mem + no_mask embeds,
        feat_s0 [65536, 32], feat_s1 [16384, 64])."""
        return self._neck(self._trunk(image_u8))

    def _neck(self, feats):
        lat = [self._gemm(x, self.w[f"neck{3 - i}.w"], 256, c, bias=self.w[f"neck{3 - i}.b"]) for i, (x, h, w, c) in enumerate(feats)]
        # top-down only into level 2 (stride 16) from level 3 (stride 32): FpnNeck fpn_top_down_levels [2, 3], nearest
        lvl2 = self._buf(64 * 64, 256)
        _lib.check(self.lib.ap_upsample2x_add(lvl2.data_ptr(), lat[2].data_ptr(), lat[3].data_ptr(), 32, 32, 256, self._stream()))
        s0 = self._gemm(lat[0], self.w["s0.w"], 32, 256, bias=self.w["s0.b"])
        s1 = self._gemm(lat[1], self.w["s1.w"], 64, 256, bias=self.w["s1.b"])
        embed = self._buf(4096, 256)
        _lib.check(self.lib.ap_add_rowvec(embed.data_ptr(), lvl2.data_ptr(), self.w["embed_add"].data_ptr(), 4096, 256, self._stream()))
        return embed, s0, s1

    def _dec_attn(self, prefix, q, k, v, tq, tk, internal):
        w = self.w
        qp = self._gemm(q, w[prefix + ".q_proj.w"], internal, 256, bias=w[prefix + ".q_proj.b"])
        kp = self._gemm(k, w[prefix + ".k_proj.w"], internal, 256, bias=w[prefix + ".k_proj.b"])
        vp = self._gemm(v, w[prefix + ".v_proj.w"], internal, 256, bias=w[prefix + ".v_proj.b"])
        a = self._attention(qp, kp, vp, nb=1, heads=8, tq=tq, tk=tk, d=internal // 8, ldq=internal, ldk=internal, ldv=internal)
        return a, w[prefix + ".out_proj.w"], w[prefix + ".out_proj.b"], internal

    def mask_logits(self, embed, s0, s1) -> torch.Tensor:
        """predict (multimask_output=False): -> logits [256, 256] of mask token 0."""
        lib, st, w = self.lib, self._stream(), self.w
        tokens, image_pe = w["tokens"], w["image_pe"]
        nt = tokens.shape[0]

        def add(a, b, n):
            o = self._buf(n)
            _lib.check(lib.ap_add(o.data_ptr(), a.data_ptr(), b.data_ptr(), n, st), "ap_add")
            return o

        queries, keys = tokens, embed
        for l in range(2):
            p = f"d{l}."
            ln = lambda x, rows, k: self._ln(x, rows, 256, w[p + f"norm{k}.w"], w[p + f"norm{k}.b"], 1e-5)
            if l == 0:
                a, ow, ob, internal = self._dec_attn(p + "self_attn", queries, queries, queries, nt, nt, 256)
                queries = self._gemm(a, ow, 256, internal, bias=ob)
            else:
                q = add(queries, tokens, nt * 256).view(nt, 256)
                a, ow, ob, internal = self._dec_attn(p + "self_attn", q, q, queries, nt, nt, 256)
                queries = self._gemm(a, ow, 256, internal, bias=ob, resid=queries)
            queries = ln(queries, nt, 1)
            q = add(queries, tokens, nt * 256).view(nt, 256)
            k = add(keys, image_pe, 4096 * 256).view(4096, 256)
            a, ow, ob, internal = self._dec_attn(p + "cross_attn_token_to_image", q, k, keys, nt, 4096, 128)
            queries = ln(self._gemm(a, ow, 256, internal, bias=ob, resid=queries), nt, 2)
            hid = self._gemm(queries, w[p + "mlp0.w"], 2048, 256, bias=w[p + "mlp0.b"], act=2)
            queries = ln(self._gemm(hid, w[p + "mlp1.w"], 256, 2048, bias=w[p + "mlp1.b"], resid=queries), nt, 3)
            q = add(queries, tokens, nt * 256).view(nt, 256)
            k = add(keys, image_pe, 4096 * 256).view(4096, 256)
            a, ow, ob, internal = self._dec_attn(p + "cross_attn_image_to_token", k, q, queries, 4096, nt, 128)
            keys = ln(self._gemm(a, ow, 256, internal, bias=ob, resid=keys), 4096, 4)
        q = add(queries, tokens, nt * 256).view(nt, 256)
        k = add(keys, image_pe, 4096 * 256).view(4096, 256)
        a, ow, ob, internal = self._dec_attn("dfin", q, k, keys, nt, 4096, 128)
        queries = self._ln(self._gemm(a, ow, 256, internal, bias=ob, resid=queries), nt, 256, w["dfin.norm.w"], w["dfin.norm.b"], 1e-5)
        # upscaling: ConvT(256->64) + feat_s1 -> LayerNorm2d -> GELU -> ConvT(64->32) + feat_s0 -> GELU
        g0 = self._gemm(keys, w["up0.w"], 256, 256)                                     # [4096, 64*4]
        u1 = self._buf(128 * 128, 64)
        _lib.check(lib.ap_convt2x2_shuffle(g0.data_ptr(), w["up0.b"].data_ptr(), s1.data_ptr(), u1.data_ptr(), 64, 64, 64, 0, st))
        u1 = self._ln(u1, 128 * 128, 64, w["up1.w"], w["up1.b"], 1e-6)
        _lib.check(lib.ap_gelu(u1.data_ptr(), 128 * 128 * 64, st))
        g1 = self._gemm(u1, w["up3.w"], 128, 64)                                        # [16384, 32*4]
        up = self._buf(256 * 256, 32)
        _lib.check(lib.ap_convt2x2_shuffle(g1.data_ptr(), w["up3.b"].data_ptr(), s0.data_ptr(), up.data_ptr(), 128, 128, 32, 1, st))
        h = queries[2:3]                                                                 # mask token 0 (obj, iou, mask0..3, prompts)
        h = self._gemm(h, w["hyper0.w"], 256, 256, bias=w["hyper0.b"], act=2)
        h = self._gemm(h, w["hyper1.w"], 256, 256, bias=w["hyper1.b"], act=2)
        h = self._gemm(h, w["hyper2.w"], 32, 256, bias=w["hyper2.b"])                    # [1, 32]
        return self._gemm(up, h, 1, 32).view(256, 256)                                   # logits[p] = sum_c up[p][c] * h[c]

    @torch.inference_mode()
    def count_flop(self) -> float:
        """Multiply-add work of ONE forward (2 * M * N * K per GEMM, 4 * tq * tk * d per attention head), counted by running the
        chain once eagerly on a blank thumbnail: what bench.py prices against the exact-f32 MFMA peak."""
        img = torch.zeros((self.input_size, self.input_size, 3), dtype=torch.uint8, device=self.device)
        self._flop = 0.0
        try:
            with torch.cuda.device(self.device):
                self.mask_logits(*self.image_features(img))
                torch.cuda.synchronize(self.device)
            return float(self._flop)
        finally:
            self._flop = None

    # ------------------------------------------------------------------ reference-facing API
    def _forward_mask(self, img: torch.Tensor) -> torch.Tensor:
        """uint8 [1024, 1024, 3] on the device -> float {0, 1} mask [1024, 1024] (set_image + predict + postprocess)."""
        logits = self.mask_logits(*self.image_features(img))
        mask = self._buf(1024, 1024)
        _lib.check(self.lib.ap_bilinear_up4_threshold(logits.data_ptr(), 256, C.c_float(self.mask_threshold), mask.data_ptr(),
                                                      self._stream()), "ap_bilinear_up4_threshold")
        return mask

    def _graph_mask(self, arr: np.ndarray) -> torch.Tensor:
        """The ~550-launch chain is the same for every slide (fixed 1024 x 1024 input, shapes from the weights only):
        it is captured once into a hipGraph (through torch's stream capture: every C-ABI launch goes to the capturing
        stream, buffers come from the graph's private pool) and replayed per slide -- one graph launch instead of 550
        kernel launches.  ATLASPATCH_SAM2_GRAPH=0 runs the launches one by one (per-kernel profiling)."""
        return self._graph_mask_device(torch.from_numpy(np.ascontiguousarray(arr)).to(self.device))

    @torch.inference_mode()
    def predict_logits(self, image_u8_1024: np.ndarray) -> torch.Tensor:
        img = torch.from_numpy(np.ascontiguousarray(image_u8_1024)).to(self.device)
        with torch.cuda.device(self.device):
            return self.mask_logits(*self.image_features(img))

    @torch.inference_mode()
    def predict_image(self, image, *, resize_to_input: bool = True) -> np.ndarray:
        """services/segmentation.py:120-140."""
        from PIL import Image
        arr = np.array(image.convert("RGB"), copy=True) if isinstance(image, Image.Image) else np.ascontiguousarray(image)
        if arr.dtype != np.uint8:
            arr = (arr * 255).astype(np.uint8) if arr.dtype.kind == "f" and arr.max() <= 1.0 else arr.astype(np.uint8)
        orig = (int(arr.shape[0]), int(arr.shape[1]))
        if orig != (self.input_size, self.input_size):
            arr = np.array(Image.fromarray(arr).resize((self.input_size, self.input_size), Image.Resampling.BILINEAR), copy=True)
        with torch.cuda.device(self.device):
            out = self._graph_mask(arr).cpu().numpy()
        if resize_to_input and orig != (self.input_size, self.input_size):
            pil = Image.fromarray((out * 255).astype(np.uint8), mode="L").resize((orig[1], orig[0]), resample=Image.Resampling.NEAREST)
            out = np.asarray(pil, dtype=np.float32) / 255.0
        return out

    RESAMPLER_CACHE = 16

    def _resampler_for(self, h: int, w: int):
        """(BILINEAR resampler to 1024 x 1024, NEAREST row / column indices back) for a thumbnail shape: an LRU — a hit
        moves the entry to the young end, eviction takes the oldest.  Callers hold the returned tuple for as long as they
        need it (a batch with more distinct shapes than the cache holds still works)."""
        from ..utils.resample import DeviceResampler, pillow_nearest_index
        S = self.input_size
        rs = self._resamplers.pop((h, w), None)
        if rs is None:
            yi = torch.from_numpy(pillow_nearest_index(S, h)).to(self.device)
            xi = torch.from_numpy(pillow_nearest_index(S, w)).to(self.device)
            rs = (DeviceResampler((h, w), (S, S), "bilinear", self.device), yi, xi)
        self._resamplers[(h, w)] = rs
        while len(self._resamplers) > self.RESAMPLER_CACHE:
            self._resamplers.pop(next(iter(self._resamplers)))
        return rs

    @torch.inference_mode()
    def predict_device(self, thumb: torch.Tensor, *, resize_to_input: bool = True) -> np.ndarray:
        """``predict_image`` for a thumbnail that already lives in HBM (uint8 [h, w, 3]): the PIL BILINEAR resize to
        1024 x 1024 (``_resize_input_for_sam``, segmentation.py:104-110) runs on the device bit-identically to Pillow
        (``ap_resample_u8``), the network is the same captured graph, and the mask returns to the thumbnail's shape by
        PIL NEAREST semantics (``_resize_mask``, :112-118) as a device gather.  One D2H copy: the float {0, 1} mask."""
        assert thumb.is_cuda and thumb.dtype == torch.uint8 and thumb.dim() == 3 and thumb.shape[2] == 3
        h, w = int(thumb.shape[0]), int(thumb.shape[1])
        S = self.input_size
        with torch.cuda.device(self.device):
            img = thumb.contiguous()
            if (h, w) != (S, S):
                rs = self._resampler_for(h, w)
                img = rs[0](img[None])[0]
            mask = self._graph_mask_device(img)
            if resize_to_input and (h, w) != (S, S):
                _, yi, xi = rs
                out = self._buf(h, w)
                _lib.check(self.lib.ap_gather2d_f32(mask.data_ptr(), S, S, yi.data_ptr(), xi.data_ptr(), h, w, out.data_ptr(),
                                                    self._stream()), "ap_gather2d_f32")
                mask = out
            return mask.cpu().numpy()

    def _graph_mask_device(self, img: torch.Tensor) -> torch.Tensor:
        """``_graph_mask`` for an input already in HBM (device-to-device copy into the graph's static input)."""
        import os
        if os.environ.get("ATLASPATCH_SAM2_GRAPH", "1") == "0":
            return self._forward_mask(img)
        if self._graph is None:
            self._static_img = torch.empty((self.input_size, self.input_size, 3), dtype=torch.uint8, device=self.device)
            self._static_img.copy_(img)
            self._forward_mask(self._static_img)                 # warm-up outside the capture (lazy initialisation)
            with _lib.HIP_CAPTURE_LOCK:                          # no weight upload of a side thread inside the capture
                torch.cuda.synchronize(self.device)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    self._static_mask = self._forward_mask(self._static_img)
            self._graph = graph
        self._static_img.copy_(img)
        self._graph.replay()
        return self._static_mask

    # ------------------------------------------------------------------ batched (``--seg-batch-size`` > 1)
    def _forward_masks(self, imgs: torch.Tensor) -> torch.Tensor:
        """uint8 [B, 1024, 1024, 3] -> float {0, 1} masks [B, 1024, 1024]: the trunk on the stacked batch, neck + decoder per
        image.  Mask b equals ``_forward_mask(imgs[b])`` bit for bit (tests)."""
        B = int(imgs.shape[0])
        masks = self._buf(B, 1024, 1024)
        for b, (embed, s0, s1) in enumerate(self.image_features_batch(imgs)):
            logits = self.mask_logits(embed, s0, s1)
            _lib.check(self.lib.ap_bilinear_up4_threshold(logits.data_ptr(), 256, C.c_float(self.mask_threshold),
                                                          masks[b].data_ptr(), self._stream()), "ap_bilinear_up4_threshold")
        return masks

    def _graph_masks_device(self, imgs: torch.Tensor) -> torch.Tensor:
        """One captured graph per batch size (the reference batches ``seg_batch_size`` thumbnails per forward,
        segmentation.py:142-180); the returned tensor is the graph's static output: consume it before the next call."""
        import os
        B = int(imgs.shape[0])
        if B == 1:
            return self._graph_mask_device(imgs[0])[None]
        if os.environ.get("ATLASPATCH_SAM2_GRAPH", "1") == "0":
            return self._forward_masks(imgs)
        entry = self._batch_graphs.get(B)
        if entry is None:
            static_in = torch.empty((B, self.input_size, self.input_size, 3), dtype=torch.uint8, device=self.device)
            static_in.copy_(imgs)
            self._forward_masks(static_in)                       # warm-up outside the capture (scratch growth, lazy init)
            with _lib.HIP_CAPTURE_LOCK:
                torch.cuda.synchronize(self.device)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    static_out = self._forward_masks(static_in)
            entry = self._batch_graphs[B] = (graph, static_in, static_out)
            while len(self._batch_graphs) > 3:                   # a run uses one batch size plus at most two remainder sizes (cohort, MAX_BATCH chunk)
                self._batch_graphs.pop(next(iter(self._batch_graphs)))
        graph, static_in, static_out = entry
        static_in.copy_(imgs)
        graph.replay()
        return static_out

    @torch.inference_mode()
    def capture_graphs(self, batch_sizes) -> None:
        """Capture the graphs of every batch size a run will use BEFORE its worker threads start (the runner calls this
        with {seg_batch_size, n_slides % seg_batch_size}): stream capture then never coincides with the coordinate workers'
        legacy-stream calls (hipMalloc / hipFree / hipMemcpyAsync of ap_contours / ap_grid_coords) -- without this the
        remainder-size graph would be captured during the LAST group, while earlier groups' workers are in flight."""
        S = self.input_size
        with torch.cuda.device(self.device):
            for B in sorted({min(int(b), self.MAX_BATCH) for b in batch_sizes if int(b) > 0}):
                if (B == 1 and self._graph is not None) or (B > 1 and B in self._batch_graphs):
                    continue
                self._graph_masks_device(torch.zeros((B, S, S, 3), dtype=torch.uint8, device=self.device))
            torch.cuda.synchronize(self.device)

    @torch.inference_mode()
    def predict_batch_device(self, thumbs, *, resize_to_input: bool = True) -> list:
        """``predict_device`` for several thumbnails (each uint8 [h, w, 3] in HBM, any sizes) in ONE forward: each is resized
        to 1024 x 1024 as Pillow would, the batch goes through the trunk stacked, each mask returns to its thumbnail's shape.
        Element i equals ``predict_device(thumbs[i])`` bit for bit."""
        S = self.input_size
        if len(thumbs) > self.MAX_BATCH:            # stage-1 windows: 1024 per image, 65 535 per fused-attention launch
            out = []
            for i in range(0, len(thumbs), self.MAX_BATCH):
                out += self.predict_batch_device(thumbs[i:i + self.MAX_BATCH], resize_to_input=resize_to_input)
            return out
        with torch.cuda.device(self.device):
            imgs = torch.empty((len(thumbs), S, S, 3), dtype=torch.uint8, device=self.device)
            shapes, used = [], []              # the batch keeps its own references: cache eviction cannot take them away
            for b, thumb in enumerate(thumbs):
                assert thumb.is_cuda and thumb.dtype == torch.uint8 and thumb.dim() == 3 and thumb.shape[2] == 3
                h, w = int(thumb.shape[0]), int(thumb.shape[1])
                shapes.append((h, w))
                if (h, w) == (S, S):
                    imgs[b].copy_(thumb)
                    used.append(None)
                    continue
                rs = self._resampler_for(h, w)
                used.append(rs)
                imgs[b].copy_(rs[0](thumb.contiguous()[None])[0])
            masks = self._graph_masks_device(imgs)
            out = []
            for b, (h, w) in enumerate(shapes):
                m = masks[b]
                if resize_to_input and (h, w) != (S, S):
                    _, yi, xi = used[b]
                    g = self._buf(h, w)
                    _lib.check(self.lib.ap_gather2d_f32(m.data_ptr(), S, S, yi.data_ptr(), xi.data_ptr(), h, w, g.data_ptr(),
                                                        self._stream()), "ap_gather2d_f32")
                    m = g
                out.append(m.cpu().numpy())
            return out

    def close(self) -> None:
        self._graph = None
        self._batch_graphs = {}
        self._static_img = self._static_mask = None
        self.w = {}


def load_sam2_state_dict(checkpoint_path) -> dict:
    """The reference's checkpoint layout (``torch.load(path)["model"]``, segmentation.py:66-67); a bare state dict and a
    DataParallel ``module.`` prefix are accepted too, and so is the Hugging Face ``Sam2Model`` layout (a torch file or a
    ``.safetensors`` export), which is renamed through ``sam2_keys.hf_to_facebook`` — that map is pinned against
    ``transformers.models.sam2`` by tests/golden/gen_golden_hf_sam2.py.  Every tensor the image path reads is checked up
    front so that a checkpoint with other key names fails with the list of what is missing instead of a KeyError
    mid-construction."""
    from .sam2_keys import hf_to_facebook, is_hf_layout
    path = str(checkpoint_path)
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        obj = load_file(path, device="cpu")
    else:
        obj = torch.load(path, map_location="cpu", weights_only=True)
    sd = obj["model"] if isinstance(obj, dict) and "model" in obj else obj
    if sd and all(k.startswith("module.") for k in sd):
        sd = {k[len("module."):]: v for k, v in sd.items()}
    if is_hf_layout(sd):
        sd = hf_to_facebook(sd)
    missing = [k for k in required_sam2_keys() if k not in sd]
    if missing:
        raise KeyError(f"{checkpoint_path}: {len(missing)} tensors of the SAM2.1 Hiera-T image path are missing, e.g. "
                       f"{missing[:6]} (expected the sam2 package's state-dict names or the transformers Sam2Model layout)")
    return sd


def required_sam2_keys() -> list:
    """Names (sam2 package state dict) of every tensor ``Sam2HipPredictor`` reads."""
    plan, _ = block_plan()
    t, d, p = "image_encoder.trunk.", "sam_mask_decoder.", "sam_prompt_encoder."
    keys = [t + "patch_embed.proj.weight", t + "patch_embed.proj.bias", t + "pos_embed", t + "pos_embed_window", "no_mem_embed",
            p + "no_mask_embed.weight", p + "pe_layer.positional_encoding_gaussian_matrix", p + "not_a_point_embed.weight",
            p + "point_embeddings.2.weight", p + "point_embeddings.3.weight",
            d + "obj_score_token.weight", d + "iou_token.weight", d + "mask_tokens.weight",
            d + "conv_s0.weight", d + "conv_s0.bias", d + "conv_s1.weight", d + "conv_s1.bias",
            d + "transformer.norm_final_attn.weight", d + "transformer.norm_final_attn.bias",
            d + "output_upscaling.0.weight", d + "output_upscaling.0.bias", d + "output_upscaling.1.weight",
            d + "output_upscaling.1.bias", d + "output_upscaling.3.weight", d + "output_upscaling.3.bias"]
    for i, (din, dout, _, _, _) in enumerate(plan):
        b = f"{t}blocks.{i}."
        keys += [b + n for n in ("norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight",
                                 "attn.proj.bias", "norm2.weight", "norm2.bias", "mlp.layers.0.weight", "mlp.layers.0.bias",
                                 "mlp.layers.1.weight", "mlp.layers.1.bias")]
        if din != dout:
            keys += [b + "proj.weight", b + "proj.bias"]
    for n in range(4):
        keys += [f"image_encoder.neck.convs.{n}.conv.weight", f"image_encoder.neck.convs.{n}.conv.bias"]
    for l in range(2):
        b = f"{d}transformer.layers.{l}."
        for a in ("self_attn", "cross_attn_token_to_image", "cross_attn_image_to_token"):
            for pj in ("q_proj", "k_proj", "v_proj", "out_proj"):
                keys += [b + f"{a}.{pj}.weight", b + f"{a}.{pj}.bias"]
        keys += [b + f"norm{k}.{w}" for k in range(1, 5) for w in ("weight", "bias")]
        keys += [b + f"mlp.layers.{k}.{w}" for k in range(2) for w in ("weight", "bias")]
    for pj in ("q_proj", "k_proj", "v_proj", "out_proj"):
        keys += [d + f"transformer.final_attn_token_to_image.{pj}.weight", d + f"transformer.final_attn_token_to_image.{pj}.bias"]
    keys += [d + f"output_hypernetworks_mlps.0.layers.{k}.{w}" for k in range(3) for w in ("weight", "bias")]
    return keys
