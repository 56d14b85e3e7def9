"""Per-operator parity of the HIP kernels behind the C ABI (ap_gemm / ap_layernorm / ap_attention)
against a plain PyTorch fp32 reference of the same op on the same (dtype-rounded) inputs.

Tolerances: f32 results ~1e-5; f16 / bf16 results are rounded to the operand type, so the bound is
half an ulp of the output magnitude plus operand rounding (written per test)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

EPI = {"bias": 0, "gelu": 1, "resid": 2}


@pytest.fixture(scope="module")
def env():
    from atlaspatch_amd import _lib
    dev = torch.device("cuda:0")
    return _lib, _lib.load(), dev, _lib.current_stream_ptr(dev)


def _gemm(env, dt, epi, A, W, bias, gamma, out, impl):
    _lib, lib, dev, stream = env
    M, K = A.shape
    _lib.check(lib.ap_gemm(_lib.torch_dtype_code(dt), EPI[epi], A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0),
                           M, W.shape[0], K, bias.data_ptr(), gamma.data_ptr() if gamma is not None else None,
                           out.data_ptr(), out.stride(0), impl, 0, stream), "ap_gemm")
    torch.cuda.synchronize()


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("epi", ["bias", "gelu", "resid"])
@pytest.mark.parametrize("shape", [(300, 256, 128), (1182, 768, 768), (100, 256, 384), (3941, 768, 3072),
                                   (20000, 2304, 768)])
def test_gemm_persistent_kernel_vs_torch(env, dt, epi, shape):
    """256x256 persistent kernel (impl 256): ragged M, one-tile and multi-tile-per-workgroup problems, all
    epilogues; also bit-identical to the 128x128 kernel (the product may pick either by problem size)."""
    _lib, lib, dev, stream = env
    M, N, K = shape
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    A = (torch.rand((M, K), device=dev, generator=g) * 2 - 1).to(dt)
    W = ((torch.rand((N, K), device=dev, generator=g) * 2 - 1) * (2.0 / K ** 0.5)).to(dt)
    bias = torch.rand(N, device=dev, generator=g) - 0.5
    gamma = torch.rand(N, device=dev, generator=g) + 0.5 if epi != "gelu" and M % 2 else None
    resid = torch.rand((M, N), device=dev, generator=g) if epi == "resid" else None
    ref = A.float() @ W.float().t() + bias
    if epi == "gelu":
        ref = torch.nn.functional.gelu(ref)
    elif gamma is not None:
        ref = ref * gamma
    if epi == "resid":
        ref = resid + ref
    outs = {}
    for impl in (256, 128):
        out = resid.clone() if epi == "resid" else torch.full((M, N), float("nan"), device=dev, dtype=dt)
        _gemm(env, dt, epi, A, W, bias, gamma, out, impl)
        outs[impl] = out
    # |C| <= ~4: half an ulp of f16 is 2e-3, of bf16 1.6e-2; the f32 residual epilogue is exact to f32 rounding
    tol = 2e-4 if epi == "resid" else (3e-3 if dt == torch.float16 else 2e-2)
    assert (outs[256].float() - ref).abs().max().item() <= tol
    assert torch.equal(outs[256], outs[128])


@pytest.mark.parametrize("shape", [(2304, 768, "bias"), (768, 768, "bias"), (3072, 768, "gelu"), (768, 3072, "bias")])
def test_gemm_tile_seam_with_cold_bias(env, shape):
    """Regression screen for the tile seam of the persistent kernel: the next tile's bias is fetched while a tile ends.
    The four ViT-B/16 GEMM shapes at the row count of a 348-tile slide (several tiles per workgroup, ragged last row
    tile), many launches each with a DIFFERENT, cache-cold bias vector (vectors 1 MiB apart in a 256 MiB pool, so the
    load misses L2 and comes back late); every result must equal the 128x128 kernel's bit for bit.  (A counted wait
    behind the epilogue's stores once let a late bias load slip through: a store may retire before an older load;
    seen as one or two wrong images in about one forward of ten.)"""
    _lib, lib, dev, stream = env
    dt = torch.float16
    N, K, epi = shape
    M = 348 * 197
    g = torch.Generator(device=dev).manual_seed(9)
    A = (torch.rand((M, K), device=dev, generator=g) * 2 - 1).to(dt)
    W = ((torch.rand((N, K), device=dev, generator=g) * 2 - 1) * (2.0 / K ** 0.5)).to(dt)
    pool = torch.rand((256, 262144), device=dev, generator=g) - 0.5          # row stride 1 MiB
    out256 = torch.empty((M, N), device=dev, dtype=dt)
    out128 = torch.empty((M, N), device=dev, dtype=dt)
    for it in range(48):
        bias = pool[(it * 37) % 256, :N]
        _gemm(env, dt, epi, A, W, bias, None, out256, 256)
        _gemm(env, dt, epi, A, W, bias, None, out128, 128)
        assert torch.equal(out256, out128), f"launch {it}"


# ----------------------------------------------------------------------------- fused-LayerNorm GEMM epilogues
def _fused(env, dt, epi, A, W, bias, colsum, rowstats, partial, out, impl=256):
    _lib, lib, dev, stream = env
    M, K = A.shape
    p = lambda t: t.data_ptr() if t is not None else None
    _lib.check(lib.ap_gemm_fused(_lib.torch_dtype_code(dt), epi, A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0),
                                 M, W.shape[0], K, bias.data_ptr(), p(colsum), p(rowstats), p(partial), out.data_ptr(),
                                 out.stride(0), impl, stream), "ap_gemm_fused")
    torch.cuda.synchronize()


@pytest.mark.parametrize("dt,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
@pytest.mark.parametrize("shape", [(1, 512, 256), (265, 8192, 1536), (3000, 2048, 384), (256 * 9 + 5, 1024, 768)])
def test_gemm_norm_swiglu_epilogue(env, dt, tol, shape):
    """AP_EPI_NORM_SWIGLU (uni_v2's SwiGLUPacked fc1): LayerNorm algebra + silu(x1) * x2 in the epilogue on row-interleaved
    weights, against torch on the same rounded operands; the 128 x 128 and the persistent kernel give the same bits."""
    _lib, lib, dev, stream = env
    M, N, K = shape                                    # N = 2 H packed rows
    H = N // 2
    g = torch.Generator(device=dev).manual_seed(M + N)
    x = (torch.randn((M, K), device=dev, generator=g) * 1.5 + 0.2).to(dt)
    W = (torch.randn((N, K), device=dev, generator=g) * (1.5 / K ** 0.5)).to(dt)         # the LayerNorm gain is folded already
    bias = torch.randn(N, device=dev, generator=g) * 0.2
    xf = x.float()
    mean, var = xf.mean(1, keepdim=True), xf.var(1, unbiased=False, keepdim=True)
    rstd = torch.rsqrt(var + 1e-6)
    rowstats = torch.cat([rstd, -mean * rstd], 1).contiguous()
    colsum = W.float().sum(1)
    # interleave: row 64 q + 32 half + j  <-  packed row half * H + 32 q + j
    r = torch.arange(N, device=dev)
    src = torch.where((r % 64) < 32, torch.zeros_like(r), torch.full_like(r, H)) + 32 * (r // 64) + r % 32
    Wi, bi, ci = W[src].contiguous(), bias[src].contiguous(), colsum[src].contiguous()
    outs = {}
    for impl in (256, 128):
        if impl == 256 and (N % 256 or K % 128):
            continue
        out = torch.full((M, H), float("nan"), device=dev, dtype=dt)
        _fused(env, dt, 8, x, Wi, bi, ci, rowstats, None, out, impl=impl)
        outs[impl] = out
    y = ((xf - mean) * rstd) @ W.float().t() + bias
    want = torch.nn.functional.silu(y[:, :H]) * y[:, H:]
    got = next(iter(outs.values()))
    assert (got.float() - want).abs().max().item() <= tol * max(1.0, want.abs().max().item())
    if len(outs) == 2:
        assert torch.equal(outs[256], outs[128])


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(1, 768, 768), (197, 768, 768), (1000, 768, 3072), (256 * 40 + 77, 1024, 1024),
                                   (348 * 197, 768, 768)])
def test_gemm_resid_stats_epilogue(env, dt, shape):
    """AP_EPI_RESID_STATS: x <- T(x + T(A W^T + b)) in place and per (row, 64-column group) the sum / sum of squares of
    the NEW row; then ap_rowstats_finalize -> (rstd, -mean rstd).  Against torch on the same rounded operands: the stream
    is exact up to the rounding of the f32 accumulator (one ulp of T at the branch's magnitude where the two round
    differently), the partial sums are sums of the values actually stored (exact up to f32 summation order); ragged M,
    M < 256, several tiles per workgroup; the same launch repeated is bit-identical."""
    _lib, lib, dev, stream = env
    M, N, K = shape
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    A = torch.randn((M, K), device=dev, generator=g).to(dt)
    W = (torch.randn((N, K), device=dev, generator=g) * 0.05).to(dt)
    bias = torch.randn(N, device=dev, generator=g) * 0.1
    x0 = (torch.randn((M, N), device=dev, generator=g) * 2.0 + 0.3).to(dt)
    runs = []
    for _ in range(3):
        x = x0.clone()
        part = torch.full((M, N // 64, 2), float("nan"), device=dev)
        _fused(env, dt, 6, A, W, bias, None, None, part, x)
        runs.append((x, part))
    assert all(torch.equal(runs[0][0], r[0]) and torch.equal(runs[0][1], r[1]) for r in runs[1:])
    x, part = runs[0]
    # the 128 x 128 kernel (what small problems are served by) gives the same stream AND the same partial sums, bit for bit
    x128 = x0.clone()
    part128 = torch.full((M, N // 64, 2), float("nan"), device=dev)
    _fused(env, dt, 6, A, W, bias, None, None, part128, x128, impl=128)
    assert torch.equal(x128, x) and torch.equal(part128, part)
    branch = (A.float() @ W.float().t() + bias)
    want = (x0.float() + branch.to(dt).float()).to(dt)
    ulp = 2.0 ** -10 if dt == torch.float16 else 2.0 ** -7
    # where the kernel's f32 accumulation order rounds the branch the other way (one ulp of the branch), x moves by that
    # much or lands on a neighbouring value of T
    assert bool(((x.float() - want.float()).abs() <= ulp * (branch.abs() + want.float().abs()) * 1.001 + 1e-6).all())
    assert (x != want).float().mean().item() < 2e-3
    xs = x.float().view(M, N // 64, 64)
    assert torch.allclose(part[..., 0], xs.sum(-1), rtol=0, atol=2e-3)
    assert torch.allclose(part[..., 1], (xs * xs).sum(-1), rtol=2e-6, atol=1e-3)
    rs = torch.empty((M, 2), device=dev)
    _lib.check(lib.ap_rowstats_finalize(part.data_ptr(), M, N // 64, N, 1e-6, rs.data_ptr(), stream), "ap_rowstats_finalize")
    xf = x.double()
    mean, rstd = xf.mean(-1), torch.rsqrt(xf.var(-1, unbiased=False) + 1e-6)
    assert ((rs[:, 0].double() - rstd).abs() / rstd).max().item() < 2e-6
    assert (rs[:, 1].double() + mean * rstd).abs().max().item() < 2e-6


@pytest.mark.parametrize("dt,tol", [(torch.float16, 1.5e-3), (torch.bfloat16, 1.2e-2)])
@pytest.mark.parametrize("gelu", [False, True])
@pytest.mark.parametrize("shape", [(1, 768, 768), (300, 2304, 768), (20000, 3072, 768), (348 * 197, 1536, 1024)])
def test_gemm_norm_epilogue_equals_layernorm_then_linear(env, dt, tol, gelu, shape):
    """AP_EPI_NORM / AP_EPI_NORM_GELU: rstd (x W'^T - mean colsum) + b' with W' = T(W gamma), b' = b + W beta equals
    Linear(LayerNorm(x)) (+ GELU) computed by torch in float32 -- including rows with a mean far from zero, where the
    rank-one mean correction cancels most of the accumulator.  Bound: rounding of the output to T plus the rounding of
    W gamma to T (element-wise, relative to the output scale)."""
    _lib, lib, dev, stream = env
    M, N, K = shape
    g = torch.Generator(device=dev).manual_seed(M + N + K + int(gelu))
    x = torch.randn((M, K), device=dev, generator=g) * 1.5
    x[: max(1, M // 4)] += 6.0                                  # rows with |mean| = 4 sigma
    x[:, 7] += 40.0                                             # one massive channel
    x = x.to(dt)
    Wl = torch.randn((N, K), device=dev, generator=g) * 0.05
    gamma = 1.0 + 0.2 * torch.randn(K, device=dev, generator=g)
    beta = 0.1 * torch.randn(K, device=dev, generator=g)
    b = 0.1 * torch.randn(N, device=dev, generator=g)
    Wf = (Wl * gamma).to(dt)
    colsum = Wf.float().sum(-1).contiguous()
    bf = (b + Wl @ beta).contiguous()
    stats = torch.empty((M, 2), device=dev)
    x16 = torch.empty((M, K), device=dev, dtype=dt)
    _lib.check(lib.ap_stream_init(_lib.torch_dtype_code(dt), x.float().contiguous().data_ptr(), M, K, 1e-6, x16.data_ptr(),
                                  stats.data_ptr(), stream), "ap_stream_init")
    assert torch.equal(x16, x)
    outs = []
    for _ in range(3):
        out = torch.full((M, N), float("nan"), device=dev, dtype=dt)
        _fused(env, dt, 5 if gelu else 4, x16, Wf, bf, colsum, stats, None, out)
        outs.append(out)
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    out128 = torch.full((M, N), float("nan"), device=dev, dtype=dt)
    _fused(env, dt, 5 if gelu else 4, x16, Wf, bf, colsum, stats, None, out128, impl=128)
    assert torch.equal(out128, outs[0])                      # 128 x 128 twin: same bits
    want = torch.nn.functional.layer_norm(x.float(), (K,), gamma, beta, 1e-6) @ Wl.t() + b
    if gelu:
        want = torch.nn.functional.gelu(want)
    err = ((outs[0].float() - want).abs() / (want.abs() + 0.05 * want.abs().max())).max().item()
    assert err <= tol * 4, err          # element-wise statistic: ~4 x the norm-wise one
    assert (torch.linalg.norm(outs[0].float() - want) / torch.linalg.norm(want)).item() <= tol


def test_gemm_f32_exact_mfma(env):
    _lib, lib, dev, stream = env
    g = torch.Generator(device=dev).manual_seed(3)
    M, N, K = 777, 384, 160
    A = torch.rand((M, K), device=dev, generator=g) * 2 - 1
    W = (torch.rand((N, K), device=dev, generator=g) * 2 - 1) * 0.2
    bias = torch.rand(N, device=dev, generator=g)
    out = torch.empty((M, N), device=dev)
    _gemm(env, torch.float32, "bias", A, W, bias, None, out, 0)
    ref = (A.double() @ W.double().t() + bias).float()
    assert (out - ref).abs().max().item() <= 2e-5


def _split_rows(env, W):
    _lib, lib, dev, stream = env
    out = torch.empty_like(W)
    _lib.check(lib.ap_split_f16_weights(W.data_ptr(), out.data_ptr(), W.numel(), stream), "ap_split_f16_weights")
    torch.cuda.synchronize()
    return out


def test_split_f16_weight_rows_are_hi_then_scaled_lo_per_32(env):
    """ap_split_f16_weights: per 32 consecutive float32 values 32 float16 `hi` = f16(w) followed by 32 float16
    `lo` = f16((w - hi) * 2^11); hi + lo * 2^-11 reproduces w to 2^-22 |w| (and exactly below f16's subnormal range)."""
    _lib, lib, dev, stream = env
    g = torch.Generator(device=dev).manual_seed(5)
    W = torch.randn((96, 128), device=dev, generator=g) * torch.exp2(torch.randint(-20, 4, (96, 128), device=dev, generator=g).float())
    W[0, :4] = torch.tensor([0.0, -0.0, 1.0, 65504.0], device=dev)
    got = _split_rows(env, W).cpu().view(torch.float16).reshape(-1, 2, 32)
    w = W.cpu().reshape(-1, 32)
    hi = w.to(torch.float16)
    lo = ((w - hi.float()) * 2048.0).to(torch.float16)
    assert torch.equal(got[:, 0].view(torch.int16), hi.view(torch.int16))
    assert torch.equal(got[:, 1].view(torch.int16), lo.view(torch.int16))
    back = hi.double() + lo.double() / 2048.0
    assert ((back - w.double()).abs() <= w.double().abs() * 2.0 ** -21 + 2.0 ** -36).all()
    with pytest.raises(_lib.HipLibraryError):
        _lib.check(lib.ap_split_f16_weights(W.data_ptr(), W.data_ptr(), 48, stream))        # not whole groups of 32 / in place


@pytest.mark.parametrize("epi", ["bias", "gelu", "resid"])
@pytest.mark.parametrize("case", [((777, 384, 160), 1.0, 0.2), ((3941, 2304, 768), 1.0, 0.03), ((1182, 768, 3072), 0.02, 0.02),
                                  ((1, 128, 64), 1.0, 0.1), ((1000, 768, 768), 1.0, 1e-5), ((1000, 768, 768), 300.0, 0.03),
                                  ((2000, 1024, 1024), None, 0.03)])
def test_gemm_split_f16_products_are_float32_accurate(env, epi, case):
    """ap_gemm impl 129 (what AP_VIT_OPT_SPLIT_F16 runs): float32 buffers, every product as
    w_hi a_hi + 2^-11 (w_hi a_lo + w_lo a_hi) on the f16 MFMA with f32 accumulation.  Against a float64 matmul of the same
    float32 operands it must be as accurate as the exact f32 MFMA chain (impl 128) -- norm-wise <= 1e-6 and no worse than
    1.25x the exact chain's own error -- on ragged M, large activations and operands whose magnitudes span 18 binades;
    a matrix whose weights ALL sit below f16's normal range (1e-5: hi and lo are subnormal, 2^-24 / 2^-35 absolute) keeps
    2^-19 relative (bound 4e-6) -- no checkpoint has one; bit-repeatable."""
    _lib, lib, dev, stream = env
    (M, N, K), ascale, wscale = case
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    A = torch.randn((M, K), device=dev, generator=g)
    A = A * ascale if ascale is not None else A * torch.exp2(torch.randint(-12, 6, (M, K), device=dev, generator=g).float())
    W = torch.randn((N, K), device=dev, generator=g) * wscale
    bias = torch.randn(N, device=dev, generator=g) * 0.1 * wscale * max(1.0, ascale or 1.0) * K ** 0.5
    gamma = torch.rand(N, device=dev, generator=g) + 0.5 if epi == "resid" else None
    resid = torch.randn((M, N), device=dev, generator=g) * wscale * (ascale or 1.0) if epi == "resid" else None
    C = A.double() @ W.double().t() + bias.double()
    ref = torch.nn.functional.gelu(C) if epi == "gelu" else (resid.double() + C * gamma.double() if epi == "resid" else C)
    Ws = _split_rows(env, W)
    err = {}
    for tag, impl, w in (("exact", 128, W), ("split", 129, Ws), ("split again", 129, Ws)):
        out = resid.clone() if epi == "resid" else torch.full((M, N), float("nan"), device=dev)
        _lib.check(lib.ap_gemm(_lib.AP_F32, EPI[epi], A.data_ptr(), K, w.data_ptr(), K, M, N, K, bias.data_ptr(),
                               gamma.data_ptr() if gamma is not None else None, out.data_ptr(), N, impl, 0, stream), "ap_gemm")
        torch.cuda.synchronize()
        err[tag] = float((out.double() - ref).norm() / ref.norm())
        if tag == "split":
            first = out
    assert torch.equal(first, out)
    print(f"SPLIT_F16 gemm {epi} {case}: exact {err['exact']:.2e} split {err['split']:.2e}")
    if wscale < 6e-5:
        assert err["split"] <= 4e-6, err
    else:
        assert err["split"] <= 1e-6 and err["split"] <= 1.25 * err["exact"] + 1e-8, err


@pytest.mark.parametrize("case", [(65536, 288, 96, 0, False), (4100, 96, 384, 0, True), (4900, 1152, 384, 0, False), (4096, 1536, 384, 1, False),
                                  (1000, 384, 1536, 0, True), (130, 32, 64, 1, True), (257, 576, 192, 0, False)])
def test_gemm_split_f16_rowwise_layer_with_separate_residual(env, case):
    """ap_gemm_split_f16 (the SAM2 trunk's row-wise layers): any N % 32 == 0 (the 128-wide tile's tail masked), optional
    bias, erf GELU, optional SEPARATE residual added after the activation; float32-accurate against float64 (<= 5e-7
    norm-wise); columns beyond N and rows beyond M of a padded output buffer stay untouched."""
    _lib, lib, dev, stream = env
    M, N, K, act, with_res = case
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    A = torch.randn((M, K), device=dev, generator=g)
    W = torch.randn((N, K), device=dev, generator=g) * K ** -0.5
    bias = torch.randn(N, device=dev, generator=g) if M % 2 == 0 else None
    res = torch.randn((M, N + 8), device=dev, generator=g) if with_res else None
    out = torch.full((M + 3, N + 4), 7.0, device=dev)
    Ws = _split_rows(env, W)
    _lib.check(lib.ap_gemm_split_f16(A.data_ptr(), K, Ws.data_ptr(), M, N, K, bias.data_ptr() if bias is not None else None, act,
                                     res.data_ptr() if with_res else None, N + 8, out.data_ptr(), N + 4, stream), "ap_gemm_split_f16")
    torch.cuda.synchronize()
    ref = A.double() @ W.double().t()
    if bias is not None:
        ref = ref + bias.double()
    if act:
        ref = torch.nn.functional.gelu(ref)
    if with_res:
        ref = ref + res[:, :N].double()
    err = float((out[:M, :N].double() - ref).norm() / ref.norm())
    assert err <= 5e-7, err
    assert bool((out[M:] == 7.0).all()) and bool((out[:, N:] == 7.0).all())


@pytest.mark.parametrize("geom", [(1, 64, 64, 14), (2, 32, 32, 7), (1, 256, 256, 8), (3, 30, 22, 8)])
def test_gemm_split_f16_windows_equals_partition_then_gemm(env, geom):
    """ap_gemm_split_f16_windows: the window partition as operand addressing (mode 1) and the un-partition + residual add
    as output addressing (mode 2) give the SAME BITS as ap_window_partition -> ap_gemm_split_f16 and ap_gemm_split_f16 ->
    ap_window_unpartition_add, zero padding at the right / bottom edge included."""
    _lib, lib, dev, stream = env
    B, H, W, ws = geom
    Cin, Cout = 96, 288
    nwy, nwx = -(-H // ws), -(-W // ws)
    M = B * nwy * nwx * ws * ws
    g = torch.Generator(device=dev).manual_seed(B * H + ws)
    x = torch.randn((B * H * W, Cin), device=dev, generator=g)
    Wq = torch.randn((Cout, Cin), device=dev, generator=g) * 0.1
    bq = torch.randn(Cout, device=dev, generator=g)
    Wqs = _split_rows(env, Wq)
    # mode 1
    win = torch.empty((M, Cin), device=dev)
    _lib.check(lib.ap_window_partition(x.data_ptr(), B, H, W, Cin, ws, win.data_ptr(), stream))
    want = torch.empty((M, Cout), device=dev); got = torch.full((M, Cout), float("nan"), device=dev)
    _lib.check(lib.ap_gemm_split_f16(win.data_ptr(), Cin, Wqs.data_ptr(), M, Cout, Cin, bq.data_ptr(), 0, None, 0, want.data_ptr(), Cout, stream))
    _lib.check(lib.ap_gemm_split_f16_windows(x.data_ptr(), Cin, Wqs.data_ptr(), M, Cout, Cin, bq.data_ptr(), 0, None, 0, got.data_ptr(), Cout,
                                             1, B, H, W, ws, stream), "ap_gemm_split_f16_windows")
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    # mode 2
    a = torch.randn((M, Cout), device=dev, generator=g)
    Wp = torch.randn((Cin, Cout), device=dev, generator=g) * 0.1        # [N = Cin, K = Cout]
    bp = torch.randn(Cin, device=dev, generator=g)
    Wps = _split_rows(env, Wp)
    shortcut = torch.randn((B * H * W, Cin), device=dev, generator=g)
    tmp = torch.empty((M, Cin), device=dev)
    _lib.check(lib.ap_gemm_split_f16(a.data_ptr(), Cout, Wps.data_ptr(), M, Cin, Cout, bp.data_ptr(), 0, None, 0, tmp.data_ptr(), Cin, stream))
    want2 = torch.empty((B * H * W, Cin), device=dev); got2 = torch.full((B * H * W, Cin), float("nan"), device=dev)
    _lib.check(lib.ap_window_unpartition_add(tmp.data_ptr(), shortcut.data_ptr(), B, H, W, Cin, ws, want2.data_ptr(), stream))
    _lib.check(lib.ap_gemm_split_f16_windows(a.data_ptr(), Cout, Wps.data_ptr(), M, Cin, Cout, bp.data_ptr(), 0, shortcut.data_ptr(), Cin,
                                             got2.data_ptr(), Cin, 2, B, H, W, ws, stream), "ap_gemm_split_f16_windows")
    torch.cuda.synchronize()
    # (the stand-alone pass adds resid + v, the epilogue v + resid: the same float32 sum)
    assert torch.equal(got2, want2)
    with pytest.raises(_lib.HipLibraryError):
        _lib.check(lib.ap_gemm_split_f16_windows(a.data_ptr(), Cout, Wps.data_ptr(), M + 32, Cin, Cout, bp.data_ptr(), 0, None, 0, got2.data_ptr(), Cin,
                                                 2, B, H, W, ws, stream))


def test_gemm_split_f16_refuses_other_types(env):
    _lib, lib, dev, stream = env
    A = torch.zeros((128, 64), device=dev, dtype=torch.float16)
    b = torch.zeros(128, device=dev)
    with pytest.raises(_lib.HipLibraryError, match="float32"):
        _lib.check(lib.ap_gemm(_lib.AP_F16, 0, A.data_ptr(), 64, A.data_ptr(), 64, 128, 128, 64, b.data_ptr(), None, A.data_ptr(), 128, 129, 0, stream))


def test_gemm_repeatable(env):
    """Race screen: the same launch five times must be bit-identical (counted-vmcnt LDS-DMA pipeline)."""
    _lib, lib, dev, stream = env
    g = torch.Generator(device=dev).manual_seed(0)
    A = (torch.rand((50000, 768), device=dev, generator=g) * 2 - 1).half()
    W = ((torch.rand((2304, 768), device=dev, generator=g) * 2 - 1) * 0.07).half()
    bias = torch.rand(2304, device=dev, generator=g)
    outs = []
    for _ in range(5):
        out = torch.empty((50000, 2304), device=dev, dtype=torch.float16)
        _gemm(env, torch.float16, "bias", A, W, bias, None, out, 256)
        outs.append(out)
    assert all(torch.equal(outs[0], o) for o in outs[1:])


@pytest.mark.parametrize("dt,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.6e-2), (torch.float32, 2e-5)])
@pytest.mark.parametrize("rows,dim", [(1, 768), (37, 768), (1000, 1024), (50, 512),
                                      # the float32 narrow-row kernel (SAM2 dims): rows off every rows-per-wave multiple
                                      (1, 96), (65537, 96), (4099, 192), (4097, 384), (9, 256), (33, 128)])
def test_layernorm_vs_torch(env, dt, tol, rows, dim):
    _lib, lib, dev, stream = env
    g = torch.Generator(device=dev).manual_seed(rows + dim)
    x = torch.randn((rows, dim), device=dev, generator=g) * 3 + 1
    gamma = torch.rand(dim, device=dev, generator=g) + 0.5
    beta = torch.rand(dim, device=dev, generator=g) - 0.5
    out = torch.empty((rows, dim), device=dev, dtype=dt)
    _lib.check(lib.ap_layernorm(_lib.torch_dtype_code(dt), x.data_ptr(), dim, rows, dim, gamma.data_ptr(), beta.data_ptr(),
                                1e-6, out.data_ptr(), stream))
    torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm(x, (dim,), gamma, beta, 1e-6)
    assert (out.float() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item() / 2)


def _attn_ref(qkv, n, T, H, hd=64):
    q, k, v = qkv.float().view(n, T, 3, H, hd).permute(2, 0, 3, 1, 4)
    p = torch.softmax(q @ k.transpose(-1, -2) / (hd ** 0.5), -1)
    return (p @ v).permute(0, 2, 1, 3).reshape(n * T, H * hd)


@pytest.mark.parametrize("dt,tol", [(torch.float16, 4e-3), (torch.bfloat16, 3e-2), (torch.float32, 2e-5)])
@pytest.mark.parametrize("shape", [(3, 197, 12), (2, 50, 16), (1, 257, 12), (5, 1, 12), (2, 785, 12), (1, 1025, 4),
                                   (3, 785, 3), (1, 300, 13), (7, 64, 1),      # (image, head) counts off the XCD walk's 8
                                   (2, 150, 12), (1, 192, 5), (1, 400, 6), (2, 265, 24)])   # key-tile counts = 3 (mod 4): the
                                   # output staging wraps around the ring (round 4 fix); 265 = uni_v2
def test_attention_vs_torch(env, dt, tol, shape):
    """|out| <= 6: half an ulp of the output + the rounding of P to the MFMA operand type."""
    _lib, lib, dev, stream = env
    n, T, H = shape
    if dt == torch.float32 and T > 288:
        pytest.skip("float32 attention is the register-strip kernel (T <= 288)")
    g = torch.Generator(device=dev).manual_seed(T)
    qkv = (torch.randn((n * T, 3 * H * 64), device=dev, generator=g) * 1.5).to(dt)
    out = torch.full((n * T, H * 64), float("nan"), device=dev, dtype=dt)
    _lib.check(lib.ap_attention(_lib.torch_dtype_code(dt), qkv.data_ptr(), out.data_ptr(), n, T, H, 64, stream))
    torch.cuda.synchronize()
    assert (out.float() - _attn_ref(qkv, n, T, H)).abs().max().item() <= tol


@pytest.mark.parametrize("dt,tol", [(torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("shape", [(2, 197, 4), (1, 1370, 16), (3, 50, 2), (1, 150, 3), (2, 448, 5), (9, 1, 1), (1, 2000, 1), (2, 257, 16)])
@pytest.mark.parametrize("hd", [96, 128])
def test_attention_wide_heads_vs_torch(env, dt, tol, shape, hd):
    """The HD = 96 and HD = 128 instantiations of the tiled kernel (the 80-wide heads of vit_h_14 / Virchow are stored zero-padded
    to 96; 128 = DINOv3 ViT-7B): full-width random heads against torch, incl. 1370 tokens (vit_h_14 at 518 px), 257 (Virchow) and
    key-tile counts of every residue mod 4."""
    _lib, lib, dev, stream = env
    n, T, H = shape
    g = torch.Generator(device=dev).manual_seed(T + 7)
    qkv = (torch.randn((n * T, 3 * H * hd), device=dev, generator=g) * 1.2).to(dt)
    out = torch.full((n * T, H * hd), float("nan"), device=dev, dtype=dt)
    _lib.check(lib.ap_attention(_lib.torch_dtype_code(dt), qkv.data_ptr(), out.data_ptr(), n, T, H, hd, stream))
    torch.cuda.synchronize()
    assert (out.float() - _attn_ref(qkv, n, T, H, hd)).abs().max().item() <= tol


@pytest.mark.parametrize("dt,tol", [(torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("shape", [(2, 197, 12, 150), (1, 785, 12, 700), (1, 300, 4, 70)])
def test_attention_forced_rescale(env, dt, tol, shape):
    """The online-softmax rescale is deferred while the running max grows by < 2^8: force the branch with a
    key whose score against one query jumps far above everything seen before it (full-tensor reference)."""
    _lib, lib, dev, stream = env
    n, T, H, at = shape
    g = torch.Generator(device=dev).manual_seed(at)
    qkv = torch.randn((n * T, 3 * H * 64), device=dev, generator=g).to(dt)
    v = qkv.view(n, T, 3, H, 64)
    v[:, at, 1] = v[:, 5, 0] * 6.0
    v[:, at + 3, 1] *= 5.0
    out = torch.full((n * T, H * 64), float("nan"), device=dev, dtype=dt)
    _lib.check(lib.ap_attention(_lib.torch_dtype_code(dt), qkv.data_ptr(), out.data_ptr(), n, T, H, 64, stream))
    torch.cuda.synchronize()
    assert (out.float() - _attn_ref(qkv, n, T, H)).abs().max().item() <= tol


def test_operator_error_paths(env):
    _lib, lib, dev, stream = env
    x = torch.zeros(16, device=dev)
    assert lib.ap_gemm(1, 0, None, 0, None, 0, 1, 1, 1, None, None, None, 0, 0, 0, stream) == -1
    assert lib.ap_gemm(1, 7, x.data_ptr(), 128, x.data_ptr(), 128, 4, 128, 128, x.data_ptr(), None, x.data_ptr(), 128, 0, 0,
                       stream) == -1
    assert lib.ap_attention(1, x.data_ptr(), x.data_ptr(), 1, 4, 1, 32, stream) == -1     # head_dim not 64 / 96 / 128
    assert lib.ap_attention(0, x.data_ptr(), x.data_ptr(), 1, 4, 1, 96, stream) == -1     # 96-wide heads: f16 / bf16 only
    assert lib.ap_attention(0, x.data_ptr(), x.data_ptr(), 1, 4, 1, 128, stream) == -1    # 128-wide heads: f16 / bf16 only
    assert b"head_dim" in lib.ap_last_error()
    # fused-LayerNorm operators: missing operands, float32, bad kernel choice, shapes the persistent kernel cannot take
    h = torch.zeros((256, 256), device=dev, dtype=torch.float16)
    f = torch.zeros(4096, device=dev)
    args = (h.data_ptr(), 256, h.data_ptr(), 256, 256, 256, 256, f.data_ptr())
    assert lib.ap_gemm_fused(1, 4, *args, None, None, None, h.data_ptr(), 256, 0, stream) == -1          # NORM without statistics
    assert lib.ap_gemm_fused(1, 6, *args, None, None, None, h.data_ptr(), 256, 0, stream) == -1          # RESID without partial
    assert lib.ap_gemm_fused(0, 6, *args, None, None, f.data_ptr(), h.data_ptr(), 256, 0, stream) == -1  # float32
    assert lib.ap_gemm_fused(1, 9, *args, None, None, f.data_ptr(), h.data_ptr(), 256, 0, stream) == -1  # unknown epilogue
    assert lib.ap_gemm_fused(1, 6, *args, None, None, f.data_ptr(), h.data_ptr(), 256, 64, stream) == -1  # unknown impl
    assert lib.ap_gemm_fused(1, 6, h.data_ptr(), 256, h.data_ptr(), 256, 256, 128, 256, f.data_ptr(), None, None, f.data_ptr(),
                             h.data_ptr(), 256, 256, stream) == -1                                       # N = 128 on the 256 kernel
    assert lib.ap_rowstats_finalize(f.data_ptr(), 4, 3, 192, 1e-6, f.data_ptr(), stream) == -1           # odd group count
    assert lib.ap_sattention_f32(f.data_ptr(), 48, f.data_ptr(), 48, f.data_ptr(), 48, 1, 1, 4, 4, 48, 0.1, f.data_ptr(), 48,
                                 stream) == -1                                                            # d = 48
    assert b"sattention" in lib.ap_last_error()


def test_tile_content_counts_bit_exact_vs_cv2_restatement(env):
    """--no-fast-mode filters: device counts == the integer restatement of cv2's RGB2GRAY / RGB2HSV on the host."""
    from atlaspatch_amd.utils.image import tile_content_counts, tile_content_flags
    from oracle import cv2_restated as cv2r
    _lib, lib, dev, stream = env
    rng = np.random.default_rng(0)
    tiles = rng.integers(0, 256, (9, 256, 256, 3), dtype=np.uint8)
    tiles[1] = 0                                                    # black
    tiles[2] = 245                                                  # white
    tiles[3] = rng.integers(230, 256, (256, 256, 1), dtype=np.uint8)   # bright greys: S = 0
    tiles[4] = rng.integers(0, 60, (256, 256, 3), dtype=np.uint8)  # dark noise around the black threshold
    tiles[5, :, :, 0] = 250; tiles[5, :, :, 1] = rng.integers(236, 251, (256, 256)); tiles[5, :, :, 2] = 246   # S near threshold
    tiles[6, :180] = 255                                            # 70.3 % white rows
    tiles[7, :179] = 255                                            # 69.9 %
    d = torch.from_numpy(tiles).to(dev)
    for bt, st in ((50, 15), (40, 5), (1, 1), (255, 255)):
        got = tile_content_counts(d, black_thresh=bt, sat_thresh=st)
        want = np.zeros((9, 2), np.int64)
        for i, t in enumerate(tiles):
            want[i, 0] = int((cv2r.cvtColor_RGB2GRAY(t) < bt).sum())
            s, v = cv2r.cvtColor_RGB2HSV_sv(t)
            want[i, 1] = int(((s < st) & (v >= 200)).sum())
        assert np.array_equal(got, want), (bt, st)
    black, white = tile_content_flags(d, black_thresh=50, white_thresh=15)
    assert [bool(cv2r.is_black_patch(t, rgb_thresh=50)) for t in tiles] == black.tolist()
    assert [bool(cv2r.is_white_patch(t, sat_thresh=15)) for t in tiles] == white.tolist()
    assert black[1] and white[2] and white[6] and not white[7]


# ----------------------------------------------------------------------------- cv2.resize on the device
_CV2_CASES = [
    # (h, w) -> (oh, ow), interpolation, n
    ((512, 512), (256, 256), 1, 5),        # the 40x -> 20x tile: INTER_LINEAR at exactly 2 x 2 = 2 x 2 area average
    ((1024, 1024), (256, 256), 1, 3),      # 80x-ish read: plain non-antialiased bilinear
    ((300, 300), (256, 256), 1, 4), ((180, 200), (256, 256), 1, 4), ((511, 513), (256, 256), 1, 2),
    ((256, 256), (256, 256), 1, 2),        # dsize == ssize: copy
    ((768, 768), (256, 256), 3, 2), ((96, 64), (32, 16), 3, 3),                       # integer-ratio area (3x3, 3x4)
    ((411, 300), (100, 128), 3, 2), ((733, 1024), (699, 500), 3, 1), ((100, 100), (77, 33), 3, 3),   # general area
    ((100, 80), (200, 256), 3, 2),         # area when enlarging = bilinear with area-mode coordinates
    ((40, 60), (80, 30), 3, 2),            # one axis up, one down
    ((100, 80), (200, 256), 2, 2), ((300, 300), (256, 256), 2, 2), ((60, 75), (163, 201), 2, 2),   # cubic
    ((6250 // 4, 6250 // 4), (3125 // 4, 3125 // 4), 3, 1),    # thumbnail-shaped 2 x 2 area
]


@pytest.mark.parametrize("in_hw,out_hw,interp,n", _CV2_CASES)
def test_cv2_resize_device_equals_oracle_bit_for_bit(env, in_hw, out_hw, interp, n):
    """ap_cv2_resize_u8 == oracle/cv2_resize.py (OpenCV's 8-bit resize restated; parity unpinned against cv2 itself)
    for every mode the reference's path can take: feature_embedding.py:94-95 (INTER_LINEAR on tiles), iwsi.py:305-321
    (INTER_AREA / INTER_CUBIC thumbnails)."""
    from atlaspatch_amd.utils.resample import cv2_resize_device
    from oracle import cv2_resize as R
    _lib, lib, dev, stream = env
    rng = np.random.default_rng(in_hw[0] * 7 + out_hw[1] + interp)
    tiles = rng.integers(0, 256, (n, in_hw[0], in_hw[1], 3), dtype=np.uint8)
    got = cv2_resize_device(torch.from_numpy(tiles).to(dev), (out_hw[1], out_hw[0]), interp).cpu().numpy()
    for i in range(n):
        want = R.resize(tiles[i], (out_hw[1], out_hw[0]), interp)
        assert np.array_equal(got[i], want), (i, np.abs(got[i].astype(int) - want).max(), (got[i] != want).sum())
    if interp == 2:                          # OpenCV's scalar vertical pass, selectable
        got = cv2_resize_device(torch.from_numpy(tiles).to(dev), (out_hw[1], out_hw[0]), interp, flags=1).cpu().numpy()
        for i in range(n):
            assert np.array_equal(got[i], R.resize(tiles[i], (out_hw[1], out_hw[0]), interp, cubic_vertical="scalar"))


def test_thumbnail_at_power_resizes_on_device_like_the_oracle(tmp_path):
    """IWSI.get_thumbnail_at_power (iwsi.py:246-323) on a 40x synthetic slide with levels 1/4/16: level 2 is read in full
    and shrunk 2 x with INTER_AREA; on a single-level slide the level-0 image is shrunk by a non-integer factor."""
    import json
    from atlaspatch_amd.core.wsi.synth_wsi import SynthWSI
    from atlaspatch_amd.core.wsi.synth_pixels import SynthSpec, render_region
    from oracle import cv2_resize as R
    p = tmp_path / "t.synth"
    json.dump({"width": 16000, "height": 11200, "seed": 5, "mag": 40, "mpp": 0.25, "downsamples": [1, 4, 16]}, open(p, "w"))
    wsi = SynthWSI(str(p))
    thumb = np.asarray(wsi.get_thumbnail_at_power(power=1.25))
    spec = SynthSpec(width=16000, height=11200, seed=5, mag=40, mpp=0.25)
    level = render_region(spec, 0, 0, 1000, 700, 2)
    assert thumb.shape == (350, 500, 3)
    assert np.array_equal(thumb, R.resize(level, (500, 350), R.INTER_AREA))
    p2 = tmp_path / "u.synth"
    json.dump({"width": 3000, "height": 2100, "seed": 6, "mag": 5, "mpp": 2.0, "downsamples": [1]}, open(p2, "w"))
    wsi2 = SynthWSI(str(p2))
    thumb2 = np.asarray(wsi2.get_thumbnail_at_power(power=1.3))            # ds 3.846...: general area table
    spec2 = SynthSpec(width=3000, height=2100, seed=6, mag=5, mpp=2.0, downsamples=(1.0,))
    full = render_region(spec2, 0, 0, 3000, 2100, 0)
    ds = 5 / 1.3
    want = R.resize(full, (int(round(3000 / ds)), int(round(2100 / ds))), R.INTER_AREA)
    assert np.array_equal(thumb2, want)


# ----------------------------------------------------------------------------- exact-f32 MFMA GEMM of the SAM2 operator set
@pytest.mark.parametrize("case", [
    # batch, M, N, K, w_kn, act, bias, resid, alpha
    (1, 65536, 96, 147, False, 0, True, True, 1.0),        # Hiera patch embed: K = 147 (scalar-load path), + pos embed
    (1, 4096, 1152, 384, False, 0, True, False, 1.0),      # qkv of a stage-3 block
    (1, 1000, 384, 96, False, 1, True, False, 1.0),        # ragged M, GELU
    (1024, 64, 64, 96, False, 0, False, False, 0.1020620726),    # windowed q k^T, 64 x 64 tiles
    (1024, 64, 96, 64, True, 0, False, False, 1.0),        # windowed P V (NN weight)
    (3, 196, 49, 96, False, 0, False, False, 0.5),         # pooled-query window attention: N = 49
    (3, 49, 96, 196, True, 0, False, False, 1.0),
    (1, 9, 256, 2048, False, 2, True, True, 1.0),          # decoder MLP on 9 tokens, ReLU + residual
    (1, 65536, 1, 32, False, 0, False, False, 1.0),        # final hypernetwork product: N = 1
    (2, 130, 257, 70, True, 0, True, True, 2.0),           # odd everything, NN with ragged N (per-element loads)
    (8, 9, 16, 4096, True, 0, False, False, 1.0),          # token-to-image P V: tiny output, long K -> split-K
    (4, 4096, 96, 4096, True, 0, False, False, 1.0),       # global attention P V (heads as the batch, stride d)
    (1, 1024, 768, 3072, False, 0, True, True, 1.0),       # stage-4 fc2: 48 tiles of 128 -> 64-tiles + split-K, bias + residual
    (1, 70, 50, 1000, False, 1, True, False, 0.5),         # split-K with a ragged last chunk, GELU after the reduction
])
def test_sgemm_mfma_vs_torch(env, case):
    """ap_sgemm (v_mfma_f32_32x32x2_f32, exact f32 products, f32 accumulation) vs torch fp32 matmul on the same operands:
    only the summation order differs -> relative error of a few ulp * sqrt(K)."""
    import ctypes as C
    _lib, lib, dev, stream = env
    batch, M, N, K, w_kn, act, use_bias, use_resid, alpha = case
    g = torch.Generator(device=dev).manual_seed(M * 3 + N * 5 + K)
    A = torch.randn((batch, M, K), device=dev, generator=g)
    W = torch.randn((batch, K, N) if w_kn else (batch, N, K), device=dev, generator=g)
    bias = torch.randn(N, device=dev, generator=g) if use_bias else None
    resid = torch.randn((batch, M, N), device=dev, generator=g) if use_resid else None
    out = torch.full((batch, M, N), float("nan"), device=dev)
    _lib.check(lib.ap_sgemm(A.data_ptr(), K, M * K, W.data_ptr(), N if w_kn else K, N * K, 1 if w_kn else 0, batch, M, N, K,
                            C.c_float(alpha), bias.data_ptr() if use_bias else None, act,
                            resid.data_ptr() if use_resid else None, N, M * N, out.data_ptr(), N, M * N, stream), "ap_sgemm")
    torch.cuda.synchronize()
    ref = torch.matmul(A.double(), (W if w_kn else W.transpose(1, 2)).double()) * alpha
    if use_bias:
        ref = ref + bias.double()
    if act == 1:
        ref = torch.nn.functional.gelu(ref)
    elif act == 2:
        ref = torch.relu(ref)
    if use_resid:
        ref = ref + resid.double()
    err = (out.double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert torch.isfinite(out).all() and err <= 2e-6 * scale * max(1.0, K ** 0.5 / 4), (err, scale)


@pytest.mark.parametrize("case", [(1, 4, 4096, 4096, 96), (1, 2, 64, 128, 32), (3, 3, 160, 256, 64), (25, 4, 196, 196, 96),
                                  (25, 8, 49, 196, 96), (64, 1, 64, 64, 96), (100, 2, 16, 64, 96), (1, 8, 9, 9, 32),
                                  (7, 2, 4, 16, 96), (2, 1, 33, 1, 64)])
@pytest.mark.parametrize("form", ["ap_sattention_f32", "ap_sattention_split_f16"])
def test_sattention_f32_vs_torch(env, case, form):
    """ap_sattention_f32 (fused exact-f32 MFMA attention of the SAM2 trunk: image-wide blocks and batched windows, ragged
    query / key counts, key tiles split over four waves and merged) and its split-f16 form (same arguments; both products as
    three f16 MFMA passes on hi / lo halves) against softmax(q k^T scale) v in float64, the SAME bound; repeatable."""
    import math
    _lib, lib, dev, stream = env
    nb, heads, tq, tk, d = case
    g = torch.Generator(device=dev).manual_seed(nb * 7 + tq + tk + d)
    ld = 3 * heads * d
    qkv_q = torch.randn((nb * tq, ld), device=dev, generator=g) * 1.3
    qkv_k = torch.randn((nb * tk, ld), device=dev, generator=g) * 1.3
    q, k, v = qkv_q[:, :heads * d], qkv_k[:, heads * d:2 * heads * d], qkv_k[:, 2 * heads * d:]
    scale = 1.0 / math.sqrt(d)
    outs = []
    for _ in range(3):
        o = torch.full((nb * tq, heads * d), float("nan"), device=dev)
        _lib.check(getattr(lib, form)(q.data_ptr(), ld, k.data_ptr(), ld, v.data_ptr(), ld, nb, heads, tq, tk, d, scale,
                                      o.data_ptr(), heads * d, stream), form)
        torch.cuda.synchronize()
        outs.append(o)
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    qh = q.reshape(nb, tq, heads, d).permute(0, 2, 1, 3).double()
    kh = k.reshape(nb, tk, heads, d).permute(0, 2, 1, 3).double()
    vh = v.reshape(nb, tk, heads, d).permute(0, 2, 1, 3).double()
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1) @ vh).permute(0, 2, 1, 3).reshape(nb * tq, heads * d).float()
    err = (outs[0] - ref).abs().max().item()
    print(f"SATTENTION {form} {case}: max abs err {err:.2e}")
    assert err <= 2e-5


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16, torch.float32])
def test_gelu_epilogue_deviation_from_erf_is_isolated_and_bounded(env, dt):
    """The f16 / bf16 GEMM epilogue evaluates GELU as x * sigmoid(x * (c0 + c1 t + c2 t^2)), t = min(x^2, 50) (a minimax fit of the
    erf form, ap_common.h), the float32 mode calls erff.  Isolated here with an identity weight: out = gelu(x) for x swept over
    [-9, 9]; against nn.GELU() (erf) in float64 the error is the fit's 2.6e-5 plus half an ulp of the output type."""
    _lib, lib, dev, stream = env
    K = N = 256
    M = 4096
    x = torch.linspace(-9.0, 9.0, M * K, device=dev, dtype=torch.float64).reshape(M, K)
    A = x.to(dt).contiguous()
    W = torch.eye(N, K, device=dev, dtype=dt).contiguous()
    bias = torch.zeros(N, device=dev, dtype=torch.float32)
    out = torch.empty((M, N), device=dev, dtype=dt)
    for impl in ((256, 128) if dt != torch.float32 else (128,)):
        _gemm(env, dt, "gelu", A, W, bias, None, out, impl)
        xin = A.double()
        want = torch.nn.functional.gelu(xin)
        err = (out.double() - want).abs()
        half_ulp = {torch.float16: 2.0 ** -11, torch.bfloat16: 2.0 ** -8, torch.float32: 2.0 ** -23}[dt]
        bound = 3e-5 + half_ulp * want.abs()
        assert (err <= bound).all(), (impl, float(err.max()), float((err - bound).max()))
        assert float(err.max()) > 0 or dt == torch.float32


def test_cv2_resize_device_equals_oracle_on_random_shapes(env):
    """60 seeded random (source size, destination size, interpolation) triples, 1 .. 300 pixels per side, including one-pixel
    sources and destinations, extreme ratios and mixed up / down scaling: the library's C++ table builder and kernels against
    the NumPy restatement, bit for bit (both vertical-pass variants for INTER_CUBIC)."""
    from atlaspatch_amd.utils.resample import cv2_resize_device
    from oracle import cv2_resize as R
    _lib, lib, dev, stream = env
    rng = np.random.default_rng(2024)
    sizes = [1, 2, 3, 5, 8, 17, 31, 64, 100, 127, 256, 300]
    for trial in range(60):
        h, w, oh, ow = (int(rng.choice(sizes)) for _ in range(4))
        interp = int(rng.choice([1, 2, 3]))
        tiles = rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)
        for flags, mode in ((0, "sse"), (1, "scalar")):
            if interp != 2 and flags:
                continue
            got = cv2_resize_device(torch.from_numpy(tiles).to(dev), (ow, oh), interp, flags=flags).cpu().numpy()
            for i in range(2):
                want = R.resize(tiles[i], (ow, oh), interp, cubic_vertical=mode)
                assert np.array_equal(got[i], want), (trial, (h, w), (oh, ow), interp, mode,
                                                      int(np.abs(got[i].astype(int) - want).max()))
    # empty batch: nothing launched, nothing touched
    empty = torch.empty((0, 8, 8, 3), dtype=torch.uint8, device=dev)
    assert cv2_resize_device(empty, (4, 4), 1).shape == (0, 4, 4, 3)


def test_cv2_resize_exact_halving_quad_kernel(env):
    """The exact 2 x 2 shrink with four output pixels per thread (the 40x -> 20x tile path; taken when the output width
    is a multiple of 4): batches of 1 .. 7 tiles, square and non-square, INTER_LINEAR (which OpenCV re-routes to the
    2 x 2 average) and INTER_AREA, against the NumPy restatement bit for bit; an output width off the multiple of 4 runs
    the one-pixel-per-thread kernel and must agree too."""
    from atlaspatch_amd.utils.resample import cv2_resize_device
    from oracle import cv2_resize as R
    _lib, lib, dev, stream = env
    rng = np.random.default_rng(77)
    for (n, oh, ow) in ((1, 4, 4), (3, 20, 32), (5, 256, 256), (7, 52, 100), (2, 33, 8), (2, 16, 6)):
        tiles = rng.integers(0, 256, (n, 2 * oh, 2 * ow, 3), dtype=np.uint8)
        tiles[0, ::2] = 255                                           # rounding at .5: (255 + 255 + x + y + 2) >> 2
        for interp in (1, 3):
            got = cv2_resize_device(torch.from_numpy(tiles).to(dev), (ow, oh), interp).cpu().numpy()
            for i in range(n):
                assert np.array_equal(got[i], R.resize(tiles[i], (ow, oh), interp)), (n, oh, ow, interp, i)


def test_cv2_resize_takes_unaligned_pointers_and_a_bounded_table_cache(env):
    """The C ABI accepts any pointer: source / destination buffers that start 1 .. 3 bytes off an aligned address run the
    one-pixel-per-thread kernels and agree with the aligned launch bit for bit (the vector kernels are only dispatched on
    aligned pointers).  And the per-shape table cache is bounded: 80 distinct shapes (more than it holds) in a row, then the
    first again, all equal the NumPy restatement."""
    from oracle import cv2_resize as R
    _lib, lib, dev, stream = env
    rng = np.random.default_rng(5)
    n, oh, ow = 3, 20, 32
    tiles = rng.integers(0, 256, (n, 2 * oh, 2 * ow, 3), dtype=np.uint8)
    want = np.stack([R.resize(t, (ow, oh), 3) for t in tiles])
    flat = torch.from_numpy(tiles.reshape(-1)).to(dev)
    for so, do in ((0, 0), (1, 0), (0, 1), (3, 2)):
        src = torch.zeros(flat.numel() + 16, dtype=torch.uint8, device=dev)
        src[so:so + flat.numel()] = flat
        dst = torch.zeros(n * oh * ow * 3 + 16, dtype=torch.uint8, device=dev)
        _lib.check(lib.ap_cv2_resize_u8(src.data_ptr() + so, n, 2 * oh, 2 * ow, dst.data_ptr() + do, oh, ow, 3, 0, stream), "resize")
        torch.cuda.synchronize()
        got = dst[do:do + n * oh * ow * 3].cpu().numpy().reshape(n, oh, ow, 3)
        assert np.array_equal(got, want), (so, do)
    from atlaspatch_amd.utils.resample import cv2_resize_device
    shapes = [(17 + i, 23 + (i * 7) % 31, 9 + i % 13, 11 + (i * 3) % 17) for i in range(80)]
    for (h, w, oh2, ow2) in shapes + shapes[:1]:
        img = rng.integers(0, 256, (1, h, w, 3), dtype=np.uint8)
        got = cv2_resize_device(torch.from_numpy(img).to(dev), (ow2, oh2), 3).cpu().numpy()[0]
        assert np.array_equal(got, R.resize(img[0], (ow2, oh2), 3)), (h, w, oh2, ow2)


def test_vit_set_params_equals_per_tensor_uploads_and_rejects_bad_entries(env):
    """``ap_vit_set_params`` (a whole checkpoint in one native call, pipelined through pinned staging) gives the same
    encoder as one ``ap_vit_set_param`` per tensor -- bit-identical features -- and fails loudly, with the parameter's name,
    on an unknown name or a wrong element count."""
    import ctypes as C
    from atlaspatch_amd.encoders.vit import ARCHS, HipViT, random_canonical_state_dict
    _lib, lib, dev, stream = env
    arch = dict(ARCHS["vit_b_16"], depth=2)
    state = random_canonical_state_dict(arch, 3)
    tiles = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (5, 256, 256, 3), dtype=np.uint8)).to(dev)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    batched = HipViT(arch, state, device=dev, dtype=torch.float16)           # uses ap_vit_set_params
    a = torch.empty((5, 768), dtype=torch.float32, device=dev)
    batched.forward_u8(tiles, mean, std, a)
    single = HipViT.__new__(HipViT)                                          # the same object built tensor by tensor
    single.lib, single.device, single.dtype, single.arch, single.embed_dim, single._workspace = lib, dev, torch.float16, arch, 768, None
    cfg = _lib.VitConfig(arch["image_size"], arch["patch_size"], arch["dim"], arch["depth"], arch["heads"], arch["mlp_dim"],
                         float(arch["ln_eps"]), 0, _lib.AP_F16, 0, 0, 0, 1e-5)
    h = C.c_void_p()
    _lib.check(lib.ap_vit_create(C.byref(cfg), C.byref(h)), "create")
    single._handle = h
    for name, t in state.items():
        arr = np.ascontiguousarray(t.float().numpy())
        _lib.check(lib.ap_vit_set_param(h, name.encode(), arr.ctypes.data_as(C.c_void_p), arr.size), name)
    _lib.check(lib.ap_vit_finalize(h), "finalize")
    b = torch.empty((5, 768), dtype=torch.float32, device=dev)
    single.forward_u8(tiles, mean, std, b)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    # error paths of the batched call
    arr = np.zeros(7, np.float32)
    names = (C.c_char_p * 1)(b"no.such.parameter")
    ptrs = (C.c_void_p * 1)(arr.ctypes.data)
    counts = (C.c_size_t * 1)(7)
    assert lib.ap_vit_set_params(h, names, ptrs, counts, 1) != 0 and b"no.such.parameter" in lib.ap_last_error()
    first = next(iter(state))
    names = (C.c_char_p * 1)(first.encode())
    assert lib.ap_vit_set_params(h, names, ptrs, counts, 1) != 0 and first.encode() in lib.ap_last_error()
    batched.release(); single.release()


def test_vit_create_rejects_configurations_the_kernels_cannot_run():
    """ap_vit_create validates the ABI-v18 fields: every rejected configuration names itself in ap_last_error and leaves no
    handle behind."""
    import ctypes as C
    from atlaspatch_amd import _lib
    lib = _lib.load()
    base = dict(image_size=224, patch_size=16, dim=768, depth=2, heads=12, mlp_dim=3072, ln_eps=1e-6, layer_scale=0, compute_dtype=1,
                pool=0, pool_dim=0, pool_heads=0, pool_ln_eps=1e-5, reg_tokens=0, no_embed_class=0, mlp_type=0, head_dim=0,
                attn_scale=0.0, pre_norm=0, act=0, proj_dim=0, rope=0)

    def create(**kw):
        cfg = _lib.VitConfig(*[{**base, **kw}[name] for name, _ in _lib.VitConfig._fields_ if name != "struct_size"])
        h = C.c_void_p()
        rc = lib.ap_vit_create(C.byref(cfg), C.byref(h))
        if rc == 0:
            lib.ap_vit_destroy(h)
        return rc, lib.ap_last_error().decode()

    assert create()[0] == 0
    assert create(pool=2)[0] == 0 and create(pre_norm=1, act=1, proj_dim=512)[0] == 0 and create(rope=1)[0] == 0
    for kw, word in ((dict(pool=3), "pool"), (dict(proj_dim=500), "proj_dim"), (dict(proj_dim=1024), "proj_dim"),
                     (dict(proj_dim=512, pool=2), "proj_dim"), (dict(act=1, mlp_type=1), "act"), (dict(act=7), "act"),
                     (dict(rope=1, dim=1280, heads=16, mlp_dim=5120, head_dim=128), "rope")):
        rc, msg = create(**kw)
        assert rc != 0 and word in msg, (kw, rc, msg)


def test_clock_probe_reports_a_plausible_shader_clock():
    """ap_clock_probe (bench.py's clock line): s_memtime / s_memrealtime stamps per compute unit around a few GEMM-sized launches give
    a shader clock inside the part's range on every XCD."""
    from atlaspatch_amd.utils.telemetry import ClockProbe, PowerSampler
    dev = torch.device("cuda:0")
    probe = ClockProbe(dev)
    a = torch.randn(4096, 4096, device=dev, dtype=torch.float16)
    a = (a @ a).clamp_(-1, 1)                    # first use of the GEMM library: hundreds of idle milliseconds (an idle chip clocks
    torch.cuda.synchronize(dev)                  # at tens of MHz, and the probe reports the AVERAGE over its region)
    with PowerSampler(dev) as ps:
        probe.start()
        for _ in range(20):
            a = (a @ a).clamp_(-1, 1)
        probe.stop()
        torch.cuda.synchronize(dev)
    info = probe.read()
    # stamps are differenced per compute unit (their s_memtime counters are not aligned with each other): every XCD's median
    # and the 5 .. 95 % band over the units lie in the part's clock range
    assert info["xcds"] == 8 and info["compute_units"] >= 64, info
    assert 0.5 <= info["min_GHz"] and info["max_GHz"] <= 2.6 and 0.5 <= info["shader_clock_GHz"] <= 2.6, info
    assert 0.4 <= info["cu_spread_GHz"][0] and info["cu_spread_GHz"][1] <= 2.7, info
    s = ps.summary()
    assert s["samples"] == 0 or 20.0 <= s["package_W_mean"] <= 2000.0, s
