"""GPU parity tests: HIP path (through the C ABI) vs the CPU oracle and the golden fixtures."""
import numpy as np
import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


# ----------------------------------------------------------------------------- K1
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_preproc_chw_bit_exact(dtype):
    import ctypes as C
    from atlaspatch_amd import _lib
    from oracle import vit_oracle
    lib = _lib.load()
    rng = np.random.default_rng(3)
    src = rng.integers(0, 256, (5, 256, 256, 3), dtype=np.uint8)
    ref = vit_oracle.preprocess_center_crop(src).to(dtype)
    d_src = torch.from_numpy(src).to(_dev())
    d_out = torch.empty((5, 3, 224, 224), dtype=dtype, device=_dev())
    _lib.check(lib.ap_preproc_u8hwc_to_chw(d_src.data_ptr(), 5, 256, 256, 16, 16, 224, 224, _lib.f3(MEAN),
                                           _lib.f3(STD), d_out.data_ptr(), _lib.torch_dtype_code(dtype),
                                           _lib.current_stream_ptr()))
    torch.cuda.synchronize()
    got = d_out.cpu()
    assert torch.equal(got.view(torch.int16 if dtype != torch.float32 else torch.int32),
                       ref.view(torch.int16 if dtype != torch.float32 else torch.int32))


@pytest.mark.parametrize("seed", range(12))
def test_preproc_random_geometry_bit_exact(seed):
    """K1 on random tile sizes, crop windows, batch sizes and normalisation constants, both output layouts (CHW and
    patch rows), all three output types: bit-identical to ((x / 255) - mean) / std in float32, rounded once."""
    from atlaspatch_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(900 + seed)
    dtype = [torch.float32, torch.float16, torch.bfloat16][seed % 3]
    crop = int(rng.choice([224, 448, 112, 64]))
    h, w = crop + int(rng.integers(0, 70)), crop + int(rng.integers(0, 70))
    top, left = int(rng.integers(0, h - crop + 1)), int(rng.integers(0, w - crop + 1))
    n = int(rng.integers(1, 7))
    mean = [float(v) for v in (rng.uniform(0.3, 0.6, 3) if seed % 2 else MEAN)]
    std = [float(v) for v in (rng.uniform(0.15, 0.35, 3) if seed % 2 else STD)]
    src = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    x = torch.from_numpy(src)[:, top:top + crop, left:left + crop, :].permute(0, 3, 1, 2).to(torch.float32).div(255)
    ref = x.sub(torch.tensor(mean, dtype=torch.float32).view(1, 3, 1, 1)).div(torch.tensor(std, dtype=torch.float32).view(1, 3, 1, 1))
    bits = torch.int32 if dtype == torch.float32 else torch.int16
    d_src = torch.from_numpy(src).to(_dev())
    d_out = torch.empty((n, 3, crop, crop), dtype=dtype, device=_dev())
    _lib.check(lib.ap_preproc_u8hwc_to_chw(d_src.data_ptr(), n, h, w, top, left, crop, crop, _lib.f3(mean), _lib.f3(std),
                                           d_out.data_ptr(), _lib.torch_dtype_code(dtype), _lib.current_stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(d_out.cpu().view(bits), ref.to(dtype).view(bits))
    g = crop // 16
    rows = ref.unfold(2, 16, 16).unfold(3, 16, 16).permute(0, 2, 3, 1, 4, 5).reshape(n * g * g, 768).contiguous().to(dtype)
    d_rows = torch.empty((n * g * g, 768), dtype=dtype, device=_dev())
    _lib.check(lib.ap_preproc_u8hwc_to_patchrows(d_src.data_ptr(), n, h, w, top, left, crop, crop, 16, _lib.f3(mean), _lib.f3(std),
                                                 d_rows.data_ptr(), 768, _lib.torch_dtype_code(dtype), _lib.current_stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(d_rows.cpu().view(bits), rows.view(bits))


def test_preproc_golden_vector(golden_dir):
    import os
    from atlaspatch_amd import _lib
    lib = _lib.load()
    g = np.load(os.path.join(golden_dir, "extract_batch.npz"))
    d_src = torch.from_numpy(g["preproc_in"][None]).to(_dev())
    d_out = torch.empty((1, 3, 224, 224), dtype=torch.float32, device=_dev())
    _lib.check(lib.ap_preproc_u8hwc_to_chw(d_src.data_ptr(), 1, 256, 256, 16, 16, 224, 224, _lib.f3(MEAN),
                                           _lib.f3(STD), d_out.data_ptr(), _lib.AP_F32, _lib.current_stream_ptr()))
    assert np.array_equal(d_out.cpu().numpy()[0], g["preproc_out"])


def test_preproc_patchrows_matches_unfold():
    from atlaspatch_amd import _lib
    from oracle import vit_oracle
    lib = _lib.load()
    rng = np.random.default_rng(4)
    src = rng.integers(0, 256, (3, 256, 256, 3), dtype=np.uint8)
    ref = vit_oracle.preprocess_center_crop(src)                       # [n,3,224,224]
    rows = ref.unfold(2, 16, 16).unfold(3, 16, 16)                     # [n,3,14,14,16,16]
    rows = rows.permute(0, 2, 3, 1, 4, 5).reshape(3 * 196, 768).contiguous()
    d_src = torch.from_numpy(src).to(_dev())
    d_out = torch.empty((3 * 196, 768), dtype=torch.float32, device=_dev())
    _lib.check(lib.ap_preproc_u8hwc_to_patchrows(d_src.data_ptr(), 3, 256, 256, 16, 16, 224, 224, 16,
                                                 _lib.f3(MEAN), _lib.f3(STD), d_out.data_ptr(), 768,
                                                 _lib.AP_F32, _lib.current_stream_ptr()))
    assert torch.equal(d_out.cpu(), rows)


# ----------------------------------------------------------------------------- ViT
def _hf_extractor(layers, dtype, **kw):
    from atlaspatch_amd.encoders.vit import build_hip_vit_extractor
    from oracle import vit_oracle
    model = vit_oracle.make_hf_vit(layers=layers)
    sd = dict(model.state_dict())
    ex = build_hip_vit_extractor(name=f"hfvit_L{layers}", arch="vit_b_16", depth=layers, state_dict=sd,
                                 device=_dev(), dtype=dtype, source="hf", **kw)
    return ex, sd


# tolerance: features vs the CPU fp32 path.  Three statistics:
#   norm-wise      ||a - b||_F / ||b||_F
#   element-wise   max_ij |a_ij - b_ij| / (|b_ij| + 0.05 max|b|)    (elements below 5 % of the feature scale are compared
#                                                                    on that absolute scale)
#   q99.9          the 99.9th percentile of the same per-element statistic
# Bounds are set PER CASE from what was measured on MI355X (round 3, profiles/r03_parity_lines.txt: every test is seeded
# and the device path is bit-repeatable, so a case reproduces its numbers), with 1.2x headroom on the norm-wise figure,
# 1.25x on q99.9 and 1.5x on the element-wise maximum (one element of ~25 000: heavy tailed, moves by 20-30 % between two
# equally accurate builds of the attention row sums).  float32 additionally keeps the north star's 1e-3 on every statistic
# (exact-f32 MFMA measures ~2e-6 / ~3e-5), and the 16-bit modes must also sit inside the REFERENCE's own 16-bit error
# (G1b / G1c envelope tests below).  A case without an entry falls back to the widest measured bound of its type.
MEASURED = {
    # (case, dtype): (norm-wise, element-wise max, q99.9)
    ("G1 L2", "float32"): (1.54e-6, 2.36e-5, 1.57e-5),
    ("G1 L2", "float16"): (6.33e-4, 1.13e-2, 6.93e-3),
    ("G1 L2", "bfloat16"): (4.81e-3, 9.61e-2, 5.52e-2),
    ("vit_b_16 L12 vs reference golden", "float32"): (2.11e-6, 2.70e-5, 2.05e-5),
    ("vit_b_16 L12 vs oracle", "float32"): (2.03e-6, 3.43e-5, 2.31e-5),
    ("vit_b_16 L12 vs reference golden", "float16"): (7.72e-4, 1.59e-2, 8.86e-3),
    ("vit_b_16 L12 vs oracle", "float16"): (8.01e-4, 1.60e-2, 8.72e-3),
    ("vit_b_16 L12 vs oracle, plain 16-bit stream", "float16"): (1.256e-3, 1.617e-2, 1.260e-2),
    ("vit_b_16 L12 vs reference golden, plain 16-bit stream", "float16"): (1.260e-3, 1.884e-2, 1.311e-2),
    ("vit_b_16 L12 vs oracle, f32_stream", "float16"): (8.71e-4, 1.421e-2, 9.04e-3),
    ("vit_b_16 L12 vs reference golden, f32_stream", "float16"): (8.51e-4, 1.268e-2, 9.94e-3),
    ("massive activations, fused", "float16"): (2.77e-4, 1.16e-3, 8.99e-4),
    ("massive activations, f32_stream", "float16"): (2.46e-4, 1.15e-3, 8.2e-4),
    ("massive activations, fused", "bfloat16"): (1.98e-3, 8.11e-3, 7.02e-3),
    ("massive activations, f32_stream", "bfloat16"): (1.94e-3, 7.84e-3, 6.45e-3),
    ("uni_v1 L24", "float16"): (8.38e-4, 1.25e-2, 9.10e-3),
    ("uni_v1 L24, f32_stream", "float16"): (9.39e-4, 1.441e-2, 9.97e-3),
    ("uni_v1 L24", "float32"): (2.36e-6, 2.89e-5, 2.18e-5),
    # round 6 (GELU clamp as the packed multiply's CLAMP bit: last-bit changes of the f16 GELU outputs).  Over the 81 float16 cases of
    # this file and test_encoder_zoo.py the new / old ratios have geometric means 0.998 (norm-wise), 1.011 (element-wise max), 0.999
    # (q99.9): an equally accurate build; this case's single worst element moved 1.27e-2 -> 1.95e-2 (profiles/r06i_parity_lines.txt)
    ("vit_l_16 L24", "float16"): (8.60e-4, 1.95e-2, 8.95e-3),
    ("vit_l_16 L24, f32_stream", "float16"): (9.24e-4, 1.334e-2, 1.048e-2),
    ("conch_v1 L12 @448", "float16"): (6.54e-4, 9.79e-3, 7.86e-3),
}
HEADROOM = (1.2, 1.5, 1.25)
FALLBACK = {"float32": (2.4e-6, 3.5e-5, 2.4e-5), "float16": (1.6e-3, 3.2e-2, 1.7e-2), "bfloat16": (1.03e-2, 9e-2, 5.8e-2)}
NORTH_STAR_F32 = 1e-3


def _elem(a, b, floor=0.05):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float((np.abs(a - b) / (np.abs(b) + floor * np.abs(b).max())).max())


def _elem_q(a, b, q=0.999, floor=0.05):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.quantile(np.abs(a - b) / (np.abs(b) + floor * np.abs(b).max()), q))


def _bounds(what, dtype):
    name = str(dtype).split(".")[-1]
    key = what.split(" n=")[0]
    measured = MEASURED.get((key, name), FALLBACK[name])
    return tuple(m * h for m, h in zip(measured, HEADROOM))


def _check(got, want, dtype, what=""):
    r, e, eq = _rel(got, want), _elem(got, want), _elem_q(got, want)
    br, be, bq = _bounds(what, dtype)
    print(f"PARITY {what} {str(dtype).split('.')[-1]}: norm-wise {r:.3e} (bound {br:.2e}) element-wise max {e:.3e} ({be:.2e}) "
          f"q99.9 {eq:.3e} ({bq:.2e})")
    assert r <= br and e <= be and eq <= bq, (what, dtype, (r, br), (e, be), (eq, bq))
    if dtype == torch.float32:
        assert max(r, e, eq) <= NORTH_STAR_F32
    return r


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_extract_batch_matches_reference_golden(dtype, golden_dir):
    """G1: outputs of the REFERENCE's extract_batch (HF ViT-B/16-shaped, 2 layers)."""
    import os
    g = np.load(os.path.join(golden_dir, "extract_batch.npz"))
    ex, _ = _hf_extractor(2, dtype)
    ns = (0, 1, 5, 32, 33)
    patches = helpers.golden_patches(ns)
    for n in ns:
        out = ex.extract_batch(patches[n], batch_size=32)
        assert out.dtype == np.float32 and out.shape == (n, 768) and out.flags.c_contiguous
        if n:
            _check(out, g[f"L2_n{n}_out"], dtype, f"G1 L2 n={n}")
    ex.cleanup()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_vit_b16_full_depth_vs_golden_and_oracle(dtype, golden_dir):
    import os
    from oracle import vit_oracle
    g = np.load(os.path.join(golden_dir, "extract_batch.npz"))
    ex, sd = _hf_extractor(12, dtype)
    patches = helpers.golden_patches((5,))[5]
    out = ex.extract_batch(patches, batch_size=32)
    _check(out, g["L12_n5_out"], dtype, "vit_b_16 L12 vs reference golden")
    rng = np.random.default_rng(11)
    more = [rng.integers(0, 256, (256, 256, 3), dtype=np.uint8) for _ in range(19)]
    want = vit_oracle.extract_batch(sd, more, heads=12, batch_size=32)
    got = ex.extract_batch(more, batch_size=7)        # chunking must not matter
    _check(got, want, dtype, "vit_b_16 L12 vs oracle")
    if dtype != torch.float32:                        # the f32-residual-stream dataflow stays available and is closer
        ex.vit.set_option("f32_stream", True)
        _check(ex.extract_batch(more, batch_size=7), want, dtype, "vit_b_16 L12 vs oracle, f32_stream")
        _check(ex.extract_batch(patches, batch_size=32), g["L12_n5_out"], dtype, "vit_b_16 L12 vs reference golden, f32_stream")
    ex.cleanup()


def test_float32_fast_mode_meets_1e_3_on_every_element(golden_dir):
    """`--feature-precision float32` (cli.py:175-181, models/patch/base.py:95-106) runs the split-f16 products by default
    (AP_VIT_OPT_SPLIT_F16: float32 buffers / LayerNorm / softmax / stream, GEMM products as three f16 MFMA passes on hi / lo
    halves).  The north star's tolerance -- 1e-3 relative, float32 -- is asserted here as a FIXED bound on the element-wise
    maximum (and so on every weaker statistic), against the reference's own float32 features (G1, depth 12) and the fp32
    CPU oracle; the exact f32 MFMA chain (option off) stays available and meets it too; the two agree to 1e-4 element-wise;
    the option toggles cleanly, forwards are bit-repeatable and independent of the batch cut."""
    import os
    from oracle import vit_oracle
    g = np.load(os.path.join(golden_dir, "extract_batch.npz"))
    ex, sd = _hf_extractor(12, torch.float32)
    patches = helpers.golden_patches((5,))[5]
    rng = np.random.default_rng(11)
    more = [rng.integers(0, 256, (256, 256, 3), dtype=np.uint8) for _ in range(19)]
    want = vit_oracle.extract_batch(sd, more, heads=12, batch_size=32)
    split_g = ex.extract_batch(patches, batch_size=32)
    split_o = ex.extract_batch(more, batch_size=32)
    assert np.array_equal(split_o, ex.extract_batch(more, batch_size=7))
    assert np.array_equal(split_o, ex.extract_batch(more, batch_size=32))
    ex.vit.set_option("split_f16", False)
    exact_g = ex.extract_batch(patches, batch_size=32)
    exact_o = ex.extract_batch(more, batch_size=32)
    ex.vit.set_option("split_f16", True)
    assert np.array_equal(split_o, ex.extract_batch(more, batch_size=32))
    ex.cleanup()
    assert not np.array_equal(split_o, exact_o)                     # the option does select another arithmetic
    for tag, got, ref in (("split vs reference golden", split_g, g["L12_n5_out"]), ("split vs oracle", split_o, want),
                          ("exact vs reference golden", exact_g, g["L12_n5_out"]), ("exact vs oracle", exact_o, want)):
        r, e, q = _rel(got, ref), _elem(got, ref), _elem_q(got, ref)
        print(f"PARITY float32 {tag}: norm-wise {r:.3e} element-wise max {e:.3e} q99.9 {q:.3e}")
        assert max(r, e, q) <= NORTH_STAR_F32, (tag, r, e, q)
        # plain element-wise relative error on the elements that are not near zero (> 1 % of the feature scale)
        big = np.abs(ref) > 0.01 * np.abs(ref).max()
        assert float((np.abs(got - ref)[big] / np.abs(ref)[big]).max()) <= NORTH_STAR_F32
    assert _elem(split_o, exact_o) <= 1e-4


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_exact_class_rows_meet_the_north_star_in_float16(dtype, golden_dir):
    """Option exact_cls (default on, ABI v19): the class rows' residual stream is also carried in float32.  Against the reference's
    own float32 features (G1, depth 12) and the fp32 oracle the default must (a) beat the plain 16-bit stream (option off =
    rounds 2-3's dataflow, whose measured bounds it must still reproduce), (b) in float16 meet the north star's 1e-3
    norm-wise, (c) not depend on the batch cut, (d) agree with the full-last-block form."""
    import os
    from oracle import vit_oracle
    g = np.load(os.path.join(golden_dir, "extract_batch.npz"))
    ex, sd = _hf_extractor(12, dtype)
    patches = helpers.golden_patches((5,))[5]
    rng = np.random.default_rng(11)
    more = [rng.integers(0, 256, (256, 256, 3), dtype=np.uint8) for _ in range(19)]
    want = vit_oracle.extract_batch(sd, more, heads=12, batch_size=32)
    on_g = ex.extract_batch(patches, batch_size=32)
    on_o = ex.extract_batch(more, batch_size=32)
    assert np.array_equal(on_o, ex.extract_batch(more, batch_size=7))
    ex.vit.set_option("full_last_block", True)
    full = ex.extract_batch(more, batch_size=32)
    ex.vit.set_option("full_last_block", False)
    ex.vit.set_option("exact_cls", False)
    off_g = ex.extract_batch(patches, batch_size=32)
    off_o = ex.extract_batch(more, batch_size=32)
    ex.vit.set_option("exact_cls", True)
    assert np.array_equal(on_o, ex.extract_batch(more, batch_size=32))          # the option toggles cleanly
    ex.cleanup()
    if dtype == torch.float16:                      # option off = the dataflow whose bounds rounds 2-3 measured
        _check(off_o, want, dtype, "vit_b_16 L12 vs oracle, plain 16-bit stream")
        _check(off_g, g["L12_n5_out"], dtype, "vit_b_16 L12 vs reference golden, plain 16-bit stream")
    r_on_o, r_off_o, r_on_g, r_off_g = _rel(on_o, want), _rel(off_o, want), _rel(on_g, g["L12_n5_out"]), _rel(off_g, g["L12_n5_out"])
    print(f"PARITY exact_cls {str(dtype).split('.')[-1]}: vs oracle {r_on_o:.3e} (plain {r_off_o:.3e}), vs reference golden {r_on_g:.3e} "
          f"(plain {r_off_g:.3e}), full last block vs tail {_rel(full, on_o):.3e}")
    assert r_on_o < 0.75 * r_off_o and r_on_g < 0.75 * r_off_g
    if dtype == torch.float16:
        assert r_on_o <= 1e-3 and r_on_g <= 1e-3
    assert _rel(full, on_o) <= (3e-4 if dtype == torch.float16 else 3e-3)


@pytest.mark.parametrize("dtype,tag", [(torch.float16, "f16"), (torch.bfloat16, "bf16")])
def test_lowp_error_not_above_the_references_own(dtype, tag, golden_dir):
    """G1b: the reference run in float16 / bfloat16 converts the whole module (model.to(dtype), models/patch/base.py:66),
    residual stream and LayerNorm I/O included.  Its own distance from its float32 features on the depth-12 model is
    the envelope: the build's 16-bit modes (fused-LayerNorm dataflow by default) must sit inside it."""
    import os
    g = np.load(os.path.join(golden_dir, "extract_batch.npz"))
    lp = np.load(os.path.join(golden_dir, "extract_batch_lowp.npz"))
    want32 = g["L12_n5_out"]
    ref_err = _rel(lp[f"L12_n5_out_{tag}"], want32)
    ex, _ = _hf_extractor(12, dtype)
    patches = helpers.golden_patches((5,))[5]
    ours = _rel(ex.extract_batch(patches, batch_size=32), want32)
    ex.vit.set_option("f32_stream", True)
    ours_f32s = _rel(ex.extract_batch(patches, batch_size=32), want32)
    ex.cleanup()
    print(f"PARITY lowp {tag}: reference's own {ref_err:.3e}, build (fused LayerNorm) {ours:.3e}, build (f32 stream) {ours_f32s:.3e}")
    assert ours <= ref_err and ours_f32s <= ref_err, (tag, ref_err, ours, ours_f32s)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_fused_dataflow_with_large_row_means_and_massive_channels(dtype):
    """The fused-LayerNorm dataflow multiplies the RAW 16-bit stream and subtracts mean * colsum afterwards: rows whose
    mean is far from zero and channels two orders of magnitude above the rest (the 'massive activations' of trained ViTs)
    are where that cancels most.  The position embedding of the seeded depth-12 model is shifted by +3 in every channel
    and by +60 / -45 / +30 in three channels; features must stay inside the same bounds against the fp32 oracle (the
    all-16-bit module the reference would run measures 1.3e-3 on this construction in a CPU emulation)."""
    from atlaspatch_amd.encoders.vit import build_hip_vit_extractor
    from oracle import vit_oracle
    torch.set_num_threads(min(32, torch.get_num_threads()))
    model = vit_oracle.make_hf_vit(layers=12)
    sd = dict(model.state_dict())
    boost = torch.full((768,), 3.0)
    boost[[7, 300, 511]] += torch.tensor([60.0, -45.0, 30.0])
    sd["embeddings.position_embeddings"] = sd["embeddings.position_embeddings"] + boost
    ex = build_hip_vit_extractor(name="hfvit_massive", arch="vit_b_16", depth=12, state_dict=sd, device=_dev(), dtype=dtype,
                                 source="hf")
    rng = np.random.default_rng(41)
    tiles = [rng.integers(0, 256, (256, 256, 3), dtype=np.uint8) for _ in range(9)]
    want = vit_oracle.extract_batch(sd, tiles, heads=12, batch_size=32)
    got = ex.extract_batch(tiles, batch_size=32)
    ex.vit.set_option("f32_stream", True)
    got_f32s = ex.extract_batch(tiles, batch_size=32)
    ex.cleanup()
    _check(got, want, dtype, "massive activations, fused")
    _check(got_f32s, want, dtype, "massive activations, f32_stream")


def _stress_massive_sd():
    """The weights of golden G1c (a): the seeded depth-12 HF ViT with fc2 rows of three channels x50 in blocks 2 / 5 / 8."""
    from oracle import vit_oracle
    sd = dict(vit_oracle.make_hf_vit(layers=12).state_dict())
    for blk in (2, 5, 8):
        key = next(k for k in sd if k.endswith("weight") and (f"layer.{blk}.output.dense" in k or f"layers.{blk}.mlp.fc2" in k))
        w = sd[key].clone()
        w[[7, 300, 511], :] *= 50.0
        sd[key] = w
    return sd


@pytest.mark.parametrize("dtype,tag", [(torch.float32, "f32"), (torch.float16, "f16"), (torch.bfloat16, "bf16")])
def test_massive_activations_inside_the_references_own_16bit_envelope(dtype, tag, golden_dir):
    """G1c (a): per-block massive activations.  fc2 of blocks 2, 5 and 8 writes three channels of the residual stream at
    ~50x the scale of the rest, so from block 2 on every row of the 16-bit stream carries outliers that set its ulp and
    dominate its LayerNorm statistics.  The REFERENCE's own float16 / bfloat16 extract_batch on these weights (generated
    here by running it, tests/golden/gen_golden.py stress) is the envelope: the build's error against the reference's
    float32 features must not exceed the reference's own in the same type; float32 mode keeps the north star's 1e-3."""
    import os
    from atlaspatch_amd.encoders.vit import build_hip_vit_extractor
    g = np.load(os.path.join(golden_dir, "extract_batch_stress.npz"))
    want = g["massive_f32"]
    sd = _stress_massive_sd()
    ex = build_hip_vit_extractor(name="hfvit_massive_blocks", arch="vit_b_16", depth=12, state_dict=sd, device=_dev(),
                                 dtype=dtype, source="hf")
    rng = np.random.default_rng(77)
    tiles = [rng.integers(0, 256, (256, 256, 3), dtype=np.uint8) for _ in range(5)]
    got = ex.extract_batch(tiles, batch_size=32)
    ours = _rel(got, want)
    if dtype == torch.float32:
        ex.cleanup()
        print(f"PARITY stress massive f32: {ours:.3e}")
        assert ours <= 4e-6 and _elem(got, want) <= 1e-4
        return
    ex.vit.set_option("f32_stream", True)
    ours_f32s = _rel(ex.extract_batch(tiles, batch_size=32), want)
    ex.cleanup()
    ref_err = _rel(g[f"massive_{tag}"], want)
    print(f"PARITY stress massive {tag}: reference's own {ref_err:.3e}, build (fused LayerNorm) {ours:.3e}, build (f32 stream) {ours_f32s:.3e}")
    assert ours <= ref_err and ours_f32s <= ref_err, (tag, ref_err, ours, ours_f32s)


@pytest.mark.parametrize("dtype,tag", [(torch.float32, "f32"), (torch.float16, "f16"), (torch.bfloat16, "bf16")])
def test_layerscale_at_unis_init_value_inside_the_references_own_16bit_envelope(dtype, tag, golden_dir):
    """G1c (b): 24-block ViT-L/16 with LayerScale gamma = 1e-5 (UNI's init_values, models/patch/uni.py:35).  The build folds
    gamma into the proj / fc2 weights (one rounding to the compute type): at 1e-5 the folded float16 weights are
    subnormal.  Envelope = the reference's own 16-bit run of a timm-ordered module with the same weights."""
    import os
    from atlaspatch_amd.encoders.vit import ARCHS, TRANSFORM_RESIZE, build_hip_vit_extractor, random_canonical_state_dict
    g = np.load(os.path.join(golden_dir, "extract_batch_stress.npz"))
    want = g["layerscale_f32"]
    sd = random_canonical_state_dict(ARCHS["uni_v1"], seed=31)
    ex = build_hip_vit_extractor(name="uni_ls_init", arch="uni_v1", state_dict=sd, source="canonical", device=_dev(), dtype=dtype,
                                 resize=TRANSFORM_RESIZE["uni_v1"], expect_size=None)
    rng = np.random.default_rng(77)
    tiles = [rng.integers(0, 256, (256, 256, 3), dtype=np.uint8) for _ in range(3)]
    got = ex.extract_batch(tiles, batch_size=32)
    ours = _rel(got, want)
    if dtype == torch.float32:
        ex.cleanup()
        print(f"PARITY stress layerscale f32: {ours:.3e}")
        assert ours <= 1e-6 and _elem(got, want) <= 1e-4
        return
    ex.vit.set_option("f32_stream", True)
    ours_f32s = _rel(ex.extract_batch(tiles, batch_size=32), want)
    ex.cleanup()
    ref_err = _rel(g[f"layerscale_{tag}"], want)
    print(f"PARITY stress layerscale {tag}: reference's own {ref_err:.3e}, build (fused LayerNorm) {ours:.3e}, build (f32 stream) {ours_f32s:.3e}")
    assert ours <= ref_err and ours_f32s <= ref_err, (tag, ref_err, ours, ours_f32s)


def test_parameter_updates_after_finalize_cannot_leave_stale_folded_weights():
    """The fused-LayerNorm path folds the LayerNorm gains (and LayerScale, and a T copy of pos_embed) into derived weights at
    ap_vit_finalize.  Setting one of their inputs afterwards must not be able to run on stale folded weights: the forward
    refuses until finalize is called again, finalize refuses until the four matrices of every block are uploaded again,
    and after that the features equal those of a freshly built encoder with the changed parameter, bit for bit."""
    import ctypes as C
    from atlaspatch_amd import _lib
    from atlaspatch_amd.encoders.vit import ARCHS, build_hip_vit_extractor, random_canonical_state_dict
    arch = dict(ARCHS["vit_b_16"]); arch["depth"] = 2
    sd = random_canonical_state_dict(arch, seed=12)
    ex = build_hip_vit_extractor(name="t", arch=arch, state_dict=sd, source="canonical", device=_dev(), dtype=torch.float16,
                                 expect_size=256)
    rng = np.random.default_rng(8)
    tiles = [rng.integers(0, 256, (256, 256, 3), dtype=np.uint8) for _ in range(3)]
    before = ex.extract_batch(tiles)
    lib, h = ex.vit.lib, ex.vit._handle
    new_g = (sd["blocks.0.ln1.weight"] * 1.5 + 0.1).contiguous()
    arr = new_g.numpy()
    assert lib.ap_vit_set_param(h, b"blocks.0.ln1.weight", arr.ctypes.data_as(C.c_void_p), arr.size) == 0
    with pytest.raises(_lib.HipLibraryError, match="finalize"):
        ex.extract_batch(tiles)                                            # not finalised any more
    assert lib.ap_vit_finalize(h) == -5 and b"upload" in lib.ap_last_error()   # AP_ERR_STATE: matrices needed again
    for name, t in sd.items():
        if name.startswith("blocks.") and name.endswith((".qkv.weight", ".proj.weight", ".fc1.weight", ".fc2.weight")):
            a = np.ascontiguousarray(t.numpy())
            assert lib.ap_vit_set_param(h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size) == 0
    assert lib.ap_vit_finalize(h) == 0
    after = ex.extract_batch(tiles)
    ex.cleanup()
    sd2 = dict(sd); sd2["blocks.0.ln1.weight"] = new_g
    ex2 = build_hip_vit_extractor(name="t2", arch=arch, state_dict=sd2, source="canonical", device=_dev(), dtype=torch.float16,
                                  expect_size=256)
    fresh = ex2.extract_batch(tiles)
    ex2.cleanup()
    assert np.array_equal(after, fresh) and not np.array_equal(after, before)


# ----------------------------------------------------------------------------- BASELINE configs 3 / 5 at their real depth
def _pil_resized(tiles, size, filt):
    from PIL import Image
    pf = {"bicubic": Image.Resampling.BICUBIC, "bilinear": Image.Resampling.BILINEAR}[filt]
    return np.stack([np.asarray(Image.fromarray(t).resize((size, size), pf)) for t in tiles], 0)


@pytest.mark.parametrize("name,dtype", [("uni_v1", torch.float16), ("uni_v1", torch.float32), ("vit_l_16", torch.float16)])
def test_vit_l_full_depth_vs_fp32_oracle(name, dtype):
    """Config 3's encoder at its real size: 24 blocks, D = 1024, 16 heads; uni_v1 with LayerScale (gamma drawn in
    [0.2, 0.7] so both residual branches matter) behind timm's Resize(224, bicubic); vit_l_16 behind torchvision's
    ImageClassification(crop 224, resize 242) -- against the fp32 CPU restatement on 8 tiles."""
    from atlaspatch_amd.encoders.vit import ARCHS, TRANSFORM_RESIZE, build_hip_vit_extractor, random_canonical_state_dict
    from oracle import vit_oracle
    torch.set_num_threads(min(32, torch.get_num_threads()))
    arch = dict(ARCHS[name])
    sd = random_canonical_state_dict(arch, seed=21)
    g = torch.Generator().manual_seed(22)
    if arch.get("layer_scale"):
        for i in range(arch["depth"]):
            sd[f"blocks.{i}.ls1"] = torch.rand(1024, generator=g) * 0.5 + 0.2
            sd[f"blocks.{i}.ls2"] = torch.rand(1024, generator=g) * 0.5 + 0.2
    size, filt = TRANSFORM_RESIZE[name]
    ex = build_hip_vit_extractor(name=name, arch=arch, state_dict=sd, source="canonical", device=_dev(), dtype=dtype,
                                 resize=(size, filt), expect_size=None)
    rng = np.random.default_rng(23)
    tiles = [rng.integers(0, 256, (256, 256, 3), dtype=np.uint8) for _ in range(8)]
    got = ex.extract_batch(tiles, batch_size=32)
    got_f32s = None
    if dtype != torch.float32:
        ex.vit.set_option("f32_stream", True)
        got_f32s = ex.extract_batch(tiles, batch_size=32)
    ex.cleanup()
    x = vit_oracle.preprocess_center_crop(_pil_resized(tiles, size, filt), crop=224)
    want = vit_oracle.vit_tokens_canonical(sd, x, heads=16, depth=24)[:, 0].numpy()
    assert got.shape == (8, 1024)
    _check(got, want, dtype, f"{name} L24")
    if got_f32s is not None:
        _check(got_f32s, want, dtype, f"{name} L24, f32_stream")


def test_conch_v1_full_depth_f16_vs_fp32_oracle():
    """Config 5's encoder at its real size: 12-block ViT-B/16 trunk on 448-px input (785 tokens) + attentional pooler,
    float16, against the fp32 CPU restatement on 8 tiles (parity unpinned against the absent conch package)."""
    from atlaspatch_amd.encoders.vit import (ARCHS, OPENAI_CLIP_MEAN, OPENAI_CLIP_STD, TRANSFORM_RESIZE, attn_pool_canonical,
                                             build_hip_vit_extractor, random_attn_pool, random_canonical_state_dict)
    from oracle import vit_oracle
    torch.set_num_threads(min(32, torch.get_num_threads()))
    arch = dict(ARCHS["conch_v1"])
    trunk_arch = {k: v for k, v in arch.items() if not k.startswith("pool")}
    trunk = random_canonical_state_dict(trunk_arch, seed=31)
    pool = random_attn_pool(arch, seed=31)
    state = dict(trunk); state.update(attn_pool_canonical(pool))
    rng = np.random.default_rng(32)
    tiles = [rng.integers(0, 256, (256, 256, 3), dtype=np.uint8) for _ in range(8)]
    ex = build_hip_vit_extractor(name="conch_v1", arch=arch, state_dict=state, source="canonical", device=_dev(),
                                 dtype=torch.float16, mean=OPENAI_CLIP_MEAN, std=OPENAI_CLIP_STD,
                                 resize=TRANSFORM_RESIZE["conch_v1"], expect_size=None)
    got = ex.extract_batch(tiles, batch_size=32)
    ex.cleanup()
    want = vit_oracle.conch_encode_image(trunk, pool, tiles, heads=12, depth=12, pool_heads=8)
    assert got.shape == (8, 512)
    _check(got, want, torch.float16, "conch_v1 L12 @448")


def test_vit_layer_scale_uni_shape():
    """UNI-style ViT-L/16 block (LayerScale) at reduced depth vs the oracle."""
    from atlaspatch_amd.encoders.vit import ARCHS, build_hip_vit_extractor, random_canonical_state_dict
    from oracle import vit_oracle
    arch = dict(ARCHS["uni_v1"]); arch["depth"] = 2
    sd = random_canonical_state_dict(arch, seed=5)
    for i in range(2):        # make LayerScale matter
        sd[f"blocks.{i}.ls1"] = torch.rand(1024) * 0.5 + 0.2
        sd[f"blocks.{i}.ls2"] = torch.rand(1024) * 0.5 + 0.2
    ex = build_hip_vit_extractor(name="uni_small", arch=arch, state_dict=sd, device=_dev(),
                                 dtype=torch.float32, source="canonical", expect_size=256)
    # oracle wants HF names: build them from the canonical dict
    hf = {"embeddings.patch_embeddings.projection.weight": sd["patch_embed.weight"],
          "embeddings.patch_embeddings.projection.bias": sd["patch_embed.bias"],
          "embeddings.cls_token": sd["cls_token"].view(1, 1, -1),
          "embeddings.position_embeddings": sd["pos_embed"][None],
          "layernorm.weight": sd["norm.weight"], "layernorm.bias": sd["norm.bias"]}
    ls = {}
    for i in range(2):
        p, b = f"layers.{i}.", f"blocks.{i}."
        q, k, v = sd[b + "qkv.weight"].chunk(3, 0); qb, kb, vb = sd[b + "qkv.bias"].chunk(3, 0)
        hf.update({p + "layernorm_before.weight": sd[b + "ln1.weight"], p + "layernorm_before.bias": sd[b + "ln1.bias"],
                   p + "attention.q_proj.weight": q, p + "attention.q_proj.bias": qb,
                   p + "attention.k_proj.weight": k, p + "attention.k_proj.bias": kb,
                   p + "attention.v_proj.weight": v, p + "attention.v_proj.bias": vb,
                   p + "attention.o_proj.weight": sd[b + "proj.weight"], p + "attention.o_proj.bias": sd[b + "proj.bias"],
                   p + "layernorm_after.weight": sd[b + "ln2.weight"], p + "layernorm_after.bias": sd[b + "ln2.bias"],
                   p + "mlp.fc1.weight": sd[b + "fc1.weight"], p + "mlp.fc1.bias": sd[b + "fc1.bias"],
                   p + "mlp.fc2.weight": sd[b + "fc2.weight"], p + "mlp.fc2.bias": sd[b + "fc2.bias"]})
        ls[f"ls1.{i}"] = sd[b + "ls1"]; ls[f"ls2.{i}"] = sd[b + "ls2"]
    rng = np.random.default_rng(12)
    patches = [rng.integers(0, 256, (256, 256, 3), dtype=np.uint8) for _ in range(6)]
    want = vit_oracle.extract_batch(hf, patches, heads=16, layer_scale=ls)
    got = ex.extract_batch(patches)
    assert got.shape == (6, 1024)
    assert _rel(got, want) <= 1e-3
    ex.cleanup()


def test_forward_chw_plugin_boundary():
    """ap_vit_forward_chw: the [n,3,224,224] boundary a plugin preprocess feeds."""
    from oracle import vit_oracle
    ex, sd = _hf_extractor(2, torch.float32)
    rng = np.random.default_rng(13)
    src = rng.integers(0, 256, (4, 256, 256, 3), dtype=np.uint8)
    x = vit_oracle.preprocess_center_crop(src)
    want = vit_oracle.vit_forward_hf(sd, x, heads=12).numpy()
    got = ex.vit.forward_chw(x.contiguous().to(_dev())).cpu().numpy()
    assert _rel(got, want) <= 1e-3
    ex.cleanup()


# ----------------------------------------------------------------------------- coords
def test_contours_match_reference_golden():
    from atlaspatch_amd.utils.contours import mask_to_contours
    for name, case in helpers.load_coords_cases().items():
        cfg = case["info"]["config"]
        tissue, holes = mask_to_contours(case["mask"], tissue_area_thresh=cfg["tissue_thresh"])
        want_t, want_h = helpers.load_contour_case(name)
        assert len(tissue) == len(want_t), name
        for a, b in zip(tissue, want_t):
            assert np.array_equal(a.reshape(-1, 2), b), name
        assert [len(h) for h in holes] == [len(h) for h in want_h], name
        for hs, ws in zip(holes, want_h):
            for a, b in zip(hs, ws):
                assert np.array_equal(a.reshape(-1, 2), b), name


def test_coords_bit_exact_vs_reference_golden():
    from atlaspatch_amd.services.extraction import coords_from_mask
    for name, case in helpers.load_coords_cases().items():
        cfg = case["info"]["config"]
        coords, geom = coords_from_mask(case["mask"], level0_wh=(cfg["width"], cfg["height"]),
                                        downsamples=cfg["downsamples"], src_mag=cfg["mag"],
                                        tgt_mag=cfg["target_mag"], patch_size=cfg["patch_size"],
                                        step_size=cfg["step_size"], tissue_thresh=cfg["tissue_thresh"])
        assert coords.dtype == np.int32 and coords.shape == case["coords"].shape, name
        assert np.array_equal(coords, case["coords"]), name


def test_coords_full_size_vs_oracle():
    """100k x 100k synthetic slide (BASELINE config 3 geometry): HIP coords == oracle coords."""
    from atlaspatch_amd.core.wsi.synth_pixels import SynthSpec, analytic_mask
    from atlaspatch_amd.services.extraction import coords_from_mask
    from oracle import coords_oracle
    spec = SynthSpec(width=100000, height=100000)
    mask = analytic_mask(spec)
    kw = dict(level0_wh=(spec.width, spec.height), downsamples=list(spec.downsamples), src_mag=20, tgt_mag=20,
              patch_size=256, step_size=None, tissue_thresh=0.0)
    got, _ = coords_from_mask(mask, **kw)
    want, _ = coords_oracle.coords_from_mask(mask, **kw)
    assert got.shape[0] > 30000
    assert np.array_equal(got, want)


def _random_mask(rng, kind, h, w):
    from scipy import ndimage
    if kind == 0:                                   # smooth blobs with holes
        f = ndimage.gaussian_filter(rng.standard_normal((h, w)), rng.uniform(2.0, 8.0))
        m = f > np.quantile(f, rng.uniform(0.35, 0.7))
    elif kind == 1:                                 # speckle: many tiny fragments and pinholes (8- vs 4-connectivity traps)
        m = rng.random((h, w)) < rng.uniform(0.3, 0.7)
    elif kind == 2:                                 # blobs minus speckle: ragged borders, one-pixel walls, nested islands
        f = ndimage.gaussian_filter(rng.standard_normal((h, w)), 6.0)
        m = (f > np.quantile(f, 0.4)) & (rng.random((h, w)) < 0.93)
    elif kind == 3:                                 # checkerboard / diagonal lattices
        yy, xx = np.mgrid[0:h, 0:w]
        p = int(rng.integers(1, 4))
        m = ((xx // p + yy // p) % 2 == 0) & (rng.random((h, w)) < 0.97)
    else:                                           # tissue touching every border, with a hole
        m = np.ones((h, w), bool)
        m[h // 3: h // 2, w // 4: w // 2] = False
        m[h // 3 + 2: h // 3 + 4, w // 4 + 2: w // 4 + 5] = True
    return m.astype(np.float32)


@pytest.mark.parametrize("seed", range(24))
def test_coords_and_contours_random_masks_vs_oracle(seed):
    """Seeded random masks (blobs with holes, speckle, lattices, one-pixel walls, border-touching tissue) with random
    slide geometry: contours (threshold on the GPU + border following in the library) and coordinate rows (device grid
    scan / containment / compaction) equal the CPU oracle's, bit for bit, in the oracle's order."""
    from atlaspatch_amd.services.extraction import coords_from_mask
    from atlaspatch_amd.utils.contours import mask_to_contours
    from oracle import coords_oracle
    rng = np.random.default_rng(1000 + seed)
    h, w = int(rng.integers(40, 160)), int(rng.integers(40, 160))
    mask = _random_mask(rng, seed % 5, h, w)
    thresh = float(rng.choice([0.0, 0.0005, 0.01]))
    got_t, got_h = mask_to_contours(mask, tissue_area_thresh=thresh)
    want_t, want_h = coords_oracle.mask_to_contours(mask, tissue_area_thresh=thresh)
    assert len(got_t) == len(want_t) and [len(x) for x in got_h] == [len(x) for x in want_h]
    for a, b in zip(got_t, want_t):
        assert np.array_equal(np.asarray(a).reshape(-1, 2), np.asarray(b).reshape(-1, 2))
    for hs, ws in zip(got_h, want_h):
        for a, b in zip(hs, ws):
            assert np.array_equal(np.asarray(a).reshape(-1, 2), np.asarray(b).reshape(-1, 2))
    ps = int(rng.choice([224, 256, 512]))
    src_mag, tgt_mag = [(20, 20), (40, 20), (40, 40), (20, 10)][int(rng.integers(0, 4))]
    ds = [(1.0, 4.0, 16.0), (1.0, 2.0, 4.0001), (1.0, 4.00012, 16.00097)][int(rng.integers(0, 3))]
    kw = dict(level0_wh=(int(rng.integers(6000, 22000)), int(rng.integers(6000, 22000))), downsamples=list(ds),
              src_mag=src_mag, tgt_mag=tgt_mag, patch_size=ps,
              step_size=None if seed % 3 else int(ps * rng.choice([0.5, 0.75])), tissue_thresh=thresh)
    got, _ = coords_from_mask(mask, **kw)
    want, _ = coords_oracle.coords_from_mask(mask, **kw)
    assert got.dtype == np.int32 and got.shape == want.shape
    assert np.array_equal(got, want)


def _contours_host_form(mask, thresh):
    """The host border following (csrc/contours.cpp) on the same mask, in a fresh process (AP_CONTOURS_HOST is read once)."""
    import json, os, subprocess, sys, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:
        np.save(os.path.join(tmp, "m.npy"), mask)
        code = ("import numpy as np, json, sys; sys.path.insert(0, %r); from atlaspatch_amd.utils.contours import mask_to_contours; "
                "t, h = mask_to_contours(np.load(%r), tissue_area_thresh=%r); "
                "np.savez(%r, n=len(t), **{f't{i}': a.reshape(-1, 2) for i, a in enumerate(t)}, "
                "**{f'h{i}_{j}': b.reshape(-1, 2) for i, hs in enumerate(h) for j, b in enumerate(hs)}, "
                "nh=np.array([len(x) for x in h], dtype=np.int64))") % (root, os.path.join(tmp, "m.npy"), thresh, os.path.join(tmp, "o.npz"))
        env = dict(os.environ, AP_CONTOURS_HOST="1")
        subprocess.run([sys.executable, "-c", code], check=True, env=env, timeout=600)
        z = np.load(os.path.join(tmp, "o.npz"))
        n = int(z["n"])
        return [z[f"t{i}"] for i in range(n)], [[z[f"h{i}_{j}"] for j in range(int(z["nh"][i]))] for i in range(n)]


@pytest.mark.parametrize("case", ["noise_1024", "blobs_1024", "rings", "thin_walls", "nested", "full", "empty", "single", "frame_touch",
                                  "checker", "odd_733x1024", "diag_chains"])
def test_device_border_following_equals_the_host_form(case):
    """contours_device.hip (component labelling + one thread per border on an LDS bit image) against contours.cpp (Suzuki-Abe
    raster scan with marks) on masks built to break the restatement: one-pixel rings (the hole's start pixel lies on the outer
    border too), islands in holes in islands, diagonal-only contacts, holes that only touch the frame diagonally, 4- vs
    8-connectivity checkerboards, a CMU-1-shaped 733 x 1024 mask, pure noise at 1024 x 1024 (~10^5 borders).  Same contours,
    same order, same points."""
    from scipy import ndimage
    from atlaspatch_amd.utils.contours import mask_to_contours
    rng = np.random.default_rng(77)
    thresh = 0.0
    if case == "noise_1024":
        m = rng.random((1024, 1024)) < 0.5
    elif case == "blobs_1024":
        f = ndimage.gaussian_filter(rng.standard_normal((1024, 1024)), 14.0)
        m = (f > np.quantile(f, 0.55)) & (rng.random((1024, 1024)) < 0.995)
        thresh = 0.001
    elif case == "rings":
        m = np.zeros((96, 130), bool)
        for k, (y, x, r) in enumerate([(20, 20, 8), (20, 60, 3), (60, 40, 15), (60, 100, 2), (30, 110, 1)]):
            yy, xx = np.mgrid[0:96, 0:130]
            d = np.maximum(abs(yy - y), abs(xx - x))
            m |= d == r                                      # one-pixel square rings
        m[60, 40] = True                                     # island in a ring
    elif case == "thin_walls":
        m = np.ones((64, 64), bool)
        m[2:62:4, 2:62] = False                              # stripes separated by one-pixel walls
        m[:, 31] = True
    elif case == "nested":
        m = np.zeros((120, 120), bool)
        for k, r in enumerate(range(55, 4, -6)):
            m[60 - r:60 + r, 60 - r:60 + r] = k % 2 == 0     # island in hole in island ...
    elif case == "full":
        m = np.ones((50, 70), bool)
    elif case == "empty":
        m = np.zeros((50, 70), bool)
    elif case == "single":
        m = np.zeros((9, 9), bool); m[4, 4] = True; m[0, 0] = True; m[8, 8] = True; m[0, 8] = True
    elif case == "frame_touch":
        m = np.ones((40, 40), bool)
        m[0, 5] = False; m[1, 6] = False                     # background pixel on the frame + a diagonal neighbour (a real hole)
        m[39, 10:20] = False; m[20:30, 0] = False; m[10, 39] = False; m[11, 38] = False
    elif case == "checker":
        yy, xx = np.mgrid[0:65, 0:67]
        m = (yy + xx) % 2 == 0
    elif case == "odd_733x1024":
        f = ndimage.gaussian_filter(rng.standard_normal((733, 1024)), 9.0)
        m = f > np.quantile(f, 0.6)
        thresh = 0.0005
    else:                                                    # diagonal chains: 8-connected foreground, 4-separated background
        m = np.zeros((80, 80), bool)
        for k in range(70):
            m[5 + k, 5 + k] = True; m[5 + k, 74 - k] = True
        m[40:44, 10:70] = True
    mask = m.astype(np.float32)
    got_t, got_h = mask_to_contours(mask, tissue_area_thresh=thresh)
    want_t, want_h = _contours_host_form(mask, thresh)
    assert len(got_t) == len(want_t), (case, len(got_t), len(want_t))
    assert [len(x) for x in got_h] == [len(x) for x in want_h], case
    for a, b in zip(got_t, want_t):
        assert np.array_equal(np.asarray(a).reshape(-1, 2), b), case
    for hs, ws in zip(got_h, want_h):
        for a, b in zip(hs, ws):
            assert np.array_equal(np.asarray(a).reshape(-1, 2), b), case


@pytest.mark.parametrize("seed", range(8))
def test_coords_row_bucketed_scan_equals_the_full_scan(seed, monkeypatch):
    """The grid kernel visits, per grid row, only the polygon edges whose y-range meets the row's probe band (the host
    buckets them).  Against the full scan of every vertex (AP_GRID_FLAGS_LEGACY=1: the kernel of rounds 1-2) on random
    masks with geometry chosen so that probes fall ON vertices and ON horizontal / vertical edges (integer scale factors,
    steps that divide them): identical rows, and identical to the oracle."""
    from atlaspatch_amd.services.extraction import coords_from_mask
    from oracle import coords_oracle
    rng = np.random.default_rng(4200 + seed)
    h, w = int(rng.integers(30, 120)), int(rng.integers(30, 120))
    mask = _random_mask(rng, seed % 5, h, w)
    scale = int(rng.choice([32, 64, 128]))                       # level-0 size = mask size x an integer: contours on a lattice
    ps = int(rng.choice([64, 128, 256]))                         # patch / step commensurate with it: probes hit the lattice
    kw = dict(level0_wh=(w * scale, h * scale), downsamples=[1.0, 4.0, 16.0], src_mag=20, tgt_mag=20, patch_size=ps,
              step_size=None if seed % 2 else ps // 2, tissue_thresh=0.0)
    got, _ = coords_from_mask(mask, **kw)
    monkeypatch.setenv("AP_GRID_FLAGS_LEGACY", "1")
    legacy, _ = coords_from_mask(mask, **kw)
    monkeypatch.delenv("AP_GRID_FLAGS_LEGACY")
    want, _ = coords_oracle.coords_from_mask(mask, **kw)
    assert got.shape == legacy.shape == want.shape and got.shape[0] > 0
    assert np.array_equal(got, legacy)
    assert np.array_equal(got, want)


def test_synth_tiles_bit_exact():
    import ctypes as C
    from atlaspatch_amd import _lib
    from atlaspatch_amd.core.wsi.synth_pixels import SynthSpec, render_region
    lib = _lib.load()
    spec = SynthSpec(width=40000, height=40000)
    xy = np.array([[0, 0], [12800, 20224], [39900, 39900], [-100, 500], [20000, 20000]], dtype=np.int32)
    ell = torch.from_numpy(spec.ellipses()).to(_dev())
    d_xy = torch.from_numpy(xy).to(_dev())
    out = torch.empty((len(xy), 256, 256, 3), dtype=torch.uint8, device=_dev())
    _lib.check(lib.ap_synth_tiles(d_xy.data_ptr(), len(xy), 256, 1, 0, spec.width, spec.height, spec.seed,
                                  ell.data_ptr(), ell.shape[0], out.data_ptr(), _lib.current_stream_ptr()))
    got = out.cpu().numpy()
    for i, (x, y) in enumerate(xy):
        assert np.array_equal(got[i], render_region(spec, int(x), int(y), 256, 256, 0)), i


# ----------------------------------------------------------------------------- whole-forward race screen
@pytest.mark.parametrize("name,n,iters", [("vit_b_16", 348, 160), ("uni_v1", 96, 40), ("conch_v1", 24, 40)])
def test_forward_is_repeatable(name, n, iters):
    """The same batch through the full-depth encoder many times: every output equals the first bit for bit.  This is
    the screen that exposed a seam race in the persistent GEMM (a late bias load behind the epilogue's stores: one or
    two wrong images in 4.6 % of the forwards at exactly this size, 348 tiles = 68 556 token rows, ragged last tile)."""
    import os
    from atlaspatch_amd.encoders import build_default_registry
    os.environ["ATLASPATCH_RANDOM_INIT"] = "2"
    try:
        ex = build_default_registry(device="cuda", dtype=torch.float16).create(name)
    finally:
        os.environ.pop("ATLASPATCH_RANDOM_INIT", None)
    tiles = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (n, 256, 256, 3), dtype=np.uint8)).to(_dev())
    ref = torch.empty((n, ex.embedding_dim), dtype=torch.float32, device=_dev())
    out = torch.empty_like(ref)
    ex.forward_device(tiles, ref)
    torch.cuda.synchronize()
    bad = []
    for it in range(iters):
        ex.forward_device(tiles, out)
        torch.cuda.synchronize()
        if not torch.equal(out, ref):
            bad.append((it, torch.nonzero((out != ref).any(dim=1)).flatten().tolist()[:6]))
    ex.cleanup()
    assert not bad, f"{len(bad)} of {iters} forwards differ from the first: {bad[:5]}"


@pytest.mark.parametrize("arch_name,n", [("vit_b_16", 300), ("uni_v1", 270), ("conch_v1", 12)])
def test_features_do_not_depend_on_batch_cut(arch_name, n):
    """The same tiles cut into device batches of 1 / 7 / 255 / 257 tiles or sent as one batch: bit-identical features
    (the 128x128 and 256x256 GEMM kernels agree bitwise, every other kernel works per image or per row).  The reference
    chunks by --feature-batch-size (base.py:83); a drop-in must not make results depend on that knob."""
    from atlaspatch_amd.encoders.vit import (ARCHS, OPENAI_CLIP_MEAN, OPENAI_CLIP_STD, attn_pool_canonical,
                                             build_hip_vit_extractor, random_attn_pool, random_canonical_state_dict)
    arch = dict(ARCHS[arch_name]); arch["depth"] = 2
    if arch_name == "conch_v1":
        trunk_arch = {k: v for k, v in arch.items() if not k.startswith("pool")}
        state = dict(random_canonical_state_dict(trunk_arch, seed=6)); state.update(attn_pool_canonical(random_attn_pool(arch, seed=6)))
        ex = build_hip_vit_extractor(name="t", arch=arch, state_dict=state, source="canonical", device=_dev(),
                                     dtype=torch.float16, mean=OPENAI_CLIP_MEAN, std=OPENAI_CLIP_STD,
                                     resize=(448, "bicubic"), expect_size=None)
    else:
        ex = build_hip_vit_extractor(name="t", arch=arch, state_dict=random_canonical_state_dict(arch, seed=6),
                                     source="canonical", device=_dev(), dtype=torch.float16, expect_size=256)
    tiles = torch.from_numpy(np.random.default_rng(2).integers(0, 256, (n, 256, 256, 3), dtype=np.uint8)).to(_dev())
    ref = torch.empty((n, ex.embedding_dim), dtype=torch.float32, device=_dev())
    ex.forward_device(tiles, ref)
    for chunk in (1, 7, 255, 257):
        if chunk >= n and chunk != 257:
            continue
        out = torch.empty_like(ref)
        for lo in range(0, n, chunk):
            ex.forward_device(tiles[lo:lo + chunk], out[lo:lo + chunk])
        assert torch.equal(out, ref), chunk
    ex.cleanup()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_f32_stream_option_repeatable_and_batch_cut_invariant(dtype):
    """Option f32_stream (round 1's dataflow) keeps the two properties of the default one: bit-equal repeats, bit-equal
    features for any cut into device batches -- and switching the option back and forth leaves no state behind."""
    from atlaspatch_amd.encoders.vit import ARCHS, build_hip_vit_extractor, random_canonical_state_dict
    arch = dict(ARCHS["vit_b_16"]); arch["depth"] = 3
    ex = build_hip_vit_extractor(name="t", arch=arch, state_dict=random_canonical_state_dict(arch, seed=9),
                                 source="canonical", device=_dev(), dtype=dtype, expect_size=256)
    n = 300
    tiles = torch.from_numpy(np.random.default_rng(5).integers(0, 256, (n, 256, 256, 3), dtype=np.uint8)).to(_dev())
    refs = {}
    for mode in (False, True, False, True):
        ex.vit.set_option("f32_stream", mode)
        out = torch.empty((n, ex.embedding_dim), dtype=torch.float32, device=_dev())
        ex.forward_device(tiles, out)
        torch.cuda.synchronize()
        if mode in refs:
            assert torch.equal(out, refs[mode]), mode
        refs[mode] = out
        cut = torch.empty_like(out)
        for lo in range(0, n, 77):
            ex.forward_device(tiles[lo:lo + 77], cut[lo:lo + 77])
        assert torch.equal(cut, out), mode
    assert not torch.equal(refs[False], refs[True])        # they are different computations ...
    assert _rel(refs[False].cpu().numpy(), refs[True].cpu().numpy()) < FALLBACK[str(dtype).split(".")[-1]][0]   # ... of the same features
    ex.cleanup()


def test_full_size_batch_properties():
    """BASELINE's device batch (2048 tiles, full-depth ViT-B/16, f16): size-independent properties instead of an oracle
    run -- (1) the batch equals its four 512-tile quarters forwarded separately, (2) permuting the tiles permutes the
    features, both bit for bit (every image is independent of its batch neighbours), (3) no NaN / Inf."""
    import os
    from atlaspatch_amd.encoders import build_default_registry
    os.environ["ATLASPATCH_RANDOM_INIT"] = "0"
    try:
        ex = build_default_registry(device="cuda", dtype=torch.float16).create("vit_b_16")
    finally:
        os.environ.pop("ATLASPATCH_RANDOM_INIT", None)
    n = 2048
    g = torch.Generator(device=_dev()).manual_seed(4)
    tiles = torch.randint(0, 256, (n, 256, 256, 3), device=_dev(), dtype=torch.uint8, generator=g)
    ref = torch.empty((n, ex.embedding_dim), dtype=torch.float32, device=_dev())
    ex.forward_device(tiles, ref)
    assert torch.isfinite(ref).all()
    out = torch.empty_like(ref)
    for lo in range(0, n, 512):
        ex.forward_device(tiles[lo:lo + 512], out[lo:lo + 512])
    assert torch.equal(out, ref)
    perm = torch.randperm(n, device=_dev(), generator=g)
    ex.forward_device(tiles[perm].contiguous(), out)
    assert torch.equal(out, ref[perm])
    ex.cleanup()


def test_two_half_overlap_mode_is_bit_identical():
    """Option two_half_overlap (opt-in): the batch runs as two halves on two streams; features are the same bits."""
    import os
    from atlaspatch_amd.encoders.vit import ARCHS, build_hip_vit_extractor, random_canonical_state_dict
    arch = dict(ARCHS["vit_b_16"]); arch["depth"] = 3
    ex = build_hip_vit_extractor(name="t", arch=arch, state_dict=random_canonical_state_dict(arch, seed=8),
                                 source="canonical", device=_dev(), dtype=torch.float16, expect_size=256)
    n = 700
    tiles = torch.from_numpy(np.random.default_rng(3).integers(0, 256, (n, 256, 256, 3), dtype=np.uint8)).to(_dev())
    ref = torch.empty((n, ex.embedding_dim), dtype=torch.float32, device=_dev())
    out = torch.empty_like(ref)
    ex.forward_device(tiles, ref)
    ex.vit.set_option("two_half_overlap", True)
    for _ in range(5):
        out.zero_()
        ex.forward_device(tiles, out)
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
    ex.cleanup()


# ----------------------------------------------------------------------------- CLS-only tail of the last block
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.float16, 1e-3), (torch.bfloat16, 8e-3)])
@pytest.mark.parametrize("arch_name", ["vit_b_16", "uni_v1"])
@pytest.mark.parametrize("f32_stream", [False, True])
def test_cls_tail_equals_full_last_block(dtype, tol, arch_name, f32_stream):
    """The default forward computes the last block's K / V for every token and everything after for the CLS row only;
    option full_last_block runs the block for all tokens like the reference module does.  Same features (the CLS
    row never depends on the other rows' outputs of that block); tolerance = rounding differences of the one-row
    attention kernel (f32 softmax weights) vs the tiled one (weights rounded to T before the PV MFMA)."""
    import os
    from atlaspatch_amd.encoders.vit import ARCHS, build_hip_vit_extractor, random_canonical_state_dict
    arch = dict(ARCHS[arch_name]); arch["depth"] = 3
    state = random_canonical_state_dict(arch, seed=4)
    ex = build_hip_vit_extractor(name="t", arch=arch, state_dict=state, source="canonical", device=_dev(), dtype=dtype,
                                 expect_size=256)
    if f32_stream and dtype == torch.float32:
        ex.cleanup()
        pytest.skip("float32 always runs the f32 stream")
    ex.vit.set_option("f32_stream", f32_stream)
    rng = np.random.default_rng(3)
    tiles = [rng.integers(0, 256, (256, 256, 3), dtype=np.uint8) for _ in range(5)]
    got = ex.extract_batch(tiles)
    ex.vit.set_option("full_last_block", True)
    want = ex.extract_batch(tiles)
    ex.cleanup()
    # fused dataflow: with the tail the CLS rows leave the 16-bit stream one block early (the last block's two residual
    # adds happen in f32 instead of T): one more rounding of the stream's magnitude separates the two
    assert _rel(got, want) <= tol, _rel(got, want)


# ----------------------------------------------------------------------------- Pillow-exact device resize
@pytest.mark.parametrize("filt", ["bicubic", "bilinear"])
@pytest.mark.parametrize("in_hw,out_hw", [((256, 256), (224, 224)), ((256, 256), (448, 448)), ((300, 256), (262, 224)),
                                          ((97, 131), (224, 302)), ((512, 512), (224, 224)), ((224, 224), (224, 224))])
def test_device_resample_equals_pillow(filt, in_hw, out_hw):
    """ap_resample_u8 == PIL.Image.resize bit for bit (the oracle here is Pillow itself: pinned)."""
    from PIL import Image
    from atlaspatch_amd.utils.resample import DeviceResampler
    pf = {"bicubic": Image.Resampling.BICUBIC, "bilinear": Image.Resampling.BILINEAR}[filt]
    rng = np.random.default_rng(in_hw[0] * 7 + out_hw[1])
    tiles = rng.integers(0, 256, (5, *in_hw, 3), dtype=np.uint8)
    tiles[1] = 255; tiles[2] = 0
    tiles[3, ::2] = 255; tiles[3, 1::2] = 0                       # overshoot: exercises both clip8 branches
    got = DeviceResampler(in_hw, out_hw, filt, _dev())(torch.from_numpy(tiles).to(_dev())).cpu().numpy()
    for i in range(tiles.shape[0]):
        want = np.asarray(Image.fromarray(tiles[i]).resize((out_hw[1], out_hw[0]), pf))
        assert np.array_equal(got[i], want), (i, np.abs(got[i].astype(int) - want).max())


@pytest.mark.parametrize("seed", range(16))
def test_device_resample_random_shapes_equal_pillow(seed):
    """Random shapes through ap_resample_u8, most of them on the LDS-staged kernel (row bytes a multiple of 16 in, of 4
    out; bands of 8 .. 64 output rows, ragged last band, up- and down-scaling, one-row images), some on the two-kernel
    path: bit-identical to PIL.Image.resize."""
    from PIL import Image
    from atlaspatch_amd.utils.resample import DeviceResampler
    rng = np.random.default_rng(500 + seed)
    filt = ["bicubic", "bilinear"][seed % 2]
    pf = {"bicubic": Image.Resampling.BICUBIC, "bilinear": Image.Resampling.BILINEAR}[filt]
    w = int(rng.integers(1, 24)) * 16 if seed % 4 else int(rng.integers(5, 300))
    ow = int(rng.integers(1, 130)) * 4 if seed % 4 else int(rng.integers(5, 300))
    h, oh = int(rng.integers(1, 400)), int(rng.integers(1, 500))
    n = int(rng.integers(1, 8))
    tiles = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    got = DeviceResampler((h, w), (oh, ow), filt, _dev())(torch.from_numpy(tiles).to(_dev())).cpu().numpy()
    for i in range(n):
        want = np.asarray(Image.fromarray(tiles[i]).resize((ow, oh), pf))
        assert np.array_equal(got[i], want), (seed, (h, w), (oh, ow), filt, i)


def test_uni_transform_device_resize_matches_host_pillow():
    """uni_v1's transform (Resize(224, bicubic) + CenterCrop + Normalize) through the device resize gives the
    same features as resizing with Pillow on the host first (bit-identical inputs -> bit-identical outputs),
    including non-square tiles (shorter side -> 224, centre crop)."""
    from PIL import Image
    from atlaspatch_amd.encoders.vit import ARCHS, build_hip_vit_extractor, random_canonical_state_dict
    arch = dict(ARCHS["uni_v1"]); arch["depth"] = 2
    state = random_canonical_state_dict(arch, seed=2)
    ex = build_hip_vit_extractor(name="uni_test", arch=arch, state_dict=state, source="canonical", device=_dev(),
                                 dtype=torch.float16, resize=(224, "bicubic"), expect_size=None)
    rng = np.random.default_rng(9)
    for hw in ((256, 256), (256, 300)):
        tiles = [rng.integers(0, 256, (*hw, 3), dtype=np.uint8) for _ in range(4)]
        got = ex.extract_batch(tiles)
        h, w = hw
        nw, nh = (224, int(224 * h / w)) if w <= h else (int(224 * w / h), 224)
        pre = [np.asarray(Image.fromarray(t).resize((nw, nh), Image.Resampling.BICUBIC)) for t in tiles]
        want = ex.extract_batch(pre)
        assert np.array_equal(got, want)
    ex.cleanup()


def test_vit_b16_other_tile_sizes_resize_like_torchvision_on_pil():
    """--patch-size 512 (or any size != 256): torchvision's ImageClassification resizes the PIL tile to shorter side
    256 with Pillow's BILINEAR before the centre crop; the registered vit_b_16 does that on the device, bit-identically
    to resizing with Pillow on the host first."""
    import os
    from PIL import Image
    from atlaspatch_amd.encoders import build_default_registry
    os.environ["ATLASPATCH_RANDOM_INIT"] = "5"
    try:
        ex = build_default_registry(device="cuda", dtype=torch.float16).create("vit_b_16")
    finally:
        os.environ.pop("ATLASPATCH_RANDOM_INIT", None)
    rng = np.random.default_rng(12)
    for hw in ((512, 512), (384, 320)):
        tiles = [rng.integers(0, 256, (*hw, 3), dtype=np.uint8) for _ in range(3)]
        h, w = hw
        nw, nh = (256, int(256 * h / w)) if w <= h else (int(256 * w / h), 256)
        pre = [np.asarray(Image.fromarray(t).resize((nw, nh), Image.Resampling.BILINEAR)) for t in tiles]
        assert np.array_equal(ex.extract_batch(tiles), ex.extract_batch(pre))
    ex.cleanup()


# ----------------------------------------------------------------------------- CONCH v1 (a16)
@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1.5e-3), (torch.bfloat16, 1.2e-2)])
def test_conch_visual_tower_vs_oracle(dtype, tol):
    """ViT-B/16 trunk at 448 px (785 tokens: the tiled attention kernel) + one-query attentional pooler + LN,
    against the torch fp32 restatement whose pooler attention is torch's own multi_head_attention_forward.
    Parity unpinned against the real package (absent); tolerance = norm-wise relative error vs the fp32 CPU path."""
    from atlaspatch_amd.encoders.vit import (ARCHS, OPENAI_CLIP_MEAN, OPENAI_CLIP_STD, attn_pool_canonical,
                                             build_hip_vit_extractor, random_attn_pool, random_canonical_state_dict)
    from oracle import vit_oracle
    from PIL import Image
    arch = dict(ARCHS["conch_v1"]); arch["depth"] = 2
    trunk_arch = {k: v for k, v in arch.items() if not k.startswith("pool")}
    trunk = random_canonical_state_dict(trunk_arch, seed=11)
    pool = random_attn_pool(arch, seed=11)
    state = dict(trunk); state.update(attn_pool_canonical(pool))
    rng = np.random.default_rng(5)
    tiles = [rng.integers(0, 256, (256, 256, 3), dtype=np.uint8) for _ in range(3)]
    want = vit_oracle.conch_encode_image(trunk, pool, tiles, heads=12, depth=2, pool_heads=8)
    ex = build_hip_vit_extractor(name="conch_test", arch=arch, state_dict=state, source="canonical", device=_dev(),
                                 dtype=dtype, mean=OPENAI_CLIP_MEAN, std=OPENAI_CLIP_STD,
                                 resize=(448, "bicubic"), expect_size=None)
    got = ex.extract_batch(tiles, batch_size=32)
    ex.cleanup()
    assert got.shape == (3, 512) and got.dtype == np.float32
    assert _rel(got, want) <= tol, _rel(got, want)


def test_conch_registered_and_float32_rejected():
    from atlaspatch_amd.encoders import build_default_registry
    from atlaspatch_amd import _lib
    reg = build_default_registry(device="cuda", dtype=torch.float32)
    assert "conch_v1" in reg.available()
    import os
    os.environ["ATLASPATCH_RANDOM_INIT"] = "3"
    try:
        with pytest.raises(_lib.HipLibraryError):
            reg.create("conch_v1")          # float32: 785 tokens exceed the f32 attention kernel / pooler is f16|bf16
    finally:
        os.environ.pop("ATLASPATCH_RANDOM_INIT", None)


# ----------------------------------------------------------------------------- SAM2 Hiera-T image path (a4)
def test_sam2_image_path_vs_oracle():
    """Hiera-T trunk + FpnNeck + box-prompt mask decoder on the HIP float32 operator set vs the torch fp32
    restatement (oracle/sam2_oracle.py), same seeded random weights.  Parity unpinned against the real package
    (absent); tolerance: norm-wise 1e-4 on features / logits (float32, different summation order), masks equal
    except where the upsampled logit is within rounding of the threshold."""
    from atlaspatch_amd.services.sam2_hip import Sam2HipPredictor
    from oracle import sam2_oracle as so
    sd = so.random_state_dict(3)
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (1024, 1024, 3), dtype=np.uint8)
    img[200:700, 300:800] = (img[200:700, 300:800] // 3 + 120).astype(np.uint8)        # some structure
    pred = Sam2HipPredictor(sd, device=_dev())
    embed_w, s0_w, s1_w = so.image_features(sd, img)
    embed_w = embed_w + sd["sam_prompt_encoder.no_mask_embed.weight"].view(1, 256, 1, 1)
    d_img = torch.from_numpy(img).to(_dev())
    embed, s0, s1 = pred.image_features(d_img)
    assert _rel(embed.cpu().numpy(), embed_w[0].flatten(1).t().numpy()) <= 1e-4
    assert _rel(s0.cpu().numpy(), s0_w[0].flatten(1).t().numpy()) <= 1e-4
    assert _rel(s1.cpu().numpy(), s1_w[0].flatten(1).t().numpy()) <= 1e-4
    logits = pred.mask_logits(embed, s0, s1).cpu().numpy()
    want = so.predict_logits(sd, img)
    assert _rel(logits, want) <= 2e-4, _rel(logits, want)
    # end to end through predict_image on a non-square thumbnail (BILINEAR in, NEAREST out)
    thumb = rng.integers(0, 256, (733, 1024, 3), dtype=np.uint8)
    got = pred.predict_image(thumb)
    ref = so.predict_image(sd, thumb)
    assert got.shape == ref.shape == (733, 1024) and set(np.unique(got)) <= {0.0, 1.0}
    assert (got != ref).mean() <= 2e-4, (got != ref).mean()
    pred.close()
