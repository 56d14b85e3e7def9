"""bench.py's launch path: `--gpus N` without a launcher, `--encoder`, and the N > 1 branch (BASELINE configs 4 / 5:
slide-per-rank + one all-gather of the feature blocks; reference story: README.md:527,628 = separate jobs per GPU,
orchestration/runner.py:154-167 lock files)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "AP_BENCH_BACKEND",
                                                            "AP_BENCH_ONE_GPU")}
    res = subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + "\n" + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def test_encoder_geometry_matches_the_surveyed_flop_counts():
    """ViT-B/16 35.126 GFLOP per tile (33.695 GEMM + 1.431 attention), ViT-L/16 123.107, CONCH ~157."""
    sys.path.insert(0, ROOT)
    import bench
    from atlaspatch_amd.encoders.vit import ARCHS
    g = bench.encoder_geometry(ARCHS["vit_b_16"])
    assert g["tokens"] == 197 and abs(g["model"] / 1e9 - 35.126) < 1e-3 and abs(g["executed"] / 1e9 - 32.695) < 1e-3
    g = bench.encoder_geometry(ARCHS["uni_v1"])
    assert g["tokens"] == 197 and abs(g["model"] / 1e9 - 123.107) < 1e-3 and g["executed"] < g["model"]
    g = bench.encoder_geometry(ARCHS["conch_v1"])
    assert g["tokens"] == 785 and 156 < g["model"] / 1e9 < 160 and g["executed"] == g["model"]
    assert set(bench.ENCODERS) == {"vit_b_16", "uni_v1", "conch_v1", "vit_b_32", "vit_l_32", "vit_h_14", "uni_v2", "dinov2_small", "dinov2_base",
                                   "dinov2_large", "dinov2_giant", "phikon_v1", "phikon_v2", "midnight", "h_optimus_0", "prov_gigapath",
                                   "lunit_vit_small_patch16_dino", "lunit_vit_small_patch8_dino", "pathorchestra", "clip_vit_b_32",
                                   "clip_vit_b_16", "clip_vit_l_14", "clip_vit_l_14_336", "plip", "biomedclip", "virchow_v1", "virchow_v2",
                                   "dinov3_vits16", "dinov3_vitb16", "dinov3_vitl16", "dinov3_vith16_plus", "dinov3_vit7b16",
                                   "vit_l_16", "h_optimus_1", "h0_mini", "quilt_b_32", "quilt_b_16", "dinov3_vits16_plus",
                                   "dinov3_vitl16_sat", "dinov3_vit7b16_sat"}
    from atlaspatch_amd.encoders import build_default_registry
    assert set(bench.ENCODERS) == set(build_default_registry(device="cpu").available())       # every registered name has a bench line
    # round 4: the rest of the encoder files.  uni_v2 = 265 tokens (1 + 8 registers + 256 patches), SwiGLU packed fc1;
    # vit_h_14 = 1370 tokens at 518 px
    g = bench.encoder_geometry(ARCHS["uni_v2"])
    want = 2.0 * 256 * 1536 * 588 + 24 * (2.0 * 265 * (4 * 1536 ** 2 + 3 * 1536 * 4096) + 4.0 * 265 ** 2 * 1536)
    assert g["tokens"] == 265 and abs(g["model"] - want) < 1e3 and g["executed"] < g["model"]
    g = bench.encoder_geometry(ARCHS["vit_h_14"])
    assert g["tokens"] == 1370 and 1.9e12 < g["model"] < 2.2e12
    assert bench.encoder_geometry(ARCHS["vit_b_32"])["tokens"] == 50


@pytest.mark.gpu
def test_gpus_2_without_a_launcher_runs_two_ranks_and_gathers_both_feature_blocks(tmp_path):
    """`python bench.py --gpus 2` (no torchrun in front): bench re-executes itself under torch.distributed.run with two
    ranks; on a one-GPU box they share cuda:0 and the collective goes through gloo (marked as a rehearsal).  The matrix
    the all-gather assembles must be the two single-rank matrices, rank 0's block first, bit for bit."""
    common = ["--steps", "2", "--warmup", "1", "--batch", "64", "--slide", "12000", "--no-extras", "--no-cpu-baseline"]
    both = _run(["--gpus", "2", "--dump-features", str(tmp_path / "g.npy")] + common)
    assert both["n_gpus"] == 2 and both["config"]["ranks_in_process_group"] == 2 and both["config"]["gpus_requested"] == 2
    assert [r["rank"] for r in both["per_rank"]] == [0, 1]
    assert [r["slide_seed"] for r in both["per_rank"]] == [1234, 1235]
    assert all(r["patches_per_s"] > 0 and r["all_gather_ms"] > 0 for r in both["per_rank"])
    assert both["all_gather"]["bytes_per_rank"] == 2 * 64 * 768 * 4
    import torch
    if torch.cuda.device_count() < 2:
        assert "REHEARSAL" in both["data"]
    gathered = np.load(tmp_path / "g.npy")
    assert gathered.shape == (2 * 2 * 64, 768)
    for r, seed in enumerate((1234, 1235)):
        one = _run(["--gpus", "1", "--slide-seed", str(seed), "--dump-features", str(tmp_path / f"s{r}.npy")] + common)
        assert one["n_gpus"] == 1 and "per_rank" not in one and one["data"] == "synthetic"
        single = np.load(tmp_path / f"s{r}.npy")
        assert single.shape == (2 * 64, 768) and np.isfinite(single).all() and np.abs(single).max() > 0
        assert np.array_equal(gathered[r * 128:(r + 1) * 128], single), f"rank {r}'s block differs from its single-rank run"
    assert not np.array_equal(gathered[:128], gathered[128:])          # the two ranks embedded different slides
    # the all-pairs exchange (ATLASPATCH_GATHER_ALGO=pairs) assembles the same matrix
    os.environ["ATLASPATCH_GATHER_ALGO"] = "pairs"
    try:
        pairs = _run(["--gpus", "2", "--dump-features", str(tmp_path / "p.npy")] + common)
    finally:
        del os.environ["ATLASPATCH_GATHER_ALGO"]
    assert pairs["all_gather"]["algorithm"] == "pairs" and both["all_gather"]["algorithm"] == "allgather"
    assert np.array_equal(np.load(tmp_path / "p.npy"), gathered)


@pytest.mark.gpu
def test_a_line_is_refused_when_the_ranks_that_ran_are_not_the_gpus_asked_for():
    """`--gpus 8` inside a process group of one rank (a launcher given the wrong --nproc-per-node): bench.py exits with a message
    instead of printing an n_gpus = 8 line measured on one device."""
    env = {k: v for k, v in os.environ.items() if k not in ("AP_BENCH_BACKEND", "AP_BENCH_ONE_GPU")}
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "1", "--warmup", "0", "--batch", "16", "--no-extras",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode != 0 and "--gpus 8 but the process group has 1 rank" in res.stderr, res.stderr[-2000:]
    assert not any(ln.startswith("{") for ln in res.stdout.splitlines())


@pytest.mark.gpu
def test_encoder_option_runs_conch_fp16_with_its_own_roofline_shape():
    """Config 5's encoder (CONCH v1 visual tower, fp16, 785 tokens) has a bench line of its own."""
    line = _run(["--encoder", "conch_v1", "--steps", "2", "--warmup", "1", "--batch", "32", "--slide", "12000", "--no-extras",
                 "--cpu-sample", "4"])
    assert line["config"]["encoder"] == "conch_v1" and "conch_v1" in line["metric"] and line["dtype"] == "f16"
    assert "[B*785,768]x[768,3072]" in line["roofline"]["kernel"]
    assert line["roofline"]["algorithmic_flop_per_launch"] == 2.0 * 32 * 785 * 3072 * 768
    assert 0 < line["roofline"]["frac"] < 1 and line["value"] > 0
    assert line["cpu_baseline"]["rel_err_gpu_vs_cpu"] < 3e-3


@pytest.mark.gpu
def test_gpus_8_rehearsal_eight_ranks_on_one_device(tmp_path):
    """The shape of BASELINE configs 4 / 5 (8 ranks, slide-per-rank, one all-gather) run as eight processes that share this
    box's one GPU (gloo): ports, per-rank slide seeds, the gathered matrix's block order and `_pin_order`'s per-local-rank core
    split all execute before a real 8-GPU lease does.  Small batch: 8 x (weights + workspace of 16 tiles) fits easily."""
    common = ["--steps", "1", "--warmup", "1", "--batch", "16", "--slide", "12000", "--no-extras", "--no-cpu-baseline"]
    line = _run(["--gpus", "8", "--dump-features", str(tmp_path / "g8.npy")] + common, timeout=1500)
    assert line["n_gpus"] == 8 and line["config"]["ranks_in_process_group"] == 8 and line["scaling"] == "weak"
    assert [r["rank"] for r in line["per_rank"]] == list(range(8))
    assert [r["slide_seed"] for r in line["per_rank"]] == [1234 + r for r in range(8)]
    assert line["all_gather"]["bytes_per_rank"] == 16 * 768 * 4 and line["value"] > 0
    g = np.load(tmp_path / "g8.npy")
    assert g.shape == (8 * 16, 768) and np.isfinite(g).all()
    blocks = [g[r * 16:(r + 1) * 16] for r in range(8)]
    assert all(np.abs(b).max() > 0 for b in blocks)
    assert all(not np.array_equal(blocks[0], b) for b in blocks[1:])          # every rank embedded its own slide
    one = _run(["--gpus", "1", "--slide-seed", "1239", "--dump-features", str(tmp_path / "s5.npy")] + common)
    assert np.array_equal(np.load(tmp_path / "s5.npy"), blocks[5]) and one["n_gpus"] == 1     # rank 5's block = its single-rank run
    from atlaspatch_amd.services.tile_ring import _pin_order
    import torch
    orders = [_pin_order(torch.device("cuda:0"), local_rank=r, local_world=8) for r in range(8)]
    firsts = [o[0] for o in orders if o]
    assert len(set(firsts)) == len(firsts)                                    # the ranks' first choices are distinct cores
