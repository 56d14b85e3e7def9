#!/usr/bin/env python3
"""Smoke matrix over CLI options that the test-suite does not combine: several extractors in one run, every precision,
a non-default patch size, two slides."""
import json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ATLASPATCH_RANDOM_INIT", "1")
import numpy as np
from click.testing import CliRunner
from atlaspatch_amd.cli import cli
from atlaspatch_amd.utils.h5 import h5
ok = True
with tempfile.TemporaryDirectory() as tmp:
    slides = []
    for i, side in enumerate((9000, 7000)):
        p = os.path.join(tmp, f"s{i}.synth")
        json.dump({"width": side, "height": side - 1000, "seed": 10 + i, "mag": 20, "mpp": 0.5, "downsamples": [1, 4, 16]}, open(p, "w"))
        slides.append(p)
    cases = [(["--feature-extractors", "vit_b_16,uni_v1", "--feature-precision", "bfloat16"], "bf16 two extractors", tmp),
             (["--feature-extractors", "vit_b_16", "--feature-precision", "float32"], "f32", tmp),
             (["--feature-extractors", "vit_b_16,conch_v1", "--feature-precision", "float16", "--patch-size", "512"], "ps512", slides[0]),
             (["--feature-extractors", "vit_l_16", "--feature-precision", "float16", "--target-mag", "10", "--save-images",
               "--visualize-grids", "--visualize-mask", "--visualize-contours"], "mag20->10 (512 reads, cv2 resize) + images + overlays", slides[1]),
             (["--feature-extractors", "vit_b_16", "--feature-precision", "float16", "--no-fast-mode", "--step-size", "128",
               "--target-mag", "5"], "mag20->5 (1024 reads), no-fast-mode, overlap", slides[1]),
             # the widened encoder families off the default tile size / precision
             (["--feature-extractors", "dinov2_small,clip_vit_b_32,h0_mini", "--feature-precision", "bfloat16", "--patch-size", "512"],
              "bf16, 512-px tiles: DINOv2 / CLIP projection / class | mean pooling", slides[0]),
             (["--feature-extractors", "dinov3_vits16,phikon_v1,lunit_vit_small_patch16_dino", "--feature-precision", "float32"],
              "f32: DINOv3 rotary, HF ViT eps 1e-12, Lunit", slides[1])]
    for k, (extra, label, target) in enumerate(cases):
        out = os.path.join(tmp, f"out{k}")
        args = ["process", target, "-o", out] + (["--target-mag", "20"] if "--target-mag" not in extra else []) + \
            (["--patch-size", "256"] if "--patch-size" not in extra else []) + extra
        res = CliRunner().invoke(cli, args, catch_exceptions=False)
        good = res.exit_code == 0 and "failures: 0" in res.output
        shapes = {}
        if good:
            for f in sorted(os.listdir(os.path.join(out, "patches"))):
                if f.endswith(".h5"):
                    with h5.File(os.path.join(out, "patches", f), "r") as fh:
                        shapes[f] = {name: tuple(fh["features"][name].shape) for name in fh["features"].keys()}
                        for name in fh["features"].keys():
                            good &= bool(np.isfinite(fh["features"][name][:]).all())
        ok &= good
        print(label, "OK" if good else "FAILED", shapes if good else res.output[-600:], flush=True)
sys.exit(0 if ok else 1)
