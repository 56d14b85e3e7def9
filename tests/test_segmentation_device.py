"""The segmentation path's pixel operations on the device (services/segmentation.py:104-118,202-206; core/wsi/iwsi.py:246-323):
whole-level read, cv2.resize to 1.25x, Pillow thumbnail (reduce + bicubic), the SAM2 input / mask resizes.  Pillow IS the
reference's resampler there and is installed: every device result is compared with Pillow itself, bit for bit."""
import json

import numpy as np
import pytest
from PIL import Image

REDUCE_CASES = [((50, 61), 3, None), ((37, 40), (2, 5), None), ((64, 64), 4, None), ((33, 35), (7, 6), (2, 3, 31, 30)),
                ((20, 31), (1, 3), None), ((45, 47), 5, (0, 0, 46, 44)), ((30, 30), (2, 1), None), ((29, 31), 8, None),
                ((100, 90), (6, 4), (5, 7, 88, 97)), ((19, 23), 16, None), ((700, 1000), 3, None), ((257, 513), (2, 2), (1, 0, 512, 257))]
THUMB_CASES = [(6250, 6250, 1024), (1000, 700, 1024), (2875, 2057, 1024), (4000, 900, 1024), (1025, 1025, 1024),
               (3000, 2100, 512), (2049, 4100, 1024), (5000, 3333, 1024), (1024, 1024, 1024), (1500, 1100, 1024)]


# ----------------------------------------------------------------------------- host statements vs Pillow (no GPU)
def test_reduce_statement_and_thumbnail_plan_equal_pillow():
    from atlaspatch_amd.utils.resample import pillow_reduce_numpy, pillow_thumbnail_plan
    rng = np.random.default_rng(0)
    for (h, w), f, box in REDUCE_CASES[:10]:
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(pillow_reduce_numpy(a, f, box), np.asarray(Image.fromarray(a).reduce(f, box=box))), ((h, w), f, box)
    for w, h, m in THUMB_CASES + [(10000, 333, 1024)]:
        if w * h > 12e6:
            w, h = w // 2, h // 2             # keep the CPU suite short; the full size runs on the device test
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        im = Image.fromarray(a)
        want = im.copy()
        want.thumbnail((m, m))
        plan = pillow_thumbnail_plan((w, h), (m, m))
        if plan is None:
            assert want.size == (w, h)
            continue
        final, red = plan
        cur, box = im, None
        if red is not None:
            factor, rbox, box = red
            cur = cur.reduce(factor, box=rbox)
        got = cur.resize(final, Image.Resampling.BICUBIC, box=box)
        assert got.size == want.size and np.array_equal(np.asarray(got), np.asarray(want)), (w, h, m, plan)


def test_vectorised_resample_tables_equal_the_loop_form_and_nearest_index_equals_pillow():
    from atlaspatch_amd.utils.resample import _pillow_resample_tables_scalar, pillow_nearest_index, pillow_resample_tables
    for case in [(256, 224, "bicubic", None), (256, 448, "bicubic", None), (256, 242, "bilinear", None),
                 (2084, 1024, "bicubic", (0.0, 6250 / 3)), (733, 1024, "bilinear", None), (1024, 733, "bilinear", None),
                 (1025, 512, "bicubic", (0.0, 1024.5)), (100, 37, "bicubic", (3.25, 96.5)), (50, 1, "bilinear", None), (7, 200, "bicubic", None)]:
        a, b = pillow_resample_tables(*case), _pillow_resample_tables_scalar(*case)
        assert a[2] == b[2] and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), case
    m = (np.random.default_rng(0).random((1024, 1024)) > 0.5).astype(np.uint8) * 255
    for ow, oh in [(1024, 733), (681, 1024), (1000, 999), (37, 1), (1024, 1024)]:
        want = np.asarray(Image.fromarray(m, mode="L").resize((ow, oh), resample=Image.Resampling.NEAREST))
        assert np.array_equal(m[pillow_nearest_index(1024, oh)][:, pillow_nearest_index(1024, ow)], want), (ow, oh)


# ----------------------------------------------------------------------------- device kernels vs Pillow
@pytest.mark.gpu
@pytest.mark.parametrize("shape,factor,box", REDUCE_CASES)
def test_device_reduce_equals_pillow(shape, factor, box):
    import torch
    from atlaspatch_amd.utils.resample import pillow_reduce_device
    a = np.random.default_rng(shape[0] * 31 + shape[1]).integers(0, 256, shape + (3,), dtype=np.uint8)
    got = pillow_reduce_device(torch.from_numpy(a).cuda(), factor, box).cpu().numpy()
    want = np.asarray(Image.fromarray(a).reduce(factor, box=box))
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,m", THUMB_CASES)
def test_device_thumbnail_equals_pillow_thumbnail(w, h, m):
    """Image.thumbnail((m, m)) with Pillow 12's defaults (BICUBIC, reducing_gap 2.0) -- what the reference calls on the
    1.25x image (segmentation.py:202-206) -- on the device: same size, same bytes."""
    import torch
    from atlaspatch_amd.utils.resample import pillow_thumbnail_device
    a = np.random.default_rng(w + h).integers(0, 256, (h, w, 3), dtype=np.uint8)
    want = Image.fromarray(a)
    want.thumbnail((m, m))
    got = pillow_thumbnail_device(torch.from_numpy(a).cuda(), (m, m)).cpu().numpy()
    assert got.shape == (want.size[1], want.size[0], 3) and np.array_equal(got, np.asarray(want))


@pytest.mark.gpu
def test_synth_region_equals_render_region_and_device_thumbnail_path_equals_host_path(tmp_path):
    """read_level_device (ap_synth_region) == render_region; prepare_thumbnail_device == prepare_thumbnail, on a slide whose
    1.25x level is exact (no cv2 resize) and on one where the level image is shrunk by cv2.resize first."""
    import torch
    from atlaspatch_amd.core.config import SegmentationConfig
    from atlaspatch_amd.core.wsi.synth_pixels import SynthSpec, render_region
    from atlaspatch_amd.core.wsi.synth_wsi import SynthWSI
    from atlaspatch_amd.services.segmentation import prepare_thumbnail, prepare_thumbnail_device
    cfg = SegmentationConfig(checkpoint_path=None, config_path=tmp_path / "x.yaml", device="cuda")
    for i, kw in enumerate([dict(width=30000, height=21000, seed=3, mag=20, mpp=0.5, downsamples=[1, 4, 16]),
                            dict(width=16000, height=11200, seed=5, mag=40, mpp=0.25, downsamples=[1, 4, 16]),
                            dict(width=3000, height=2100, seed=6, mag=5, mpp=2.0, downsamples=[1])]):
        p = tmp_path / f"s{i}.synth"
        json.dump(kw, open(p, "w"))
        wsi = SynthWSI(str(p))
        wsi._ensure_loaded()
        lv = wsi.nlvl - 1
        w, h = wsi.dims[lv]
        dev_level = wsi.read_level_device(lv, (w, h), torch.device("cuda:0")).cpu().numpy()
        spec = SynthSpec(width=kw["width"], height=kw["height"], seed=kw["seed"], mag=kw["mag"], mpp=kw["mpp"],
                         downsamples=tuple(float(d) for d in kw["downsamples"]))
        assert np.array_equal(dev_level, render_region(spec, 0, 0, w, h, lv))
        host = np.asarray(prepare_thumbnail(wsi, cfg))
        dev = prepare_thumbnail_device(wsi, cfg, torch.device("cuda:0")).cpu().numpy()
        assert host.shape == dev.shape and max(host.shape[:2]) <= 1024 and np.array_equal(host, dev), (i, host.shape, dev.shape)


@pytest.mark.gpu
def test_sam2_predict_device_equals_predict_image():
    """predict_device (thumbnail in HBM: device BILINEAR resize to 1024^2, graph, device NEAREST gather back) returns the
    mask predict_image computes from the same thumbnail through host Pillow -- bit for bit, square and non-square."""
    import torch
    from atlaspatch_amd.services.sam2_hip import Sam2HipPredictor
    from atlaspatch_amd.services.segmentation import random_sam2_state_dict
    pred = Sam2HipPredictor(random_sam2_state_dict(3), device="cuda")
    rng = np.random.default_rng(5)
    for h, w in [(733, 1024), (1024, 1024), (1024, 681), (512, 700)]:
        thumb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        want = pred.predict_image(thumb, resize_to_input=True)
        got = pred.predict_device(torch.from_numpy(thumb).cuda(), resize_to_input=True)
        assert got.shape == want.shape == (h, w) and got.dtype == np.float32 and set(np.unique(got)) <= {0.0, 1.0}
        assert np.array_equal(got, want), (h, w, float((got != want).mean()))
    pred.close()


@pytest.mark.gpu
def test_sam2_batched_forward_equals_the_single_slide_forwards_bit_for_bit():
    """``--seg-batch-size`` > 1: the trunk runs on the stacked thumbnails (GEMMs planned like a single image,
    ``ap_sgemm_stacked``; windows, images and rows are independent in every other operator).  Logits AND masks of every
    image equal its own single-image forward exactly -- launch by launch and as the captured per-batch-size graph --
    for batch sizes 2, 3 and 4 with mixed thumbnail shapes, whatever the image is batched with."""
    import torch
    from atlaspatch_amd.services.sam2_hip import Sam2HipPredictor
    from atlaspatch_amd.services.segmentation import random_sam2_state_dict
    pred = Sam2HipPredictor(random_sam2_state_dict(4), device="cuda")
    rng = np.random.default_rng(11)
    shapes = [(733, 1024), (1024, 1024), (1024, 681), (512, 700), (900, 1000)]
    thumbs = [torch.from_numpy(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).cuda() for h, w in shapes]
    single = [pred.predict_device(t, resize_to_input=True) for t in thumbs]
    # logits of the stacked trunk vs the single trunk, no graph
    imgs = torch.from_numpy(rng.integers(0, 256, (3, 1024, 1024, 3), dtype=np.uint8)).cuda()
    with torch.inference_mode():
        lone = [pred.mask_logits(*pred.image_features(imgs[b])).clone() for b in range(3)]
        stacked = [pred.mask_logits(*f).clone() for f in pred.image_features_batch(imgs)]
    for b in range(3):
        assert torch.equal(lone[b], stacked[b]), (b, float((lone[b] - stacked[b]).abs().max()))
    for group in ([0, 1], [2, 3, 4], [4, 0, 1, 2], [3, 2]):
        got = pred.predict_batch_device([thumbs[i] for i in group], resize_to_input=True)
        for i, g in zip(group, got):
            assert g.shape == single[i].shape and np.array_equal(g, single[i]), (group, i, float((g != single[i]).mean()))
    pred.close()


@pytest.mark.gpu
def test_openslide_level_read_in_parallel_strips_equals_one_read_region(tmp_path):
    """OpenSlideWSI.read_level_device: full-width strips read by libopenslide on a thread pool == extract((0, 0), level,
    dims) through openslide-python, for a level with an integer downsample (strips) -- run in a subprocess, the stub
    library is resolved once per process."""
    import os
    import subprocess
    import sys
    import textwrap
    from tools import stub_openslide as so
    lib = so.build(str(tmp_path))
    code = f"""
        import numpy as np, torch
        from tools import stub_openslide as so
        from atlaspatch_amd.core.wsi import openslide_wsi
        openslide_wsi.openslide = so.python_module({lib!r})
        path = so.write_slide({str(tmp_path / 's.svs')!r}, 23000, 17000, seed=2, alpha_period=3)
        wsi = openslide_wsi.OpenSlideWSI(path)
        wsi._ensure_loaded()
        for lv in (2, 1):
            w, h = wsi.dims[lv]
            got = wsi.read_level_device(lv, (w, h), torch.device("cuda:0")).cpu().numpy()
            want = wsi.extract((0, 0), lv, (w, h))
            assert got.shape == want.shape == (h, w, 3) and np.array_equal(got, want), lv
        thumb = wsi.get_thumbnail_at_power_device(power=1.25, device="cuda:0").cpu().numpy()
        assert np.array_equal(thumb, np.asarray(wsi.get_thumbnail_at_power(power=1.25)))
        print("ok")
        """
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ATLASPATCH_LIBOPENSLIDE=lib)
    res = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], capture_output=True, text=True, env=env, cwd=root, timeout=600)
    assert res.returncode == 0 and "ok" in res.stdout, res.stdout[-1500:] + res.stderr[-3000:]
