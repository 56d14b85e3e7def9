"""The native batched OpenSlide read behind OpenSlideWSI.read_tiles_into (csrc/openslide_host.cpp) against the per-tile
path the reference runs: read_region(...).convert("RGB") (core/wsi/openslide_wsi.py:184-205).

libopenslide / openslide-python are not in the image: tools/stub_openslide supplies a stand-in shared library with
libopenslide's interface (premultiplied ARGB, partial alpha, transparent outside the slide) and an openslide-python-shaped
module that binds it the way openslide-python does.  Host-only tests run in a subprocess each: the library is resolved
once per process (dlopen at first use)."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_py(code, env_extra, tmp_path):
    env = {k: v for k, v in os.environ.items() if k != "ATLASPATCH_LIBOPENSLIDE"}
    env.update(env_extra)
    res = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], capture_output=True, text=True, env=env, cwd=ROOT,
                         timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    return json.loads(res.stdout.strip().splitlines()[-1])


def test_native_hook_equals_read_region_convert_rgb_bit_for_bit(tmp_path):
    from tools import stub_openslide as so
    lib = so.build(str(tmp_path))
    out = _run_py(f"""
        import json, os, numpy as np
        from tools import stub_openslide as so
        from atlaspatch_amd import _lib
        from atlaspatch_amd.core.wsi import openslide_wsi
        mod = so.python_module({lib!r})
        openslide_wsi.openslide = mod
        path = so.write_slide({str(tmp_path / 's.svs')!r}, 9000, 7000, seed=11, alpha_period=2)
        assert _lib.load().ap_host_openslide_available() == 1
        wsi = openslide_wsi.OpenSlideWSI(path)
        wsi._ensure_loaded()
        rng = np.random.default_rng(3)
        checked = partial = black = 0
        for level, side, n in ((0, 256, 24), (1, 256, 10), (2, 128, 8), (0, 512, 4), (0, 37, 6)):
            ds = int(wsi.ds[level])
            rows = [[int(rng.integers(-300, 9000)) if i % 5 else 8900, int(rng.integers(0, 7000)) if i % 7 else 6900, side, side, level]
                    for i in range(n)]
            rows = [[max(0, x), y, a, b, c] for x, y, a, b, c in rows]
            dst = np.full((n, side, side, 3), 99, np.uint8)
            before = mod.calls["read_region"]
            assert wsi.read_tiles_into(rows, dst.ctypes.data, side) is True
            assert mod.calls["read_region"] == before            # no per-tile Python read happened
            for i, (x, y, _, _, lv) in enumerate(rows):
                ref = wsi.extract((x, y), lv, (side, side))
                assert ref.shape == (side, side, 3) and np.array_equal(ref, dst[i]), (level, side, i)
                rgba = np.asarray(mod.OpenSlide(path).read_region((x, y), lv, (side, side)))
                partial += int(((rgba[..., 3] > 0) & (rgba[..., 3] < 255)).sum())
                black += int((rgba[..., 3] == 0).sum())
                checked += 1
        # rows that are not one level / one size are left to the per-tile path
        assert wsi.read_tiles_into([[0, 0, 256, 256, 0], [0, 0, 256, 256, 1]], 0, 256) is False
        assert wsi.read_tiles_into([[0, 0, 256, 128, 0]], 0, 256) is False
        assert wsi.read_tiles_into([], 0, 256) is True
        wsi.cleanup()
        print(json.dumps(dict(checked=checked, partial=partial, black=black)))
        """, {"ATLASPATCH_LIBOPENSLIDE": lib}, tmp_path)
    assert out["checked"] == 52 and out["partial"] > 10000 and out["black"] > 10000       # both alpha branches were exercised


def test_without_libopenslide_the_hook_reports_unsupported_and_the_backend_keeps_the_per_tile_path(tmp_path):
    from tools import stub_openslide as so
    lib = so.build(str(tmp_path))
    out = _run_py(f"""
        import ctypes as C, json
        from tools import stub_openslide as so
        from atlaspatch_amd import _lib
        from atlaspatch_amd.core.wsi import openslide_wsi
        L = _lib.load()
        avail = L.ap_host_openslide_available()
        h = C.c_void_p()
        code = L.ap_host_openslide_open(b"/nonexistent.svs", C.byref(h))
        openslide_wsi.openslide = so.python_module({lib!r})          # openslide-python "installed", the C library not resolvable
        path = so.write_slide({str(tmp_path / 's.svs')!r}, 3000, 2000)
        wsi = openslide_wsi.OpenSlideWSI(path)
        ok = wsi.read_tiles_into([[0, 0, 256, 256, 0]], 0, 256)
        tile = wsi.extract((0, 0), 0, (256, 256))
        print(json.dumps(dict(avail=avail, code=code, ok=bool(ok), err=L.ap_last_error().decode(), shape=list(tile.shape))))
        """, {}, tmp_path)
    assert out["avail"] == 0 and out["code"] == -4 and out["ok"] is False and "libopenslide" in out["err"]
    assert out["shape"] == [256, 256, 3]
    bad = _run_py("""
        import json
        from atlaspatch_amd import _lib
        L = _lib.load()
        print(json.dumps(dict(avail=L.ap_host_openslide_available(), err=L.ap_last_error().decode())))
        """, {"ATLASPATCH_LIBOPENSLIDE": str(tmp_path / "missing.so")}, tmp_path)
    assert bad["avail"] == 0 and "does not load" in bad["err"]          # an explicit path never falls through to the system's


@pytest.mark.gpu
def test_process_on_an_openslide_slide_native_hook_equals_the_per_tile_path(tmp_path):
    """`process` on a slide served by (stub) OpenSlide, twice: tiles through ap_host_openslide_read_tiles on the ring's
    pinned threads, and through per-tile read_region(...).convert("RGB").  Same coords, same features, bit for bit; in
    native mode the embedding phase makes no read_region call at all."""
    from tools import stub_openslide as so
    lib = so.build(str(tmp_path))
    code = f"""
        import json, os, sys, numpy as np
        from click.testing import CliRunner
        from tools import stub_openslide as so
        from atlaspatch_amd.cli import cli
        from atlaspatch_amd.core.wsi import openslide_wsi
        from atlaspatch_amd.utils.h5 import h5
        mod = so.python_module({lib!r})
        openslide_wsi.openslide = mod
        path = so.write_slide({str(tmp_path / 'slide.svs')!r}, 14000, 10000, seed=5, alpha_period=3)
        out = sys.argv[1]
        res = CliRunner().invoke(cli, ["segment-and-get-coords", path, "-o", out, "--patch-size", "256", "--target-mag", "20"],
                                 catch_exceptions=False)
        assert res.exit_code == 0, res.output
        seg_calls = mod.calls["read_region"]
        res = CliRunner().invoke(cli, ["process", path, "-o", out, "--patch-size", "256", "--target-mag", "20",
                                       "--feature-extractors", "vit_b_16", "--feature-precision", "float16",
                                       "--feature-num-workers", "8"], catch_exceptions=False)
        assert res.exit_code == 0 and "failures: 0" in res.output, res.output
        with h5.File(os.path.join(out, "patches", "slide.h5"), "r") as f:
            coords, feats = f["coords"][:], f["features"]["vit_b_16"][:]
        np.save(os.path.join(out, "coords.npy"), coords); np.save(os.path.join(out, "feats.npy"), feats)
        print(json.dumps(dict(n=int(coords.shape[0]), embed_calls=mod.calls["read_region"] - seg_calls)))
        """
    runs = {}
    for mode in ("1", "0"):
        out_dir = tmp_path / f"out{mode}"
        env = {k: v for k, v in os.environ.items()}
        env.update(ATLASPATCH_LIBOPENSLIDE=lib, ATLASPATCH_OPENSLIDE_NATIVE=mode, ATLASPATCH_RANDOM_INIT="4")
        res = subprocess.run([sys.executable, "-c", textwrap.dedent(code), str(out_dir)], capture_output=True, text=True, env=env,
                             cwd=ROOT, timeout=900)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
        runs[mode] = (json.loads(res.stdout.strip().splitlines()[-1]), np.load(out_dir / "coords.npy"), np.load(out_dir / "feats.npy"))
    (native, c1, f1), (pertile, c0, f0) = runs["1"], runs["0"]
    assert native["n"] == pertile["n"] > 100
    # the existing coords H5 is reused by `process` (skip-existing): no thumbnail read either, so native mode reads nothing per tile
    assert native["embed_calls"] == 0 and pertile["embed_calls"] >= pertile["n"]
    assert np.array_equal(c1, c0) and np.array_equal(f1, f0)
    assert np.isfinite(f1).all() and np.abs(f1).max() > 0


def test_decode_threads_racing_on_a_slides_first_batch_share_one_native_handle(tmp_path):
    """All TileRing decode threads call read_tiles_into on the first batch of a slide: the lazily opened ap_openslide handle
    must be opened once (an unguarded check leaked one libopenslide handle per thread)."""
    from tools import stub_openslide as so
    lib = so.build(str(tmp_path))
    out = _run_py(f"""
        import json, threading
        from tools import stub_openslide as so
        from atlaspatch_amd.core.wsi import openslide_wsi
        openslide_wsi.openslide = so.python_module({lib!r})
        path = so.write_slide({str(tmp_path / 's.svs')!r}, 4000, 3000, seed=5, alpha_period=0)
        distinct = 0
        for trial in range(20):
            wsi = openslide_wsi.OpenSlideWSI(path)
            wsi._ensure_loaded()
            gate = threading.Barrier(16)
            seen = []
            def go():
                gate.wait()
                seen.append(wsi._native_handle().value)
            ts = [threading.Thread(target=go) for _ in range(16)]
            [t.start() for t in ts]; [t.join() for t in ts]
            distinct = max(distinct, len(set(seen)))
            wsi.cleanup()
        print(json.dumps(dict(distinct=distinct)))
        """, {"ATLASPATCH_LIBOPENSLIDE": lib}, tmp_path)
    assert out["distinct"] == 1
