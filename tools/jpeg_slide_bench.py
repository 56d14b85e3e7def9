#!/usr/bin/env python3
"""usage: jpeg_slide_bench.py [slide_side=40000] [workers=64] [encoder=vit_b_16] [codec=jpg|z]
codec z = raw RGB deflated with zlib level 1: a decoder that releases the interpreter lock for the whole tile (like OpenSlide).
End-to-end `process` on a synthetic slide whose tiles are stored as JPEG files (quality 80) and decoded with Pillow by
the tile ring's host threads -- the stand-in for a real slide's compressed tiles.  Reports the
decode-only rate of the host threads, and the end-to-end rate (decode -> pinned ring -> H2D -> forward -> H5)."""
import concurrent.futures as futures, json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ATLASPATCH_RANDOM_INIT", "0")
import numpy as np, torch
from PIL import Image
from click.testing import CliRunner
from atlaspatch_amd import _lib
from atlaspatch_amd.cli import cli
from atlaspatch_amd.core.wsi.synth_pixels import SynthSpec, analytic_mask
from atlaspatch_amd.services.extraction import coords_from_mask
from atlaspatch_amd.utils.h5 import h5
side = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 64
enc = sys.argv[3] if len(sys.argv) > 3 else "vit_b_16"
codec = sys.argv[4] if len(sys.argv) > 4 else "jpg"
import zlib
dev = torch.device("cuda:0")
spec = SynthSpec(width=side, height=side, seed=1234)
coords, _ = coords_from_mask(analytic_mask(spec), level0_wh=(side, side), downsamples=list(spec.downsamples), src_mag=spec.mag,
                             tgt_mag=spec.mag, patch_size=256, step_size=None, tissue_thresh=0.0)
with tempfile.TemporaryDirectory() as tmp:
    store = os.path.join(tmp, "tiles"); os.makedirs(store)
    # build the store: pixels from the device generator, JPEG-encoded by a thread pool (untimed)
    lib = _lib.load()
    ell = torch.from_numpy(spec.ellipses()).to(dev)
    t0 = time.perf_counter()
    with futures.ThreadPoolExecutor(workers) as pool:
        for lo in range(0, len(coords), 2048):
            xy = torch.from_numpy(np.ascontiguousarray(coords[lo:lo + 2048, :2], dtype=np.int32)).to(dev)
            tiles = torch.empty((xy.shape[0], 256, 256, 3), dtype=torch.uint8, device=dev)
            _lib.check(lib.ap_synth_tiles(xy.data_ptr(), xy.shape[0], 256, 1, 0, side, side, spec.seed, ell.data_ptr(), ell.shape[0],
                                          tiles.data_ptr(), _lib.current_stream_ptr(dev)))
            host = tiles.cpu().numpy()
            def put(i):
                stem = os.path.join(store, f"{coords[lo + i, 0]}_{coords[lo + i, 1]}_256")
                if codec == "z":
                    open(stem + ".z", "wb").write(zlib.compress(host[i].tobytes(), 1))
                else:
                    Image.fromarray(host[i]).save(stem + ".jpg", quality=80)
            list(pool.map(put, range(host.shape[0])))
    build_s = time.perf_counter() - t0
    size_mb = sum(os.path.getsize(os.path.join(store, f)) for f in os.listdir(store)) / 1e6
    slide = os.path.join(tmp, "big.synth")
    json.dump({"width": side, "height": side, "seed": 1234, "mag": 20, "mpp": 0.5, "downsamples": [1, 4, 16], "jpeg_tiles": "tiles"},
              open(slide, "w"))
    # decode-only rate of the host threads
    names = [os.path.join(store, f"{x}_{y}_256.{codec}") for x, y in coords[:, :2]]
    def dec(p):
        if codec == "z":
            return len(zlib.decompress(open(p, "rb").read()))
        with Image.open(p) as im:
            return np.asarray(im.convert("RGB")).shape[0]
    t0 = time.perf_counter()
    with futures.ThreadPoolExecutor(workers) as pool:
        list(pool.map(dec, names, chunksize=64))
    decode_rate = len(names) / (time.perf_counter() - t0)
    # end to end
    from safetensors.torch import save_file
    from atlaspatch_amd.encoders.vit import ARCHS, random_canonical_state_dict
    save_file(random_canonical_state_dict(ARCHS[enc], 0), os.path.join(tmp, f"{enc}.safetensors"))
    os.environ["ATLASPATCH_WEIGHTS_DIR"] = tmp
    os.environ.pop("ATLASPATCH_RANDOM_INIT", None)
    t0 = time.perf_counter()
    res = CliRunner().invoke(cli, ["process", slide, "-o", os.path.join(tmp, "out"), "--patch-size", "256", "--target-mag", "20",
                                   "--feature-extractors", enc, "--feature-precision", "float16", "--feature-num-workers", str(workers)],
                             catch_exceptions=False)
    dt = time.perf_counter() - t0
    assert res.exit_code == 0 and "failures: 0" in res.output, res.output
    with h5.File(os.path.join(tmp, "out", "patches", "big.h5"), "r") as f:
        n = f["coords"].shape[0]
        finite = bool(np.isfinite(f["features"][enc][:]).all())
print(json.dumps({"encoder": enc, "side": side, "tiles": int(n), "codec": codec, "store_MB": round(size_mb, 1), "store_build_s": round(build_s, 1),
                  "host_threads_used": workers, "decode_only_tiles_per_s": round(decode_rate, 1),
                  "end_to_end_seconds": round(dt, 2), "end_to_end_tiles_per_s": round(n / dt, 1), "features_finite": finite}))
