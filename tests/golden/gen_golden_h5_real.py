#!/opt/conda/bin/python3.9
"""G5b: a REAL HDF5 file written by the reference's own H5PatchWriter (services/storage.py + utils/h5.py, unmodified) with
the real h5py of the image's conda environment (/opt/conda/bin/python3.9: h5py 3.3.0, numpy 1.26; no torch there, none
needed).  tests/golden/gen_golden.py records the layout through a fake h5py (the main interpreter has none); this script
pins the same thing at the file level: dataset names, dtypes, shapes, chunking, max shapes, fill of the fixed-width
passport strings, attribute names / types / values.

    /opt/conda/bin/python3.9 tests/golden/gen_golden_h5_real.py

Writes tests/golden/reference_real.h5 (+ reference_real_h5.json: the structural dump h5py gives of it).  Only the two
reference modules named above are imported, through stub parent packages (their package __init__ files import cv2 etc.).
"""
import json
import os
import sys
import types
from pathlib import Path

import h5py
import numpy as np

REF = "/root/reference"
HERE = Path(__file__).resolve().parent


def _stub_packages():
    for name in ("atlas_patch", "atlas_patch.services", "atlas_patch.utils"):
        mod = types.ModuleType(name)
        mod.__path__ = [os.path.join(REF, *name.split("."))]
        sys.modules[name] = mod


def describe(path):
    out = {"datasets": {}, "file_attrs": {}}
    with h5py.File(path, "r") as f:
        def visit(name, obj):
            if isinstance(obj, h5py.Dataset):
                out["datasets"][name] = {
                    "shape": list(obj.shape), "dtype": obj.dtype.str, "chunks": list(obj.chunks) if obj.chunks else None,
                    "maxshape": [None if m is None else int(m) for m in obj.maxshape],
                    "compression": obj.compression, "fillvalue": repr(obj.fillvalue),
                    "attrs": {k: [type(v).__name__, getattr(v, "dtype", None) and v.dtype.str, v.tolist() if hasattr(v, "tolist") else v]
                              for k, v in obj.attrs.items()}}
        f.visititems(visit)
        for k, v in f.attrs.items():
            out["file_attrs"][k] = [type(v).__name__, getattr(v, "dtype", None) and v.dtype.str, v.tolist() if hasattr(v, "tolist") else v]
    return out


def main():
    _stub_packages()
    import importlib
    storage = importlib.import_module("atlas_patch.services.storage")
    rng = np.random.default_rng(5)
    n = 23
    xs = rng.integers(0, 90000, n) // 256 * 256
    ys = rng.integers(0, 90000, n) // 256 * 256
    entries = [(int(x), int(y), 512, 512, 1, None) for x, y in zip(xs, ys)]
    out = HERE / "reference_real.h5"
    if out.exists():
        out.unlink()
    w = storage.H5PatchWriter(chunk_rows=8, patch_size=256, patch_size_level0=512, level0_mag=40, target_mag=20,
                              level0_wh=(100000, 90000), overlap=0, slide_stem="slide_A", wsi_path="/data/slide_A.svs",
                              total_patches=None, extra_file_attrs={"mpp": 0.2528})
    total, _ = w.write_coords(out, entries, batch=10)
    assert total == n
    feats = rng.standard_normal((n, 12)).astype(np.float32)
    patches = [np.full((2, 2, 3), i, dtype=np.uint8) for i in range(n)]       # the index rides in the pixel value
    entries2 = [(e[0], e[1], e[2], e[3], e[4], p) for e, p in zip(entries, patches)]

    def feature_fn(batch):
        idx = [int(p[0, 0, 0]) for p in batch]
        return feats[idx]

    wrote = w.append_features(output_path=out, entries=entries2, feature_name="tiny12", feature_fn=feature_fn,
                              feature_attrs={"embedding_dim": 12, "source": "unit"}, feature_batch=7, expected_total=n)
    assert wrote == n
    d = describe(out)
    for k in ("creation_date",):
        d["file_attrs"].pop(k, None)
    (HERE / "reference_real_h5.json").write_text(json.dumps(d, indent=1, sort_keys=True, default=str))
    np.savez_compressed(HERE / "reference_real_h5_inputs.npz", coords=np.array([e[:5] for e in entries], dtype=np.int32), feats=feats)
    print("wrote", out, os.path.getsize(out), "bytes;", len(d["datasets"]), "datasets")


if __name__ == "__main__":
    main()
