#!/opt/conda/bin/python3.9
"""Structural dump of an HDF5 file through the REAL h5py (conda python of the image): same dump as gen_golden_h5_real.py
writes for the reference's file.  usage: describe_h5_real.py <file.h5>  -> JSON on stdout"""
import json
import sys

import h5py


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


if __name__ == "__main__":
    print(json.dumps(describe(sys.argv[1]), sort_keys=True, default=str))
