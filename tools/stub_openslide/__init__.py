"""Stand-in OpenSlide stack for tests and the bench (neither libopenslide nor openslide-python is in this image).

``build(dir)`` compiles ``stub_openslide.c`` into ``<dir>/libopenslide.so.1`` (interface of libopenslide, synthetic pixels
with partial alpha).  ``python_module(lib_path)`` returns an object with openslide-python's surface (``OpenSlide`` with
``dimensions / level_count / level_downsamples / level_dimensions / properties / read_region / get_thumbnail / close``)
that binds that library through ctypes the way openslide-python's ``lowlevel`` does, including its ARGB -> RGBA conversion
(``_convert.argb2rgba``: alpha 0 untouched, alpha 255 reordered, else ``255 * c // alpha``) -- so ``read_region(...)
.convert("RGB")`` is the per-tile reference path the native hook must equal bit for bit.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import types

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def build(out_dir: str) -> str:
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(out_dir, "libopenslide.so.1")
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", lib, os.path.join(HERE, "stub_openslide.c")], check=True)
    return lib


def write_slide(path: str, width: int, height: int, seed: int = 1, alpha_period: int = 3) -> str:
    with open(path, "w") as fh:
        fh.write(f"STUBSLIDE {int(width)} {int(height)} {int(seed)} {int(alpha_period)}\n")
    return path


def argb2rgba(buf: np.ndarray) -> np.ndarray:
    """openslide-python's _convert.argb2rgba on a uint32 array (little-endian host) -> uint8 [..., 4] RGBA bytes."""
    v = buf.astype(np.uint32)
    a = v >> 24
    r, g, b = (v >> 16) & 0xFF, (v >> 8) & 0xFF, v & 0xFF
    safe = np.maximum(a, 1)
    part = (a != 0) & (a != 255)
    r2 = np.where(part, (255 * r // safe) & 0xFF, r)
    g2 = np.where(part, (255 * g // safe) & 0xFF, g)
    b2 = np.where(part, (255 * b // safe) & 0xFF, b)
    out = np.stack([r2, g2, b2, a], -1).astype(np.uint8)
    zero = a == 0                       # untouched word: its little-endian bytes (B, G, R, A) read as R, G, B, A
    out[zero] = np.stack([b, g, r, a], -1).astype(np.uint8)[zero]
    return out


def python_module(lib_path: str):
    lib = C.CDLL(lib_path)
    lib.openslide_open.restype = C.c_void_p
    lib.openslide_open.argtypes = [C.c_char_p]
    lib.openslide_close.argtypes = [C.c_void_p]
    lib.openslide_get_level_count.argtypes = [C.c_void_p]
    lib.openslide_get_level_downsample.restype = C.c_double
    lib.openslide_get_level_downsample.argtypes = [C.c_void_p, C.c_int32]
    lib.openslide_get_level_dimensions.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.openslide_read_region.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int64, C.c_int64]
    calls = {"read_region": 0}

    class OpenSlide:
        def __init__(self, path):
            self._h = lib.openslide_open(str(path).encode())
            if not self._h:
                raise OSError(f"unsupported or missing slide: {path}")
            self.level_count = int(lib.openslide_get_level_count(self._h))
            dims = []
            for lv in range(self.level_count):
                w, h = C.c_int64(), C.c_int64()
                lib.openslide_get_level_dimensions(self._h, lv, C.byref(w), C.byref(h))
                dims.append((int(w.value), int(h.value)))
            self.level_dimensions = tuple(dims)
            self.dimensions = dims[0]
            self.level_downsamples = tuple(float(lib.openslide_get_level_downsample(self._h, lv)) for lv in range(self.level_count))
            self.properties = {"openslide.mpp-x": "0.5", "openslide.mpp-y": "0.5", "openslide.objective-power": "20",
                               "openslide.vendor": "stub"}

        def read_region(self, location, level, size):
            calls["read_region"] += 1
            w, h = int(size[0]), int(size[1])
            buf = np.empty((h, w), dtype=np.uint32)
            lib.openslide_read_region(self._h, buf.ctypes.data, int(location[0]), int(location[1]), int(level), w, h)
            return Image.fromarray(argb2rgba(buf), "RGBA")

        def get_thumbnail(self, size):
            lv = self.level_count - 1
            im = self.read_region((0, 0), lv, self.level_dimensions[lv])
            im.thumbnail(size)
            return im

        def close(self):
            if self._h:
                lib.openslide_close(self._h)
                self._h = None

    return types.SimpleNamespace(OpenSlide=OpenSlide, calls=calls, lib=lib)
