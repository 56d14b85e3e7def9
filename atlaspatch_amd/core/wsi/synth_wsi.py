"""Synthetic slide backend (``.synth`` descriptor files), SURVEY.md section 8d.

A ``.synth`` file is a small JSON object ``{"width", "height", "seed", "n_ellipses",
"mag", "mpp", "downsamples"}``.  Pixels are a pure function of those numbers
(``synth_pixels.render_region``), so a 100 000 x 100 000 slide costs nothing to
store and any tile can be produced on the host (this class) or directly in HBM
(csrc/synth.hip, bit-identical).  Registered with ``WSIFactory`` under backend
name ``synth``, the same seam the reference offers (wsi_factory.py:41-53).

Optional key ``"jpeg_tiles": "<directory>"``: level-0 tiles found there as ``<x>_<y>_<size>.jpg`` are DECODED with
Pillow (or, as ``<x>_<y>_<size>.z``, inflated with zlib: raw RGB) instead of rendered -- a stand-in for a real slide's compressed tiles (SURVEY 8d "optional JPEG-tile store"),
so the tile ring can be measured against a realistic host decoder (libjpeg releases the interpreter lock); such a
slide never serves tiles from the device.  ``tools/jpeg_slide_bench.py`` builds a store.
"""
from __future__ import annotations

import json
import os
from typing import Literal, Optional, Tuple, Union

import numpy as np
from PIL import Image

from .iwsi import IWSI
from .synth_pixels import SynthSpec, analytic_mask, render_region


def load_spec(path: str) -> SynthSpec:
    with open(path, "r", encoding="utf-8") as handle:
        raw = json.load(handle)
    return SynthSpec(
        width=int(raw["width"]), height=int(raw["height"]), seed=int(raw.get("seed", 1234)),
        n_ellipses=int(raw.get("n_ellipses", 12)), mag=int(raw.get("mag", 20)),
        mpp=float(raw.get("mpp", 0.5)),
        downsamples=tuple(float(d) for d in raw.get("downsamples", (1.0, 4.0, 16.0))))


def write_spec(path: str, spec: SynthSpec) -> None:
    with open(path, "w", encoding="utf-8") as handle:
        json.dump({"width": spec.width, "height": spec.height, "seed": spec.seed,
                   "n_ellipses": spec.n_ellipses, "mag": spec.mag, "mpp": spec.mpp,
                   "downsamples": list(spec.downsamples)}, handle)


class SynthWSI(IWSI):
    def __init__(self, path: str, mpp: Optional[float] = None, **_: object) -> None:
        super().__init__(path=path, mpp=mpp)
        self.spec: Optional[SynthSpec] = None
        self.jpeg_dir: Optional[str] = None

    def _setup(self) -> None:
        spec = load_spec(self.path)
        self.spec = spec
        with open(self.path, "r", encoding="utf-8") as handle:
            jd = json.load(handle).get("jpeg_tiles")
        if jd:
            self.jpeg_dir = jd if os.path.isabs(jd) else os.path.join(os.path.dirname(os.path.abspath(self.path)), jd)
        self.w, self.h = spec.width, spec.height
        self.ds = [float(d) for d in spec.downsamples]
        self.nlvl = len(self.ds)
        self.dims = [(int(round(spec.width / d)), int(round(spec.height / d))) for d in self.ds]
        self.meta = {"openslide.vendor": "synthetic", "synth.seed": str(spec.seed)}
        self.mpp = self._extract_mpp()
        self.mag = self._extract_mag()

    def _extract_mpp(self) -> Optional[float]:
        if self._mpp_manual is not None:
            return self.validate_mpp(float(self._mpp_manual), source="user-provided mpp")
        return self.spec.mpp if self.spec else None

    def _extract_mag(self) -> Optional[int]:
        return int(self.spec.mag) if self.spec else None

    def extract(self, xy: Tuple[int, int], lv: int, wh: Tuple[int, int], *,
                mode: Literal["array", "image"] = "array") -> Union[np.ndarray, Image.Image]:
        self._ensure_loaded()
        if not 0 <= lv < (self.nlvl or 0):
            raise ValueError(f"Invalid level {lv}")
        region = None
        if self.jpeg_dir is not None and lv == 0 and wh[0] == wh[1]:
            stem = os.path.join(self.jpeg_dir, f"{int(xy[0])}_{int(xy[1])}_{int(wh[0])}")
            if os.path.exists(stem + ".z"):          # raw RGB, deflate: zlib.decompress runs without the interpreter lock
                import zlib
                with open(stem + ".z", "rb") as fh:
                    region = np.frombuffer(zlib.decompress(fh.read()), dtype=np.uint8).reshape(int(wh[1]), int(wh[0]), 3)
            elif os.path.exists(stem + ".jpg"):
                with Image.open(stem + ".jpg") as img:
                    region = np.asarray(img.convert("RGB"))
        if region is None:
            region = render_region(self.spec, int(xy[0]), int(xy[1]), int(wh[0]), int(wh[1]), int(lv))
        if mode == "array":
            return region
        if mode == "image":
            return Image.fromarray(region)
        raise ValueError(f"Invalid mode: {mode}")

    def extract_batch_device(self, rows: np.ndarray, device, patch_size: int):
        """Optional IWSI capability (device tile source): uint8 [n, read_h, read_w, 3] in HBM for coords rows
        (x, y, read_w, read_h, level) -- tiles at their READ size, like ``extract``; the caller applies the
        reference's ``cv2.resize`` to ``patch_size`` on the device when the two differ -- or None when the backend
        cannot serve them on the device.  The synthetic slide's pixel function runs as ``ap_synth_tiles``
        (bit-identical to ``render_region``, tested), which stands in for a GPU tile decoder; real backends decode
        on the host and go through the tile ring."""
        import torch
        from ... import _lib
        self._ensure_loaded()
        if self.jpeg_dir is not None:
            return None                      # compressed tiles are decoded on the host and cross the ring
        rows = np.asarray(rows)
        if rows.size == 0:
            return torch.empty((0, patch_size, patch_size, 3), dtype=torch.uint8, device=device)
        lv, side = int(rows[0, 4]), int(rows[0, 2])
        if np.any(rows[:, 4] != lv) or np.any(rows[:, 2] != side) or np.any(rows[:, 3] != side):
            return None
        lib = _lib.load()
        device = torch.device(device)
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        cache = getattr(self, "_dev_ellipses", None)
        if cache is None or cache.device != device:
            cache = torch.from_numpy(self.spec.ellipses()).to(device)
            self._dev_ellipses = cache
        xy = torch.from_numpy(np.ascontiguousarray(rows[:, :2], dtype=np.int32)).to(device)
        tiles = torch.empty((rows.shape[0], side, side, 3), dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            _lib.check(lib.ap_synth_tiles(xy.data_ptr(), rows.shape[0], side, int(round(self.ds[lv])), lv,
                                          self.spec.width, self.spec.height, self.spec.seed, cache.data_ptr(),
                                          cache.shape[0], tiles.data_ptr(), _lib.current_stream_ptr(device)),
                       "ap_synth_tiles")
        return tiles

    def read_level_device(self, level: int, wh, device):
        """The whole-level read of the thumbnail path, rendered in HBM by ``ap_synth_region`` (bit-identical to
        ``render_region``): on a 100 000 x 100 000 slide level 2 is 6250 x 6250 = 117 MB, 3.3 s through the NumPy renderer."""
        import torch
        from ... import _lib
        self._ensure_loaded()
        if self.jpeg_dir is not None:
            return None
        device = torch.device(device)
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        cache = getattr(self, "_dev_ellipses", None)
        if cache is None or cache.device != device:
            cache = torch.from_numpy(self.spec.ellipses()).to(device)
            self._dev_ellipses = cache
        w, h = int(wh[0]), int(wh[1])
        out = torch.empty((h, w, 3), dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            _lib.check(_lib.load().ap_synth_region(0, 0, w, h, int(round(self.ds[level])), int(level), self.spec.width,
                                                   self.spec.height, self.spec.seed, cache.data_ptr(), cache.shape[0],
                                                   out.data_ptr(), _lib.current_stream_ptr(device)), "ap_synth_region")
        return out

    def read_tiles_into(self, rows, dst_ptr: int, tile_side: int) -> bool:
        """Optional IWSI capability (native batched host decode): decode the tiles of ``rows`` (x, y, rw, rh, lv) into
        consecutive ``tile_side^2 * 3``-byte slots at ``dst_ptr`` in ONE call outside the interpreter lock, or return
        False when this backend cannot (the caller then reads tile by tile).  The deflate tile store is inflated by
        ``ap_host_inflate_tiles``; a plain synthetic slide is rendered by ``ap_host_synth_tiles`` (the C twin of
        ``render_region``) -- both stand in for a real format's native tile decoder."""
        import ctypes as C
        from ... import _lib
        self._ensure_loaded()
        if not rows:
            return True
        lv0 = int(rows[0][4])
        if any(int(r[2]) != tile_side or int(r[3]) != tile_side or int(r[4]) != lv0 for r in rows):
            return False
        if self.jpeg_dir is None:
            xy = np.ascontiguousarray([[r[0], r[1]] for r in rows], dtype=np.int32)
            ell = getattr(self, "_host_ellipses", None)
            if ell is None:
                ell = self._host_ellipses = np.ascontiguousarray(self.spec.ellipses(), dtype=np.int64)
            _lib.check(_lib.load().ap_host_synth_tiles(dst_ptr, xy.ctypes.data, len(rows), tile_side, int(round(self.ds[lv0])),
                                                       lv0, self.spec.width, self.spec.height, self.spec.seed,
                                                       ell.ctypes.data, ell.shape[0]), "ap_host_synth_tiles")
            return True
        lib = _lib.load()
        stems = [os.path.join(self.jpeg_dir, f"{int(x)}_{int(y)}_{int(rw)}") for x, y, rw, rh, lv in rows]
        if lv0 != 0:
            return False
        kind = getattr(self, "_store_kind", None)
        if kind is None:                                   # one probe per slide: the store holds one kind of file
            kind = self._store_kind = ".z" if os.path.exists(stems[0] + ".z") else ".jpg"
        if kind == ".jpg" and getattr(self, "_jpeg_native", True) is False:
            return False
        files = [s + kind for s in stems]
        if not all(os.path.exists(f) for f in files):      # a tile the store does not hold is rendered by the per-tile path
            return False
        arr = (C.c_char_p * len(files))(*[f.encode() for f in files])
        if kind == ".z":
            _lib.check(lib.ap_host_inflate_tiles(dst_ptr, arr, len(stems), tile_side * tile_side * 3), "ap_host_inflate_tiles")
            return True
        code = lib.ap_host_decode_jpeg_tiles(dst_ptr, arr, len(stems), tile_side)
        if code == _lib.AP_ERR_UNSUPPORTED:                # no usable libjpeg on this host: per-tile Pillow decode
            self._jpeg_native = False
            return False
        if code == _lib.AP_ERR_INVALID:
            # a tile the native decoder refuses (wrong size, or libjpeg WARNED: truncated / corrupt stream): this chunk goes
            # through the per-tile Pillow path, which raises or decodes exactly as the reference's reader would
            return False
        _lib.check(code, "ap_host_decode_jpeg_tiles")
        return True

    def get_size(self, lv: int = 0) -> Tuple[int, int]:
        self._ensure_loaded()
        return self.dims[lv]

    def get_thumb(self, max_hw: Tuple[int, int]) -> Image.Image:
        self._ensure_loaded()
        level = (self.nlvl or 1) - 1
        image = Image.fromarray(self.extract((0, 0), level, self.dims[level]))
        image.thumbnail(max_hw)
        return image

    def tissue_mask(self, thumb_max: int = 1024) -> np.ndarray:
        """Analytic tissue mask on the thumbnail grid (stands in for SAM2 on synthetic slides)."""
        self._ensure_loaded()
        return analytic_mask(self.spec, thumb_max)

    def cleanup(self) -> None:
        self._loaded = False
