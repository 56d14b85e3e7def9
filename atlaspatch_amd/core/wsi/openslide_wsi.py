"""OpenSlide backend, active only when ``openslide`` is importable (it is not in the build image).

Mirrors /root/reference/atlas_patch/core/wsi/openslide_wsi.py:71-205: MPP from metadata keys /
free text / TIFF resolution / 10 / magnification (rounded to 4 decimals), magnification from the
objective-power property or inferred from MPP, ``read_region(...).convert("RGB")`` tiles.
"""
from __future__ import annotations

import re
import threading
from typing import Literal, Optional, Tuple, Union

import numpy as np
from PIL import Image

from .iwsi import IWSI

try:  # pragma: no cover - optional dependency
    import openslide
except Exception:  # noqa: BLE001
    openslide = None

# key tables and their priority order as the reference holds them (openslide_wsi.py:19-32)
_MPP_KEYS = ("openslide.mpp-x", "openslide.mpp-y", "openslide.mirax.MPP", "aperio.MPP", "hamamatsu.XResolution")
_MPP_TEXT_KEYS = ("openslide.comment", "tiff.ImageDescription")
_MAG_KEYS = ("aperio.AppMag", "openslide.objective-power", "hamamatsu.SourceLens")
_MPP_PATTERNS = (r"mpp\s*[:=]\s*([0-9]*\.?[0-9]+)", r"microns?\s+per\s+pixel[^0-9]*([0-9]*\.?[0-9]+)")


def _mpp_from_text(text) -> Optional[float]:
    """openslide_wsi.py:149-182: first pattern that matches AND parses."""
    if not text:
        return None
    for pattern in _MPP_PATTERNS:
        match = re.search(pattern, text, flags=re.IGNORECASE)
        if match:
            try:
                return float(match.group(1))
            except ValueError:
                continue
    return None


def mpp_from_properties(meta: dict) -> Optional[float]:
    """MPP lookup of the reference (openslide_wsi.py:71-128), on a plain property dict: direct keys in priority
    order -> free-text fields -> TIFF resolution -> 10 / magnification; rounded to 4 decimals."""
    for key in _MPP_KEYS:
        if key in meta:
            try:
                return round(float(meta[key]), 4)
            except (TypeError, ValueError):
                continue
    for key in _MPP_TEXT_KEYS:
        parsed = _mpp_from_text(meta.get(key))
        if parsed is not None:
            return round(parsed, 4)
    try:
        res, unit = meta.get("tiff.XResolution"), meta.get("tiff.ResolutionUnit")
        if res and unit:
            res_f = float(res)
            if unit.lower() == "centimeter":
                return round(10000 / res_f, 4)
            elif unit.lower() == "inch":
                return round(25400 / res_f, 4)
    except (TypeError, ValueError):
        pass
    for key in _MAG_KEYS:
        value = meta.get(key)
        if value is not None:
            try:
                mag = float(value)
                if mag > 0:
                    return round(10.0 / mag, 4)
            except (TypeError, ValueError):
                continue
    return None


def mag_from_properties(meta: dict, mpp: Optional[float], infer_mag) -> Optional[int]:
    """Magnification (openslide_wsi.py:130-147): the objective-power property, else inferred from MPP."""
    power = meta.get("openslide.objective-power")
    if power:
        try:
            return int(float(power))
        except (TypeError, ValueError):
            pass
    if mpp is not None:
        try:
            return infer_mag(mpp)
        except ValueError:
            pass
    return None


class OpenSlideWSI(IWSI):
    def __init__(self, path: str, mpp: Optional[float] = None, **_: object) -> None:
        if openslide is None:
            raise RuntimeError("openslide-python is not installed; the OpenSlide backend is unavailable")
        super().__init__(path=path, mpp=mpp)
        self._slide = None
        self._native = None            # ap_openslide handle of the batched native reader; False = not available
        self._native_lock = threading.Lock()

    def _setup(self) -> None:
        self._slide = openslide.OpenSlide(self.path)
        self.w, self.h = self._slide.dimensions
        self.nlvl = self._slide.level_count
        self.ds = [float(d) for d in self._slide.level_downsamples]
        self.dims = [tuple(d) for d in self._slide.level_dimensions]
        self.meta = dict(self._slide.properties)
        if self._mpp_manual is not None:
            self.mpp = self.validate_mpp(float(self._mpp_manual), source="user-provided mpp")
        else:
            found = self._extract_mpp()
            self.mpp = self.validate_mpp(found, source="slide metadata") if found is not None else None
        self.mag = self._extract_mag()

    def _extract_mpp(self) -> Optional[float]:
        return mpp_from_properties(self.meta or {})

    def _extract_mag(self) -> Optional[int]:
        return mag_from_properties(self.meta or {}, self.mpp, self._infer_mag)

    def extract(self, xy: Tuple[int, int], lv: int, wh: Tuple[int, int], *,
                mode: Literal["array", "image"] = "array") -> Union[np.ndarray, Image.Image]:
        self._ensure_loaded()
        region = self._slide.read_region(xy, lv, wh).convert("RGB")
        if mode == "image":
            return region
        if mode == "array":
            return np.array(region)
        raise ValueError(f"Invalid mode: {mode}")

    def read_tiles_into(self, rows, dst_ptr: int, tile_side: int) -> bool:
        """Optional IWSI capability (native batched host read): the tiles of ``rows`` (x, y, rw, rh, lv) are read by
        libopenslide itself and converted to RGB into consecutive ``tile_side^2 * 3``-byte slots at ``dst_ptr`` in ONE call
        outside the interpreter lock (``ap_host_openslide_read_tiles``) -- the same pixels as ``extract`` =
        ``read_region(...).convert("RGB")`` (openslide_wsi.py:184-205).  False when libopenslide does not resolve or the
        rows are not one level / one square size: the ring then reads tile by tile through openslide-python."""
        import ctypes as C
        import os
        from ... import _lib
        self._ensure_loaded()
        if not rows:
            return True
        lv0 = int(rows[0][4])
        if any(int(r[2]) != tile_side or int(r[3]) != tile_side or int(r[4]) != lv0 for r in rows):
            return False
        lib = _lib.load()
        handle = self._native_handle()
        if handle is None:                 # no libopenslide on this host (or it cannot open what openslide-python opened)
            return False
        xy = np.ascontiguousarray([[r[0], r[1]] for r in rows], dtype=np.int64)
        _lib.check(lib.ap_host_openslide_read_tiles(handle, xy.ctypes.data, len(rows), lv0, tile_side, tile_side, dst_ptr),
                   "ap_host_openslide_read_tiles")
        return True

    def _native_handle(self):
        """ap_openslide handle of the batched native reader, or None when libopenslide does not resolve."""
        import ctypes as C
        import os
        from ... import _lib
        if self._native is False or os.environ.get("ATLASPATCH_OPENSLIDE_NATIVE", "1") == "0":
            return None
        if self._native is None:
            # every TileRing decode thread arrives here on a slide's first batch: one opens, the others wait (an unguarded
            # check would open one libopenslide handle -- descriptors + tile cache -- per thread and leak all but the last)
            with self._native_lock:
                if self._native is None:
                    handle = C.c_void_p()
                    if _lib.load().ap_host_openslide_open(str(self.path).encode(), C.byref(handle)) != _lib.AP_OK:
                        self._native = False
                    else:
                        self._native = handle
        return self._native or None

    def read_level_device(self, level: int, wh, device):
        """Whole-level read for the thumbnail (iwsi.py:296-303 is ONE read_region of the full level on one thread):
        full-width strips read by libopenslide on a thread pool outside the interpreter lock (the strips of one image are
        consecutive ``w * rows * 3``-byte slots, i.e. exactly the image), into pinned memory, then one H2D copy."""
        import concurrent.futures as futures
        import os
        import torch
        from ... import _lib
        self._ensure_loaded()
        handle = self._native_handle()
        if handle is None:
            return None
        w, h = int(wh[0]), int(wh[1])
        ds = float(self.ds[level])
        rows = max(16, min(256, -(-h // 64)))               # strip height: ~64 strips per level, 16..256 rows each
        if ds != float(int(ds)):
            # read_region takes LEVEL-0 coordinates and divides by the level's downsample: with a non-integer downsample
            # (4.0001...) a strip that starts at level row y cannot be addressed exactly (round(y * ds) / ds != y), and a
            # sub-pixel offset makes OpenSlide resample.  Such levels are read in ONE region like the reference does --
            # still outside the interpreter lock, just not in parallel.
            rows = h
        host = torch.empty((h, w, 3), dtype=torch.uint8, pin_memory=True)
        base = host.data_ptr()
        lib = _lib.load()
        strips = [(y, min(rows, h - y)) for y in range(0, h, rows)]
        tail = [s for s in strips if s[1] != rows]
        full = [s for s in strips if s[1] == rows]

        def read(group):
            # level-0 y of a strip that starts at level row y (openslide's read_region takes level-0 coordinates)
            xy = np.ascontiguousarray([[0, y * int(ds)] for y, _ in group], dtype=np.int64)
            for (y, n), pt in zip(group, xy):
                _lib.check(lib.ap_host_openslide_read_tiles(handle, pt.ctypes.data, 1, int(level), w, n,
                                                            base + y * w * 3), "ap_host_openslide_read_tiles")

        workers = max(1, min(32, len(strips), os.cpu_count() or 8))
        chunks = [full[i::workers] for i in range(workers) if full[i::workers]] + ([tail] if tail else [])
        with futures.ThreadPoolExecutor(max_workers=workers, thread_name_prefix="level") as pool:
            list(pool.map(read, chunks))
        return host.to(torch.device(device), non_blocking=False)

    def get_size(self, lv: int = 0) -> Tuple[int, int]:
        self._ensure_loaded()
        if lv < 0 or lv >= self.nlvl:
            raise IndexError(f"Level {lv} out of range")
        return self.dims[lv]

    def get_thumb(self, max_hw: Tuple[int, int]) -> Image.Image:
        self._ensure_loaded()
        return self._slide.get_thumbnail(max_hw).convert("RGB")

    def cleanup(self) -> None:
        if self._native not in (None, False):
            from ... import _lib
            _lib.load().ap_host_openslide_close(self._native)
        self._native = None
        if self._slide is not None:
            try:
                self._slide.close()
            finally:
                self._slide = None
        self._loaded = False
