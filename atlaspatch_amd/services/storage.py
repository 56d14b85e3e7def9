"""H5 output contract (reference: services/storage.py:14-392).

``<out>/patches/<stem>.h5``:
  coords      int32 [N, 5]   (x, y, read_w, read_h, level)  chunks (write_batch, 5), maxshape (None, 5)
  passports   S160  [N]      "{stem}__x{X}_y{Y}_rw{RW}_rh{RH}_lv{LV}_mag{MAG}_tmag{TMAG}_total{TOTAL}"
  features/<name>  float32 [N, D]  chunks (feature_batch, D), built under ``__tmp_<name>`` then moved
  file attrs  patch_size, patch_size_level0, level0_magnification, target_magnification, overlap,
              level0_width, level0_height, wsi_path, passport_format, passport_version=2,
              creation_date, filename, + IWSI.metadata_attrs(), num_patches

The iterator-driven ``write_coords`` / ``append_features`` keep the reference's signatures so
its services can call them; ``write_coords_array`` / ``append_feature_matrix`` are the bulk
entry points the device pipeline uses (same bytes on disk, no per-row Python).
"""
from __future__ import annotations

from datetime import datetime, timezone
from pathlib import Path
from typing import Any, Callable, Iterable, Mapping, Sequence

import numpy as np

from ..utils.h5 import H5AppendWriter, h5

PASSPORT_FORMAT = "{stem}__x{X}_y{Y}_rw{RW}_rh{RH}_lv{LV}_mag{MAG}_tmag{TMAG}_total{TOTAL}"


class H5PatchWriter:
    def __init__(self, *, chunk_rows: int, patch_size: int, patch_size_level0: int, level0_mag: int,
                 target_mag: int, level0_wh: tuple[int, int], overlap: int, slide_stem: str, wsi_path: str,
                 total_patches: int | None = None, extra_file_attrs: Mapping[str, Any] | None = None) -> None:
        self.chunk_rows = max(1, int(chunk_rows))
        self.patch_size = int(patch_size)
        self.patch_size_level0 = int(patch_size_level0)
        self.level0_mag = int(level0_mag)
        self.target_mag = int(target_mag)
        self.level0_wh = level0_wh
        self.overlap = int(overlap)
        self.slide_stem = slide_stem
        self.wsi_path = wsi_path
        self.total_patches = int(total_patches) if total_patches is not None else None
        self.extra_file_attrs = dict(extra_file_attrs) if extra_file_attrs else {}
        self._passport_dtype = np.dtype("S160")

    def to_kwargs(self) -> dict:
        """Constructor arguments of an equal writer (what a helper process needs: services/h5_writer_proc.py)."""
        return dict(chunk_rows=self.chunk_rows, patch_size=self.patch_size, patch_size_level0=self.patch_size_level0,
                    level0_mag=self.level0_mag, target_mag=self.target_mag, level0_wh=tuple(int(v) for v in self.level0_wh),
                    overlap=self.overlap, slide_stem=self.slide_stem, wsi_path=self.wsi_path, total_patches=self.total_patches,
                    extra_file_attrs=dict(self.extra_file_attrs))

    # ------------------------------------------------------------------ coords
    def _passport(self, x: int, y: int, rw: int, rh: int, lv: int) -> str:
        if self.total_patches is None:
            raise RuntimeError("total_patches must be set before generating passports")
        mag = self.level0_mag if self.level0_mag else "na"
        tmag = self.target_mag if self.target_mag else "na"
        return (f"{self.slide_stem}__x{x}_y{y}_rw{rw}_rh{rh}_lv{lv}_mag{mag}_tmag{tmag}"
                f"_total{self.total_patches}")

    def _passports(self, block: np.ndarray) -> np.ndarray:
        """S160 passports of a block of coords rows: formatted by ``ap_host_format_passports`` (same bytes as the per-row
        ``_passport`` f-string, tested); a stem that is not ASCII takes the per-row path, which fails where the
        reference's ``np.asarray(..., dtype="S160")`` fails."""
        from .. import _lib
        try:
            prefix = self.slide_stem.encode("ascii")
        except UnicodeEncodeError:
            return np.asarray([self._passport(*row) for row in block.tolist()], dtype=self._passport_dtype)
        if self.total_patches is None:
            raise RuntimeError("total_patches must be set before generating passports")
        mag = self.level0_mag if self.level0_mag else "na"
        tmag = self.target_mag if self.target_mag else "na"
        suffix = f"_mag{mag}_tmag{tmag}_total{self.total_patches}".encode("ascii")
        block = np.ascontiguousarray(block, dtype=np.int32).reshape(-1, 5)
        out = np.empty(block.shape[0], dtype=self._passport_dtype)
        _lib.check(_lib.load().ap_host_format_passports(block.ctypes.data, block.shape[0], prefix, suffix, out.ctypes.data,
                                                        self._passport_dtype.itemsize), "ap_host_format_passports")
        return out

    def _open_seeded(self, output_path: Path) -> H5AppendWriter:
        writer = H5AppendWriter(str(output_path), chunk_rows=self.chunk_rows)
        writer.append({"coords": np.empty((0, 5), dtype=np.int32),
                       "passports": np.empty((0,), dtype=self._passport_dtype)})
        width0, height0 = self.level0_wh
        attrs = {"patch_size": self.patch_size, "patch_size_level0": self.patch_size_level0,
                 "level0_magnification": self.level0_mag, "target_magnification": self.target_mag,
                 "overlap": self.overlap, "level0_width": int(width0), "level0_height": int(height0),
                 "wsi_path": self.wsi_path, "passport_format": PASSPORT_FORMAT, "passport_version": 2,
                 "creation_date": datetime.now(timezone.utc).isoformat()}
        attrs.update(self.extra_file_attrs)
        writer.update_file_attrs(attrs)
        return writer

    def passports_array(self, coords: np.ndarray) -> np.ndarray:
        """S160 passports of all rows (``total_patches`` = the row count), as ``write_coords_array`` writes them."""
        coords = np.ascontiguousarray(coords, dtype=np.int32).reshape(-1, 5)
        self.total_patches = int(coords.shape[0])
        if coords.shape[0] == 0:
            return np.empty((0,), dtype=self._passport_dtype)
        return self._passports(coords)

    def write_coords_array(self, output_path, coords: np.ndarray, passports: np.ndarray | None = None) -> int:
        """Bulk form of ``write_coords``: coords int32 [N, 5] already in the reference's order.  ``passports``: the S160
        strings of these rows when the caller has formatted them already (the helper processes of a cohort run)."""
        coords = np.ascontiguousarray(coords, dtype=np.int32).reshape(-1, 5)
        self.total_patches = int(coords.shape[0])
        if passports is not None and (passports.shape != (coords.shape[0],) or passports.dtype != self._passport_dtype):
            raise ValueError("passports must be S160 [N]")
        writer = self._open_seeded(Path(output_path))
        try:
            for start in range(0, coords.shape[0], self.chunk_rows):
                block = coords[start:start + self.chunk_rows]
                pp = self._passports(block) if passports is None else passports[start:start + self.chunk_rows]
                writer.append({"coords": block, "passports": pp})
            writer.update_file_attrs({"num_patches": int(coords.shape[0])})
            writer.close()
        except Exception:
            writer.abort()
            raise
        return int(coords.shape[0])

    def write_coords(self, output_path, entries: Iterable[tuple], *, batch: int,
                     collect_coords: bool = False):
        """Reference signature (storage.py:106-161): drains ``entries`` first (the passports embed
        the final total), then writes in chunks of ``batch`` rows."""
        rows = [(int(x), int(y), int(rw), int(rh), int(lv)) for x, y, rw, rh, lv, _ in entries]
        arr = np.asarray(rows, dtype=np.int32).reshape(-1, 5)
        keep = self.chunk_rows
        self.chunk_rows = self.chunk_rows          # dataset chunking stays chunk_rows; batch only paces appends
        total = self.write_coords_array(output_path, arr)
        self.chunk_rows = keep
        viz = arr[:, :2].copy() if collect_coords else None
        return total, viz

    # ------------------------------------------------------------------ features
    def append_features(self, *, output_path, entries: Iterable[tuple], feature_name: str,
                        feature_fn: Callable[[Sequence[np.ndarray]], np.ndarray],
                        feature_attrs: Mapping[str, int | str], feature_batch: int,
                        expected_total: int | None = None) -> int:
        """Reference signature (storage.py:250-337): buffers ``feature_batch`` patches, calls
        ``feature_fn`` per batch, validates, grows ``features/__tmp_<name>``, then moves it."""
        batch = max(1, int(feature_batch))

        def blocks():
            buf: list[np.ndarray] = []
            for *_, patch in entries:
                if patch is None:
                    continue
                buf.append(patch)
                if len(buf) >= batch:
                    yield self._checked(feature_fn(buf), len(buf), feature_name)
                    buf.clear()
            if buf:
                yield self._checked(feature_fn(buf), len(buf), feature_name)
                buf.clear()

        return self._write_feature_blocks(output_path, feature_name, blocks(), feature_attrs, batch,
                                          expected_total)

    def append_feature_matrix(self, *, output_path, feature_name: str, features: np.ndarray,
                              feature_attrs: Mapping[str, int | str], feature_batch: int,
                              expected_total: int | None = None) -> int:
        """Bulk form: the whole float32 [N, D] matrix (e.g. produced by the device pipeline)."""
        feats = self._checked(features, len(features), feature_name)
        step = max(1, int(feature_batch)) * 64
        gen = (feats[s:s + step] for s in range(0, feats.shape[0], step))
        return self._write_feature_blocks(output_path, feature_name, gen, feature_attrs,
                                          max(1, int(feature_batch)), expected_total)

    @staticmethod
    def _checked(feats, rows: int, feature_name: str) -> np.ndarray:
        arr = np.asarray(feats, dtype=np.float32)
        if arr.ndim != 2:
            raise ValueError(f"Feature extractor '{feature_name}' must return a 2D array, got shape {arr.shape}")
        if arr.shape[0] != rows:
            raise ValueError(f"Feature extractor '{feature_name}' returned {arr.shape[0]} rows for batch of size {rows}.")
        return arr

    def _write_feature_blocks(self, output_path, feature_name, blocks, feature_attrs, batch, expected_total) -> int:
        tmp_name = f"__tmp_{feature_name}"
        written = 0
        with h5.File(output_path, "a") as f:
            grp = f.require_group("features")
            if feature_name in grp:
                raise ValueError(f"Feature dataset '{feature_name}' already exists in {output_path}.")
            if tmp_name in grp:
                del grp[tmp_name]
            dataset = None
            moved = False
            try:
                for arr in blocks:
                    if dataset is None:
                        dim = int(arr.shape[1])
                        dataset = grp.create_dataset(tmp_name, shape=(0, dim), maxshape=(None, dim),
                                                     chunks=(batch, dim), dtype=np.float32)
                    elif dataset.shape[1] != arr.shape[1]:
                        raise ValueError(f"Feature dim mismatch for '{feature_name}': existing "
                                         f"{dataset.shape[1]}, new {arr.shape[1]}")
                    end = written + arr.shape[0]
                    dataset.resize((end, dataset.shape[1]))
                    dataset[written:end, :] = arr
                    written = end
                if dataset is None:
                    dim = int(feature_attrs.get("embedding_dim", 0))
                    if dim <= 0:
                        raise ValueError(f"Feature extractor '{feature_name}' missing valid embedding_dim "
                                         "to create dataset.")
                    dataset = grp.create_dataset(tmp_name, shape=(0, dim), maxshape=(None, dim),
                                                 chunks=(batch, dim), dtype=np.float32)
                if expected_total is not None and written != int(expected_total):
                    raise ValueError(f"Feature rows written ({written}) do not match expected coords "
                                     f"({expected_total})")
                grp.move(tmp_name, feature_name)
                moved = True
            except Exception:
                if tmp_name in grp:
                    del grp[tmp_name]
                elif moved and feature_name in grp:
                    del grp[feature_name]
                raise
        return int(written)


def read_coords(h5_path) -> np.ndarray:
    with h5.File(h5_path, "r") as f:
        return np.asarray(f["coords"][:], dtype=np.int32)
