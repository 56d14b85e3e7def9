"""HDF5 backend selection + the atomic append writer (reference: utils/h5.py:11-97).

``h5`` is h5py when it is importable (production installs), otherwise the ctypes binding over
libhdf5 in ``h5lite`` -- both produce the same on-disk layout.  ``H5AppendWriter`` writes to
``.<name>.tmp.<uuid>`` next to the target and ``os.replace``s it on close, so a half-written
coords file never appears under the final name.
"""
from __future__ import annotations

import json
import os
import uuid
from typing import Any, Mapping, Optional

import numpy as np

try:  # pragma: no cover - depends on the environment
    import h5py as h5
    BACKEND = "h5py"
except Exception:  # noqa: BLE001
    from . import h5lite as h5
    BACKEND = "h5lite"


def _attr_value(value: Any):
    if isinstance(value, dict):
        return json.dumps(value)
    return "None" if value is None else value


class H5AppendWriter:
    def __init__(self, path: str, chunk_rows: int = 8192) -> None:
        self.path = path
        self.chunk_rows = chunk_rows
        self._target = os.path.abspath(path)
        folder = os.path.dirname(self._target) or "."
        self._tmp: Optional[str] = os.path.join(folder, f".{os.path.basename(self._target)}.tmp.{uuid.uuid4().hex}")
        self._f = h5.File(self._tmp, "w")
        self._known: set[str] = set()
        self._closed = False

    def _dataset(self, key: str, sample: np.ndarray, attrs: Optional[Mapping[str, Any]]):
        if key not in self._known and key not in self._f:
            tail = tuple(sample.shape[1:])
            ds = self._f.create_dataset(key, shape=(0,) + tail, maxshape=(None,) + tail,
                                        chunks=(max(1, int(self.chunk_rows)),) + tail, dtype=sample.dtype)
            for name, value in (attrs or {}).items():
                ds.attrs[name] = _attr_value(value)
            self._known.add(key)
        return self._f[key]

    def append(self, assets: Mapping[str, np.ndarray],
               attributes: Optional[Mapping[str, Mapping[str, Any]]] = None) -> None:
        for key, block in assets.items():
            ds = self._dataset(key, block, attributes.get(key) if attributes else None)
            rows = int(block.shape[0])
            if rows == 0:
                continue
            start = int(ds.shape[0])
            ds.resize(start + rows, axis=0)
            ds[start:start + rows] = block

    def update_file_attrs(self, file_attrs: Mapping[str, Any]) -> None:
        for name, value in file_attrs.items():
            self._f.attrs[name] = _attr_value(value)

    def close(self) -> None:
        if self._closed:
            return
        try:
            self._f.close()
        finally:
            if self._tmp is not None:
                os.replace(self._tmp, self._target)
                self._tmp = None
            self._closed = True

    def abort(self) -> None:
        if self._closed:
            return
        try:
            self._f.close()
        finally:
            if self._tmp and os.path.exists(self._tmp):
                try:
                    os.remove(self._tmp)
                except OSError:
                    pass
            self._closed = True
