"""A minimal h5py-compatible layer over the HDF5 C library (ctypes), used when h5py itself is
not importable (this image ships libhdf5 but no h5py wheel for the system interpreter).

Only the subset the H5 output contract needs (reference: services/storage.py, utils/h5.py,
utils/features.py): files, groups (``require_group``, ``in``, ``del``, ``move``, ``items``),
chunked resizable datasets of int32 / float32 / fixed-length byte strings with row-slice
reads and writes, and scalar int / float / str attributes stored the way h5py stores them
(int64, float64, variable-length UTF-8).  Files are ordinary HDF5 and open in h5py / h5dump.
"""
from __future__ import annotations

import ctypes as C
import threading
import ctypes.util
import glob
import os
from typing import Any, Optional

import numpy as np

hid_t = C.c_int64
hsize_t = C.c_uint64
herr_t = C.c_int
htri_t = C.c_int

H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC = 0x0, 0x1, 0x2
H5P_DEFAULT = 0
H5S_ALL = 0
H5S_SELECT_SET = 0
H5S_UNLIMITED = 0xFFFFFFFFFFFFFFFF
H5T_VARIABLE = C.c_size_t(-1).value
H5T_CSET_UTF8 = 1
H5T_STR_NULLPAD = 1
H5T_STR_NULLTERM = 0
H5_INDEX_NAME = 0
H5_ITER_INC = 0
H5S_SCALAR = 0
H5T_INTEGER, H5T_FLOAT, H5T_STRING = 0, 1, 3

_lib = None


class H5LiteError(RuntimeError):
    pass


def _candidates():
    env = os.environ.get("ATLASPATCH_HDF5_LIB")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5")
    if found:
        yield found
    for pattern in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5*.so*",
                    "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*", "/usr/local/lib/libhdf5.so*"):
        for path in sorted(glob.glob(pattern)):
            if "_hl" not in path and "_cpp" not in path and "fortran" not in path:
                yield path


def available() -> bool:
    try:
        _load()
        return True
    except H5LiteError:
        return False


class _Serialised:
    """libhdf5 is not thread-safe unless built with --enable-threadsafe (the system's is not; h5py guards it with one
    global lock for the same reason).  The coordinate path writes H5 files from worker threads while the main thread reads
    others, so every call into the library goes through ONE re-entrant lock; ctypes would otherwise release the interpreter
    lock and let two threads into the library at once."""

    def __init__(self, lib) -> None:
        self._raw = lib
        self._cache: dict = {}

    def __getattr__(self, name):
        fn = self._cache.get(name)
        if fn is None:
            raw = getattr(self._raw, name)

            def call(*args, _raw=raw):
                with _H5_LOCK:
                    return _raw(*args)

            fn = self._cache[name] = call
        return fn


_H5_LOCK = threading.RLock()


def _load():
    global _lib
    if _lib is not None:
        return _lib
    last = None
    for path in _candidates():
        try:
            lib = C.CDLL(path)
            lib.H5open.restype = herr_t
            if lib.H5open() < 0:
                continue
            _declare(lib)
            _lib = _Serialised(lib)
            return _lib
        except OSError as exc:          # noqa: PERF203
            last = exc
    raise H5LiteError("No HDF5 backend: h5py is not importable and libhdf5 was not found "
                      f"(set ATLASPATCH_HDF5_LIB). Last error: {last}")


def _declare(lib):
    sig = {
        "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]),
        "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
        "H5Fclose": (herr_t, [hid_t]), "H5Fflush": (herr_t, [hid_t, C.c_int]),
        "H5Gcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]),
        "H5Gopen2": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Gclose": (herr_t, [hid_t]),
        "H5Lexists": (htri_t, [hid_t, C.c_char_p, hid_t]),
        "H5Ldelete": (herr_t, [hid_t, C.c_char_p, hid_t]),
        "H5Lmove": (herr_t, [hid_t, C.c_char_p, hid_t, C.c_char_p, hid_t, hid_t]),
        "H5Lget_name_by_idx": (C.c_ssize_t, [hid_t, C.c_char_p, C.c_int, C.c_int, hsize_t, C.c_char_p,
                                              C.c_size_t, hid_t]),
        "H5Gget_num_objs": (herr_t, [hid_t, C.POINTER(hsize_t)]),
        "H5Oopen": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Oclose": (herr_t, [hid_t]),
        "H5Iget_type": (C.c_int, [hid_t]),
        "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Screate": (hid_t, [C.c_int]), "H5Sclose": (herr_t, [hid_t]),
        "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
        "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Sselect_hyperslab": (herr_t, [hid_t, C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t),
                                         C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Pcreate": (hid_t, [hid_t]), "H5Pclose": (herr_t, [hid_t]),
        "H5Pset_chunk": (herr_t, [hid_t, C.c_int, C.POINTER(hsize_t)]),
        "H5Pget_chunk": (C.c_int, [hid_t, C.c_int, C.POINTER(hsize_t)]),
        "H5Pget_layout": (C.c_int, [hid_t]),
        "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Dclose": (herr_t, [hid_t]),
        "H5Dset_extent": (herr_t, [hid_t, C.POINTER(hsize_t)]),
        "H5Dget_space": (hid_t, [hid_t]), "H5Dget_type": (hid_t, [hid_t]),
        "H5Dget_create_plist": (hid_t, [hid_t]),
        "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Tcopy": (hid_t, [hid_t]), "H5Tclose": (herr_t, [hid_t]),
        "H5Tset_size": (herr_t, [hid_t, C.c_size_t]), "H5Tget_size": (C.c_size_t, [hid_t]),
        "H5Tset_cset": (herr_t, [hid_t, C.c_int]), "H5Tset_strpad": (herr_t, [hid_t, C.c_int]),
        "H5Tget_class": (C.c_int, [hid_t]), "H5Tis_variable_str": (htri_t, [hid_t]),
        "H5Acreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]),
        "H5Aopen": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Aclose": (herr_t, [hid_t]),
        "H5Awrite": (herr_t, [hid_t, hid_t, C.c_void_p]), "H5Aread": (herr_t, [hid_t, hid_t, C.c_void_p]),
        "H5Aexists": (htri_t, [hid_t, C.c_char_p]), "H5Adelete": (herr_t, [hid_t, C.c_char_p]),
        "H5Aget_type": (hid_t, [hid_t]), "H5Aget_num_attrs": (C.c_int, [hid_t]),
        "H5Aopen_idx": (hid_t, [hid_t, C.c_uint]),
        "H5Aget_name": (C.c_ssize_t, [hid_t, C.c_size_t, C.c_char_p]),
        "H5Dvlen_reclaim": (herr_t, [hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Eset_auto2": (herr_t, [hid_t, C.c_void_p, C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    lib.H5Eset_auto2(0, None, None)        # errors are reported through return codes -> exceptions


def _g(name: str) -> int:
    return hid_t.in_dll(_load()._raw, name).value


def _ok(code, what):
    if code < 0:
        raise H5LiteError(f"HDF5 call failed: {what}")
    return code


def _np_to_h5(dtype: np.dtype):
    """-> (type id, must_close)"""
    lib = _load()
    dtype = np.dtype(dtype)
    if dtype == np.int32:
        return _g("H5T_NATIVE_INT32_g"), False
    if dtype == np.int64:
        return _g("H5T_NATIVE_INT64_g"), False
    if dtype == np.float32:
        return _g("H5T_NATIVE_FLOAT_g"), False
    if dtype == np.float64:
        return _g("H5T_NATIVE_DOUBLE_g"), False
    if dtype == np.uint8:
        return _g("H5T_NATIVE_UINT8_g"), False
    if dtype.kind == "S":
        t = _ok(lib.H5Tcopy(_g("H5T_C_S1_g")), "H5Tcopy")
        lib.H5Tset_size(t, dtype.itemsize)
        lib.H5Tset_strpad(t, H5T_STR_NULLPAD)
        return t, True
    raise H5LiteError(f"unsupported dtype {dtype}")


def _h5_to_np(tid) -> np.dtype:
    lib = _load()
    cls, size = lib.H5Tget_class(tid), lib.H5Tget_size(tid)
    if cls == H5T_INTEGER:
        return np.dtype({1: np.uint8, 4: np.int32, 8: np.int64}[size])
    if cls == H5T_FLOAT:
        return np.dtype({4: np.float32, 8: np.float64}[size])
    if cls == H5T_STRING:
        return np.dtype(f"S{size}")
    raise H5LiteError(f"unsupported HDF5 type class {cls}")


class AttributeManager:
    def __init__(self, owner_id: int) -> None:
        self._oid = owner_id

    def __setitem__(self, name: str, value: Any) -> None:
        lib = _load()
        key = name.encode()
        if lib.H5Aexists(self._oid, key) > 0:
            _ok(lib.H5Adelete(self._oid, key), "H5Adelete")
        space = _ok(lib.H5Screate(H5S_SCALAR), "H5Screate")
        try:
            if isinstance(value, (bool, np.bool_)):
                value = int(value)
            if isinstance(value, (int, np.integer)):
                tid, buf, close = _g("H5T_NATIVE_INT64_g"), C.c_int64(int(value)), False
            elif isinstance(value, (float, np.floating)):
                tid, buf, close = _g("H5T_NATIVE_DOUBLE_g"), C.c_double(float(value)), False
            elif isinstance(value, (str, bytes)):
                raw = value.encode("utf-8") if isinstance(value, str) else value
                tid = _ok(lib.H5Tcopy(_g("H5T_C_S1_g")), "H5Tcopy")
                lib.H5Tset_size(tid, H5T_VARIABLE)
                if isinstance(value, str):
                    lib.H5Tset_cset(tid, H5T_CSET_UTF8)
                self._keep = C.c_char_p(raw)
                buf, close = C.pointer(self._keep), True
            else:
                raise H5LiteError(f"unsupported attribute value {type(value)}")
            aid = _ok(lib.H5Acreate2(self._oid, key, tid, space, H5P_DEFAULT, H5P_DEFAULT), "H5Acreate2")
            try:
                _ok(lib.H5Awrite(aid, tid, C.cast(buf if close else C.byref(buf), C.c_void_p)), "H5Awrite")
            finally:
                lib.H5Aclose(aid)
                if close:
                    lib.H5Tclose(tid)
        finally:
            lib.H5Sclose(space)

    def __contains__(self, name: str) -> bool:
        return _load().H5Aexists(self._oid, name.encode()) > 0

    def __getitem__(self, name: str):
        lib = _load()
        aid = lib.H5Aopen(self._oid, name.encode(), H5P_DEFAULT)
        if aid < 0:
            raise KeyError(name)
        try:
            tid = _ok(lib.H5Aget_type(aid), "H5Aget_type")
            try:
                cls = lib.H5Tget_class(tid)
                if cls == H5T_INTEGER:
                    v = C.c_int64()
                    _ok(lib.H5Aread(aid, _g("H5T_NATIVE_INT64_g"), C.byref(v)), "H5Aread")
                    return int(v.value)
                if cls == H5T_FLOAT:
                    d = C.c_double()
                    _ok(lib.H5Aread(aid, _g("H5T_NATIVE_DOUBLE_g"), C.byref(d)), "H5Aread")
                    return float(d.value)
                if cls == H5T_STRING:
                    if lib.H5Tis_variable_str(tid) > 0:
                        p = C.c_char_p()
                        _ok(lib.H5Aread(aid, tid, C.byref(p)), "H5Aread")
                        return (p.value or b"").decode("utf-8", errors="replace")
                    size = lib.H5Tget_size(tid)
                    buf = C.create_string_buffer(size + 1)
                    _ok(lib.H5Aread(aid, tid, buf), "H5Aread")
                    return buf.value.decode("utf-8", errors="replace")
                raise H5LiteError(f"unsupported attribute class {cls}")
            finally:
                lib.H5Tclose(tid)
        finally:
            lib.H5Aclose(aid)

    def get(self, name, default=None):
        return self[name] if name in self else default

    def keys(self):
        lib = _load()
        out = []
        for i in range(max(0, lib.H5Aget_num_attrs(self._oid))):
            aid = lib.H5Aopen_idx(self._oid, i)
            if aid < 0:
                continue
            n = lib.H5Aget_name(aid, 0, None)
            buf = C.create_string_buffer(n + 1)
            lib.H5Aget_name(aid, n + 1, buf)
            out.append(buf.value.decode())
            lib.H5Aclose(aid)
        return out

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def update(self, mapping):
        for k, v in mapping.items():
            self[k] = v


class Dataset:
    def __init__(self, did: int, name: str) -> None:
        self._id = did
        self.name = name
        self.attrs = AttributeManager(did)

    def _dims(self):
        lib = _load()
        sp = _ok(lib.H5Dget_space(self._id), "H5Dget_space")
        try:
            nd = lib.H5Sget_simple_extent_ndims(sp)
            dims = (hsize_t * nd)()
            maxd = (hsize_t * nd)()
            lib.H5Sget_simple_extent_dims(sp, dims, maxd)
            return tuple(int(d) for d in dims), tuple(None if m == H5S_UNLIMITED else int(m) for m in maxd)
        finally:
            lib.H5Sclose(sp)

    @property
    def shape(self):
        return self._dims()[0]

    @property
    def maxshape(self):
        return self._dims()[1]

    @property
    def dtype(self) -> np.dtype:
        lib = _load()
        tid = _ok(lib.H5Dget_type(self._id), "H5Dget_type")
        try:
            return _h5_to_np(tid)
        finally:
            lib.H5Tclose(tid)

    @property
    def chunks(self):
        lib = _load()
        pl = _ok(lib.H5Dget_create_plist(self._id), "H5Dget_create_plist")
        try:
            if lib.H5Pget_layout(pl) != 2:      # H5D_CHUNKED
                return None
            nd = len(self.shape)
            dims = (hsize_t * nd)()
            lib.H5Pget_chunk(pl, nd, dims)
            return tuple(int(d) for d in dims)
        finally:
            lib.H5Pclose(pl)

    def __len__(self):
        return self.shape[0]

    def resize(self, size, axis: Optional[int] = None) -> None:
        shape = list(self.shape)
        if axis is not None:
            shape[axis] = int(size)
        else:
            shape = [int(s) for s in size]
        dims = (hsize_t * len(shape))(*shape)
        _ok(_load().H5Dset_extent(self._id, dims), "H5Dset_extent")

    def _row_range(self, key):
        shape = self.shape
        if isinstance(key, tuple):
            rows = key[0]
            rest = key[1:]
            if any(not (isinstance(r, slice) and r == slice(None)) for r in rest):
                raise H5LiteError("only full trailing-dimension slices are supported")
        else:
            rows = key
        if rows is Ellipsis:
            return 0, shape[0], False
        if isinstance(rows, (int, np.integer)):
            i = int(rows)
            if i < 0:
                i += shape[0]
            return i, i + 1, True
        start, stop, step = rows.indices(shape[0])
        if step != 1:
            raise H5LiteError("strided row slices are not supported")
        return start, max(start, stop), False

    def _io(self, start, stop, arr: np.ndarray, write: bool) -> None:
        lib = _load()
        shape = self.shape
        n = stop - start
        if n <= 0:
            return
        nd = len(shape)
        fsp = _ok(lib.H5Dget_space(self._id), "H5Dget_space")
        offs = (hsize_t * nd)(start, *([0] * (nd - 1)))
        cnt = (hsize_t * nd)(n, *shape[1:])
        _ok(lib.H5Sselect_hyperslab(fsp, H5S_SELECT_SET, offs, None, cnt, None), "H5Sselect_hyperslab")
        msp = _ok(lib.H5Screate_simple(nd, cnt, None), "H5Screate_simple")
        tid, close = _np_to_h5(arr.dtype)
        try:
            fn = lib.H5Dwrite if write else lib.H5Dread
            _ok(fn(self._id, tid, msp, fsp, H5P_DEFAULT, arr.ctypes.data_as(C.c_void_p)),
                "H5Dwrite" if write else "H5Dread")
        finally:
            if close:
                lib.H5Tclose(tid)
            lib.H5Sclose(msp)
            lib.H5Sclose(fsp)

    def __setitem__(self, key, value) -> None:
        start, stop, _ = self._row_range(key)
        arr = np.ascontiguousarray(value, dtype=self.dtype)
        want = (stop - start,) + self.shape[1:]
        if arr.shape != want:
            arr = np.ascontiguousarray(np.broadcast_to(arr, want))
        self._io(start, stop, arr, True)

    def __getitem__(self, key):
        start, stop, scalar = self._row_range(key)
        arr = np.empty((stop - start,) + self.shape[1:], dtype=self.dtype)
        self._io(start, stop, arr, False)
        return arr[0] if scalar else arr

    def _close(self):
        if self._id is not None:
            _load().H5Dclose(self._id)
            self._id = None


class Group:
    def __init__(self, gid: int, name: str = "/") -> None:
        self._id = gid
        self.name = name
        self.attrs = AttributeManager(gid)
        self._open: list = []

    def __contains__(self, name: str) -> bool:
        lib = _load()
        parts = [p for p in name.split("/") if p]
        path = ""
        for p in parts:
            path = f"{path}/{p}" if path else p
            if lib.H5Lexists(self._id, path.encode(), H5P_DEFAULT) <= 0:
                return False
        return True

    def keys(self):
        lib = _load()
        n = hsize_t()
        _ok(lib.H5Gget_num_objs(self._id, C.byref(n)), "H5Gget_num_objs")
        out = []
        for i in range(n.value):
            ln = lib.H5Lget_name_by_idx(self._id, b".", H5_INDEX_NAME, H5_ITER_INC, i, None, 0, H5P_DEFAULT)
            buf = C.create_string_buffer(ln + 1)
            lib.H5Lget_name_by_idx(self._id, b".", H5_INDEX_NAME, H5_ITER_INC, i, buf, ln + 1, H5P_DEFAULT)
            out.append(buf.value.decode())
        return out

    def __iter__(self):
        return iter(self.keys())

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def __getitem__(self, name: str):
        lib = _load()
        if name not in self:
            raise KeyError(name)
        oid = _ok(lib.H5Oopen(self._id, name.encode(), H5P_DEFAULT), "H5Oopen")
        kind = lib.H5Iget_type(oid)
        lib.H5Oclose(oid)
        if kind == 2:        # H5I_GROUP
            g = Group(_ok(lib.H5Gopen2(self._id, name.encode(), H5P_DEFAULT), "H5Gopen2"), name)
            self._open.append(g)
            return g
        d = Dataset(_ok(lib.H5Dopen2(self._id, name.encode(), H5P_DEFAULT), "H5Dopen2"), name)
        self._open.append(d)
        return d

    def __delitem__(self, name: str) -> None:
        _ok(_load().H5Ldelete(self._id, name.encode(), H5P_DEFAULT), "H5Ldelete")

    def move(self, src: str, dst: str) -> None:
        if dst in self:
            raise ValueError(f"Unable to move link (destination '{dst}' exists)")
        _ok(_load().H5Lmove(self._id, src.encode(), self._id, dst.encode(), H5P_DEFAULT, H5P_DEFAULT), "H5Lmove")

    def require_group(self, name: str) -> "Group":
        lib = _load()
        if name in self:
            return self[name]
        g = Group(_ok(lib.H5Gcreate2(self._id, name.encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT),
                      "H5Gcreate2"), name)
        self._open.append(g)
        return g

    create_group = require_group

    def create_dataset(self, name: str, shape=None, maxshape=None, chunks=None, dtype=None, data=None) -> Dataset:
        lib = _load()
        if name in self:
            raise ValueError(f"Unable to create dataset (name '{name}' already exists)")
        if data is not None:
            data = np.asarray(data)
            shape = data.shape if shape is None else shape
            dtype = dtype or data.dtype
        shape = tuple(int(s) for s in shape)
        nd = len(shape)
        dims = (hsize_t * nd)(*shape)
        maxd = None
        if maxshape is not None:
            maxd = (hsize_t * nd)(*[H5S_UNLIMITED if m is None else int(m) for m in maxshape])
            if chunks is None:
                chunks = tuple(max(1, s) for s in shape)
        space = _ok(lib.H5Screate_simple(nd, dims, maxd), "H5Screate_simple")
        plist = _ok(lib.H5Pcreate(_g("H5P_CLS_DATASET_CREATE_ID_g")), "H5Pcreate")
        tid, close = _np_to_h5(np.dtype(dtype))
        try:
            if chunks is not None:
                cd = (hsize_t * nd)(*[max(1, int(c)) for c in chunks])
                _ok(lib.H5Pset_chunk(plist, nd, cd), "H5Pset_chunk")
            did = _ok(lib.H5Dcreate2(self._id, name.encode(), tid, space, H5P_DEFAULT, plist, H5P_DEFAULT),
                      "H5Dcreate2")
        finally:
            if close:
                lib.H5Tclose(tid)
            lib.H5Pclose(plist)
            lib.H5Sclose(space)
        ds = Dataset(did, name)
        self._open.append(ds)
        if data is not None and data.size:
            ds[0:shape[0]] = data
        return ds

    def _close_children(self):
        for obj in self._open:
            if isinstance(obj, Group):
                obj._close_children()
                if obj._id is not None:
                    _load().H5Gclose(obj._id)
                    obj._id = None
            else:
                obj._close()
        self._open = []


class File(Group):
    def __init__(self, path, mode: str = "r") -> None:
        lib = _load()
        self.filename = os.fspath(path)
        raw = self.filename.encode()
        if mode == "w":
            fid = lib.H5Fcreate(raw, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        elif mode == "r":
            if not os.path.exists(self.filename):
                raise FileNotFoundError(self.filename)
            fid = lib.H5Fopen(raw, H5F_ACC_RDONLY, H5P_DEFAULT)
        elif mode in ("a", "r+"):
            if os.path.exists(self.filename):
                fid = lib.H5Fopen(raw, H5F_ACC_RDWR, H5P_DEFAULT)
            elif mode == "a":
                fid = lib.H5Fcreate(raw, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
            else:
                raise FileNotFoundError(self.filename)
        else:
            raise ValueError(f"unsupported mode {mode}")
        if fid < 0:
            raise OSError(f"Unable to open HDF5 file {self.filename} (mode {mode})")
        self._fid = fid
        gid = _ok(lib.H5Gopen2(fid, b"/", H5P_DEFAULT), "H5Gopen2(/)")
        super().__init__(gid, "/")

    def close(self) -> None:
        if self._fid is None:
            return
        lib = _load()
        self._close_children()
        lib.H5Gclose(self._id)
        self._id = None
        lib.H5Fclose(self._fid)
        self._fid = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
