"""Import harness for the READ-ONLY reference at /root/reference (generation time only).

Used ONLY by tests/golden/gen_golden.py in the build container.  Nothing here
runs on the GPU box (the reference does not travel).  The reference is pure
Python but its package ``__init__`` files import cv2 / h5py / torchvision /
sam2, none of which exist in this image.  The harness therefore

  1. pre-seeds ``sys.modules`` with empty stub *packages* whose ``__path__``
     points into /root/reference/atlas_patch, so sub-modules import from the
     reference's own files while the ``__init__`` files are skipped;
  2. installs a shim ``cv2`` exposing exactly the five primitives the hot path
     uses, implemented by ``oracle/cv2_restated.py``;
  3. installs a small fake ``h5py`` (pickle-backed File/Group/Dataset) that
     records names, dtypes, shapes, chunks, maxshape and attrs.

With that, the reference's ``PatchExtractionService.extract``,
``H5PatchWriter.write_coords/append_features``,
``PatchFeatureEmbeddingService.embed_all`` and
``PatchFeatureExtractor.extract_batch`` run UNMODIFIED.
"""
from __future__ import annotations

import os
import pickle
import sys
import types

import numpy as np

REF_ROOT = "/root/reference"
REF_PKG = os.path.join(REF_ROOT, "atlas_patch")


# ----------------------------------------------------------------------------- fake h5py
class _Attrs(dict):
    pass


class FakeDataset:
    def __init__(self, name, shape, maxshape, chunks, dtype):
        self.name = name
        self.dtype = np.dtype(dtype)
        self.maxshape = tuple(maxshape) if maxshape is not None else tuple(shape)
        self.chunks = tuple(chunks) if chunks is not None else None
        self._data = np.zeros(tuple(shape), dtype=self.dtype)
        self.attrs = _Attrs()

    @property
    def shape(self):
        return self._data.shape

    def resize(self, size, axis=None):
        if axis is not None:
            new_shape = list(self._data.shape)
            new_shape[axis] = int(size)
        else:
            new_shape = [int(s) for s in size]
        new = np.zeros(tuple(new_shape), dtype=self.dtype)
        sl = tuple(slice(0, min(a, b)) for a, b in zip(new_shape, self._data.shape))
        new[sl] = self._data[sl]
        self._data = new

    def __getitem__(self, key):
        return self._data[key]

    def __setitem__(self, key, value):
        self._data[key] = value

    def __len__(self):
        return self._data.shape[0]


class FakeGroup:
    def __init__(self, name="/"):
        self.name = name
        self._items = {}
        self.attrs = _Attrs()

    def create_dataset(self, name, shape=None, maxshape=None, chunks=None, dtype=None, data=None):
        if name in self._items:
            raise ValueError(f"dataset {name} exists")
        if data is not None:
            data = np.asarray(data)
            shape = data.shape
            dtype = dtype or data.dtype
        ds = FakeDataset(name, shape, maxshape, chunks, dtype)
        if data is not None:
            ds._data[...] = data
        self._items[name] = ds
        return ds

    def require_group(self, name):
        if name not in self._items:
            self._items[name] = FakeGroup(name)
        return self._items[name]

    def create_group(self, name):
        return self.require_group(name)

    def move(self, src, dst):
        if dst in self._items:
            raise ValueError(f"{dst} exists")
        self._items[dst] = self._items.pop(src)
        self._items[dst].name = dst

    def items(self):
        return self._items.items()

    def keys(self):
        return self._items.keys()

    def __contains__(self, name):
        return name in self._items

    def __getitem__(self, name):
        return self._items[name]

    def __delitem__(self, name):
        del self._items[name]


class FakeFile(FakeGroup):
    def __init__(self, path, mode="r"):
        super().__init__("/")
        self._path = os.fspath(path)
        self._mode = mode
        if mode in ("r", "a", "r+") and os.path.exists(self._path):
            with open(self._path, "rb") as fh:
                state = pickle.load(fh)
            self._items = state["items"]
            self.attrs = state["attrs"]
        elif mode == "r":
            raise FileNotFoundError(self._path)

    def close(self):
        if self._mode != "r":
            with open(self._path, "wb") as fh:
                pickle.dump({"items": self._items, "attrs": self.attrs}, fh)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def describe_h5(path):
    """Structural dump of a fake file: names, dtypes, shapes, chunks, maxshape, attrs."""
    f = FakeFile(path, "r")

    def walk(g, prefix):
        out = {}
        for name, obj in g.items():
            full = f"{prefix}/{name}" if prefix else name
            if isinstance(obj, FakeGroup):
                out.update(walk(obj, full))
            else:
                out[full] = {
                    "dtype": obj.dtype.str,
                    "shape": list(obj.shape),
                    "chunks": list(obj.chunks) if obj.chunks else None,
                    "maxshape": [None if m is None else int(m) for m in obj.maxshape],
                    "attrs": {k: (v if isinstance(v, (int, float, str)) else str(v))
                              for k, v in obj.attrs.items()},
                }
        return out

    return {
        "datasets": walk(f, ""),
        "file_attrs": {k: (type(v).__name__) for k, v in f.attrs.items()},
        "file_attr_values": {k: (v if isinstance(v, (int, float, str)) else str(v))
                             for k, v in f.attrs.items()},
    }


def read_h5(path):
    return FakeFile(path, "r")


# ----------------------------------------------------------------------------- install
def install():
    """Install stubs + shims and return the imported reference modules as a namespace."""
    repo_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if repo_root not in sys.path:
        sys.path.insert(0, repo_root)
    if REF_ROOT not in sys.path:
        sys.path.append(REF_ROOT)

    from oracle import cv2_restated

    cv2 = types.ModuleType("cv2")
    for name in ("findContours", "contourArea", "pointPolygonTest", "boundingRect",
                 "RETR_CCOMP", "RETR_EXTERNAL", "RETR_LIST", "RETR_TREE",
                 "CHAIN_APPROX_NONE", "CHAIN_APPROX_SIMPLE"):
        setattr(cv2, name, getattr(cv2_restated, name))
    cv2.INTER_AREA, cv2.INTER_CUBIC, cv2.INTER_LINEAR = 3, 2, 1

    def _no_resize(*a, **k):
        raise NotImplementedError("cv2.resize is outside the golden harness")

    cv2.resize = _no_resize
    sys.modules["cv2"] = cv2

    h5py = types.ModuleType("h5py")
    h5py.File = FakeFile
    h5py.Group = FakeGroup
    h5py.Dataset = FakeDataset
    sys.modules["h5py"] = h5py

    for pkg in ("atlas_patch", "atlas_patch.core", "atlas_patch.core.wsi", "atlas_patch.models",
                "atlas_patch.models.patch", "atlas_patch.services", "atlas_patch.utils",
                "atlas_patch.orchestration"):
        mod = types.ModuleType(pkg)
        mod.__path__ = [os.path.join(REF_PKG, *pkg.split(".")[1:])]
        sys.modules[pkg] = mod

    import importlib

    ns = types.SimpleNamespace()
    ns.config = importlib.import_module("atlas_patch.core.config")
    ns.models = importlib.import_module("atlas_patch.core.models")
    ns.paths = importlib.import_module("atlas_patch.core.paths")
    ns.iwsi = importlib.import_module("atlas_patch.core.wsi.iwsi")
    ns.contours = importlib.import_module("atlas_patch.utils.contours")
    ns.features = importlib.import_module("atlas_patch.utils.features")
    ns.h5 = importlib.import_module("atlas_patch.utils.h5")
    ns.base = importlib.import_module("atlas_patch.models.patch.base")
    ns.registry = importlib.import_module("atlas_patch.models.patch.registry")
    ns.custom = importlib.import_module("atlas_patch.models.patch.custom")

    # names the service modules import from the (skipped) package __init__ files
    sys.modules["atlas_patch.utils"].get_existing_features = ns.features.get_existing_features
    sys.modules["atlas_patch.utils"].missing_features = ns.features.missing_features
    sys.modules["atlas_patch.utils"].parse_feature_list = ns.features.parse_feature_list

    def _build_default_registry(*, device="cuda", num_workers=0, dtype=None):
        return ns.registry.PatchFeatureExtractorRegistry()

    sys.modules["atlas_patch.models.patch"].build_default_registry = _build_default_registry

    ns.interfaces = importlib.import_module("atlas_patch.services.interfaces")
    ns.storage = importlib.import_module("atlas_patch.services.storage")
    ns.extraction = importlib.import_module("atlas_patch.services.extraction")
    ns.feature_embedding = importlib.import_module("atlas_patch.services.feature_embedding")
    return ns
