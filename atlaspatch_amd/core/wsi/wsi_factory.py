"""Backend registry keyed by file extension (reference: core/wsi/wsi_factory.py:12-141), with the
synthetic backend mapped to ``.synth``."""
from __future__ import annotations

import os
from pathlib import Path
from typing import Optional

from .image_wsi import ImageWSI
from .iwsi import IWSI
from .openslide_wsi import OpenSlideWSI
from .synth_wsi import SynthWSI

_OPENSLIDE_EXT = (".svs", ".tif", ".tiff", ".ndpi", ".vms", ".vmu", ".scn", ".mrxs", ".bif", ".biff",
                  ".dcm", ".dicom")
_IMAGE_EXT = (".png", ".jpg", ".jpeg", ".bmp", ".webp", ".gif")


class WSIFactory:
    _registry = {"openslide": OpenSlideWSI, "image": ImageWSI, "synth": SynthWSI}
    _formats = {**{e: "openslide" for e in _OPENSLIDE_EXT}, **{e: "image" for e in _IMAGE_EXT},
                ".synth": "synth"}

    @classmethod
    def register(cls, name: str, impl_class) -> None:
        cls._registry[name] = impl_class

    @classmethod
    def map_extension(cls, ext: str, backend: str) -> None:
        if backend not in cls._registry:
            raise ValueError(f"Unknown backend: {backend}")
        cls._formats[(ext if ext.startswith(".") else "." + ext).lower()] = backend

    @classmethod
    def detect(cls, path: str) -> Optional[str]:
        return cls._formats.get(Path(path).suffix.lower())

    @classmethod
    def load(cls, path: str, backend: Optional[str] = None, mpp: Optional[float] = None, **kwargs) -> IWSI:
        if not os.path.exists(path):
            raise FileNotFoundError(f"File not found: {path}")
        if backend is None:
            backend = cls.detect(path)
            if backend is None:
                raise ValueError(f"No backend found for: {path}")
        elif backend not in cls._registry:
            raise ValueError(f"Unknown backend: {backend}")
        return cls._registry[backend](path=path, mpp=mpp, **kwargs)

    @classmethod
    def try_load(cls, path: str, backends: Optional[list] = None, mpp: Optional[float] = None, **kwargs) -> IWSI:
        if not os.path.exists(path):
            raise FileNotFoundError(f"File not found: {path}")
        problems = []
        for name in (backends if backends is not None else list(cls._registry)):
            if name not in cls._registry:
                problems.append(f"{name}: not registered")
                continue
            try:
                return cls.load(path, backend=name, mpp=mpp, **kwargs)
            except Exception as exc:  # noqa: BLE001
                problems.append(f"{name}: {exc}")
        raise RuntimeError(f"All backends failed for {path}:\n" + "\n".join(problems))
