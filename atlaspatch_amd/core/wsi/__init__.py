from .image_wsi import ImageWSI
from .iwsi import IWSI
from .openslide_wsi import OpenSlideWSI
from .synth_wsi import SynthWSI
from .wsi_factory import WSIFactory

__all__ = ["IWSI", "ImageWSI", "OpenSlideWSI", "SynthWSI", "WSIFactory"]
