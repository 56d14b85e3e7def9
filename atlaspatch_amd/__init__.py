"""MI355X-native hot path of the AtlasPatch WSI patch-embedding pipeline (drop-in boundary:
CLI ``process`` / ``segment-and-get-coords`` + the encoder plugin API).  The compute lives in
``libatlaspatch_hip.so`` (hand-written gfx950 HIP kernels behind a C ABI, include/atlaspatch_hip.h)."""

__version__ = "0.1.0"
