"""ctypes binding of libatlaspatch_hip.so (the C ABI declared in include/atlaspatch_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C atlaspatch_amd/csrc``.
There is NO fallback: if the library is missing or a call fails, the product raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

AP_F32, AP_F16, AP_BF16 = 0, 1, 2
AP_OK = 0
PROF_KINDS = ("preproc", "gemm_patch_embed", "gemm_qkv", "gemm_proj", "gemm_fc1", "gemm_fc2",
              "attention", "layernorm", "cls_tail")
AP_ERR_CAPACITY = -6
AP_ERR_UNSUPPORTED = -4
AP_ERR_INVALID = -1

# Stream capture (the SAM2 hipGraph) must not overlap synchronous HIP calls on the legacy stream from other threads (the
# encoder built on a side thread uploads weights with hipMemcpy): both sides hold this lock for their critical section.
HIP_CAPTURE_LOCK = threading.RLock()

_LIB_NAME = "libatlaspatch_hip.so"
_lock = threading.Lock()
_lib = None


class HipLibraryError(RuntimeError):
    pass


ABI_VERSION = 20                # include/atlaspatch_hip.h: AP_ABI_VERSION; load() refuses a library that reports another


class VitConfig(C.Structure):
    """``ap_vit_config`` (ABI v20).  ``struct_size`` is filled in here, so positional construction starts at
    ``image_size`` as it did before the field existed."""
    _fields_ = [("struct_size", C.c_uint32), ("image_size", C.c_int), ("patch_size", C.c_int), ("dim", C.c_int),
                ("depth", C.c_int), ("heads", C.c_int), ("mlp_dim", C.c_int),
                ("ln_eps", C.c_float), ("layer_scale", C.c_int), ("compute_dtype", C.c_int),
                ("pool", C.c_int), ("pool_dim", C.c_int), ("pool_heads", C.c_int), ("pool_ln_eps", C.c_float),
                ("reg_tokens", C.c_int), ("no_embed_class", C.c_int), ("mlp_type", C.c_int), ("head_dim", C.c_int),
                ("attn_scale", C.c_float), ("pre_norm", C.c_int), ("act", C.c_int), ("proj_dim", C.c_int), ("rope", C.c_int)]

    def __init__(self, *args, **kw):
        super().__init__(C.sizeof(type(self)), *args, **kw)


# name -> (restype, argtypes); every symbol include/atlaspatch_hip.h declares
SIGNATURES = {
    "ap_abi_version": (C.c_int, []),
    "ap_last_error": (C.c_char_p, []),
    "ap_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    "ap_preproc_u8hwc_to_chw": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p,
                                          C.c_int, C.c_void_p]),
    "ap_preproc_u8hwc_to_patchrows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float),
                                                C.POINTER(C.c_float), C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ap_host_inflate_tiles": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.c_int, C.c_size_t]),
    "ap_host_decode_jpeg_tiles": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.c_int, C.c_int]),
    "ap_host_format_passports": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_void_p, C.c_int]),
    "ap_host_gather_tiles": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_size_t]),
    "ap_host_synth_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                      C.c_uint32, C.c_void_p, C.c_int]),
    "ap_resample_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "ap_cv2_resize_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p]),
    "ap_tile_content_counts": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                         C.c_void_p]),
    "ap_sizeof_vit_config": (C.c_size_t, []),
    "ap_vit_config_init": (C.c_int, [C.POINTER(VitConfig), C.c_size_t]),
    "ap_vit_create": (C.c_int, [C.POINTER(VitConfig), C.POINTER(C.c_void_p)]),
    "ap_vit_destroy": (None, [C.c_void_p]),
    "ap_vit_set_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]),
    "ap_vit_set_params": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int]),
    "ap_vit_finalize": (C.c_int, [C.c_void_p]),
    "ap_vit_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "ap_vit_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "ap_vit_embed_dim": (C.c_int, [C.c_void_p]),
    "ap_vit_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "ap_vit_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.c_int]),
    "ap_vit_forward_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float),
                                    C.POINTER(C.c_float), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ap_vit_forward_chw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_size_t, C.c_void_p]),
    "ap_gemm": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ap_gemm_fused": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ap_stream_init": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ap_rowstats_finalize": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "ap_gemm_trace": (C.c_int, [C.c_void_p, C.c_int]),
    "ap_split_f16_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ap_gemm_split_f16": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                    C.c_void_p, C.c_int, C.c_void_p]),
    "ap_gemm_split_f16_windows": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                            C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ap_layernorm": (C.c_int, [C.c_int, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                               C.c_float, C.c_void_p, C.c_void_p]),
    "ap_attention": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ap_sgemm": (C.c_int, [C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int,
                           C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_long, C.c_long,
                           C.c_void_p, C.c_long, C.c_long, C.c_void_p]),
    "ap_sgemm_stacked": (C.c_int, [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p]),
    "ap_softmax_rows": (C.c_int, [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p]),
    "ap_sattention_f32": (C.c_int, [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_long, C.c_void_p]),
    "ap_sattention_split_f16": (C.c_int, [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_long, C.c_void_p]),
    "ap_sam2_patchify": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p,
                                   C.c_void_p]),
    "ap_window_partition": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ap_window_unpartition": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ap_window_unpartition_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                            C.c_void_p]),
    "ap_maxpool2x2": (C.c_int, [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ap_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ap_add_rowvec": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "ap_gelu": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "ap_upsample2x_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ap_convt2x2_shuffle": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p]),
    "ap_bilinear_up4_threshold": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "ap_contours_from_mask": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_double,
                                        C.c_double, C.POINTER(C.c_void_p), C.c_void_p]),
    "ap_contours_destroy": (None, [C.c_void_p]),
    "ap_contours_count": (C.c_int, [C.c_void_p]),
    "ap_contours_num_holes": (C.c_int, [C.c_void_p, C.c_int]),
    "ap_contours_points": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "ap_grid_coords": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                 C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p]),
    "ap_clock_probe": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ap_synth_tiles": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                 C.c_uint32, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "ap_gather2d_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ap_synth_region": (C.c_int, [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_uint32,
                                  C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "ap_pillow_reduce_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p]),
    "ap_host_openslide_available": (C.c_int, []),
    "ap_host_openslide_open": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "ap_host_openslide_read_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ap_host_openslide_close": (None, [C.c_void_p]),
}


def library_path() -> str:
    """The in-tree product library; ATLASPATCH_HIP_LIB names another build of the same ABI (tools only: the debug
    library with the GEMM A/B twin, `make -C atlaspatch_amd/csrc twin`)."""
    override = os.environ.get("ATLASPATCH_HIP_LIB")
    if override:
        return override
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)


def load():
    """Load (once) and return the ctypes handle.  Raises HipLibraryError when the .so is absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = library_path()
        if not os.path.exists(path):
            raise HipLibraryError(
                f"{path} not found: the HIP extension has not been built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C atlaspatch_amd/csrc`. "
                "There is no CPU fallback.")
        try:
            # torch bundles its own libamdhip64.so.7; import it first so both share ONE HIP runtime
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is optional for symbol checks
            pass
        lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        # the ABI question first, on the two symbols it needs: a stale library must be refused with "rebuild", not fail
        # on the first symbol it lacks
        rebuild = "Rebuild the library (`make -C atlaspatch_amd/csrc`)."
        try:
            lib.ap_abi_version.restype, lib.ap_abi_version.argtypes = SIGNATURES["ap_abi_version"]
            got = lib.ap_abi_version()
        except AttributeError as exc:
            raise HipLibraryError(f"{path} does not export ap_abi_version: not a build of include/atlaspatch_hip.h. {rebuild}") from exc
        size = None
        if hasattr(lib, "ap_sizeof_vit_config"):
            lib.ap_sizeof_vit_config.restype, lib.ap_sizeof_vit_config.argtypes = SIGNATURES["ap_sizeof_vit_config"]
            size = lib.ap_sizeof_vit_config()
        if got != ABI_VERSION or size != C.sizeof(VitConfig):
            raise HipLibraryError(
                f"{path} reports ABI {got} with a {size}-byte ap_vit_config; this binding is written "
                f"for ABI {ABI_VERSION} / {C.sizeof(VitConfig)} bytes (include/atlaspatch_hip.h). {rebuild}")
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as exc:
                raise HipLibraryError(f"{path} (ABI {got}) does not export {name}, which include/atlaspatch_hip.h declares: "
                                      f"the library is older than this binding. {rebuild}") from exc
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return _lib


def check(code: int, what: str = "") -> None:
    if code != AP_OK:
        msg = load().ap_last_error()
        raise HipLibraryError(f"{what or 'libatlaspatch_hip'} failed with code {code}: "
                              f"{msg.decode(errors='replace') if msg else ''}")


def f3(values):
    arr = (C.c_float * 3)(*[float(v) for v in values])
    return arr


def torch_dtype_code(dtype) -> int:
    import torch
    table = {torch.float32: AP_F32, torch.float16: AP_F16, torch.bfloat16: AP_BF16}
    if dtype not in table:
        raise ValueError(f"unsupported compute dtype {dtype}")
    return table[dtype]


def current_stream_ptr(device=None) -> int:
    import torch
    return int(torch.cuda.current_stream(device).cuda_stream)
