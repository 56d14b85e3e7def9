"""INTEGRATION.md is executable: the reference-side bindings it shows are checked against the built library.

Round 4's verdict found the document's ``VitConfig`` 36 bytes shorter than the library's ``ap_vit_config``: a maintainer who
pasted the stub handed ``ap_vit_create`` a structure it read past.  Two things changed (ABI v20): the structure carries its
own size and the library never reads more than that, and this file holds the document to the header:

* CPU: every fenced Python block is parsed; its ``ctypes.Structure`` classes and ``lib.<fn>.argtypes / restype`` assignments
  are EXECUTED against the built ``.so`` and compared with the header's prototypes (arity and C type of every parameter)
  and with ``ap_sizeof_vit_config()``; the size rules of ``ap_vit_create`` are exercised (they run before any HIP call).
* GPU: the plugin of section 2.1 is loaded AS WRITTEN (the reference's ``atlas_patch`` ABC and torchvision's weights are
  the only stand-ins) through the reference's hook convention ``register_feature_extractors(registry=, device=, dtype=,
  num_workers=)`` (models/patch/custom.py:113-146) and its features are held against golden G1, the outputs of the
  reference's own ``extract_batch`` (models/patch/base.py:76-107).
"""
import ast
import ctypes as C
import os
import re
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOC = os.path.join(ROOT, "INTEGRATION.md")
HEADER = os.path.join(ROOT, "include", "atlaspatch_hip.h")


# ----------------------------------------------------------------------------- the header, as data
def _strip_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


_C_SCALARS = {"int": C.c_int, "size_t": C.c_size_t, "double": C.c_double, "float": C.c_float, "int64_t": C.c_int64,
              "uint32_t": C.c_uint32, "long": C.c_long, "long long": C.c_longlong, "ap_stream_t": C.c_void_p}


def _param_kind(decl):
    """('ptr' | 'int' | 'float', size in bytes) of one C parameter declaration."""
    decl = decl.strip()
    if "*" in decl or "[" in decl:
        return ("ptr", C.sizeof(C.c_void_p))
    words = [w for w in re.sub(r"\bconst\b", " ", decl).split()]
    name_stripped = words[:-1] if len(words) > 1 else words          # drop the parameter name
    ctype = " ".join(name_stripped)
    t = _C_SCALARS[ctype]
    if t is C.c_void_p:
        return ("ptr", C.sizeof(t))
    return ("float" if t in (C.c_double, C.c_float) else "int", C.sizeof(t))


def header_prototypes():
    """{name: (return kind, [param kinds])} for every function the header declares."""
    text = _strip_comments(open(HEADER).read())
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    out = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(ap_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, params = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        if "typedef" in ret:
            continue
        kinds = [] if params in ("void", "") else [_param_kind(p) for p in params.split(",")]
        if "*" in ret:
            rkind = ("ptr", C.sizeof(C.c_void_p))
        elif ret == "void":
            rkind = None
        else:
            rkind = _param_kind(ret + " x")
        out[name] = (rkind, kinds)
    return out


def _ctypes_kind(t):
    if t is None:
        return None
    if t in (C.c_void_p, C.c_char_p) or isinstance(t, type) and issubclass(t, (C._Pointer, C.Array)):
        return ("ptr", C.sizeof(C.c_void_p))
    if t in (C.c_float, C.c_double):
        return ("float", C.sizeof(t))
    return ("int", C.sizeof(t))


# ----------------------------------------------------------------------------- the document, as code
def doc_blocks():
    text = open(DOC).read()
    return re.findall(r"```python\n(.*?)```", text, flags=re.S)


def _is_lib_attr(node, attr):
    """lib.<fn>.<attr>"""
    return (isinstance(node, ast.Attribute) and node.attr == attr and isinstance(node.value, ast.Attribute)
            and isinstance(node.value.value, ast.Name) and node.value.value.id == "lib")


def doc_bindings(lib):
    """Execute the ctypes declarations of every block: returns ({fn: argtypes}, {fn: restype}, {class name: Structure})."""
    argtypes, restypes, structs = {}, {}, {}
    from atlaspatch_amd import _lib
    lib = C.CDLL(_lib.library_path())            # a handle of its own: the assignments below must not re-type the product's
    for block in doc_blocks():
        try:
            tree = ast.parse(block)
        except SyntaxError as exc:                                   # a block that is not valid Python is a finding too
            raise AssertionError(f"INTEGRATION.md has a python block that does not parse: {exc}\n{block[:300]}")
        ns = {"C": C, "lib": lib, "np": np}
        for node in tree.body:
            if isinstance(node, ast.ClassDef) and any(isinstance(b, ast.Attribute) and b.attr == "Structure" for b in node.bases):
                exec(compile(ast.Module([node], []), DOC, "exec"), ns)
                structs[node.name] = ns[node.name]
            elif isinstance(node, ast.Assign) and len(node.targets) == 1 and (
                    _is_lib_attr(node.targets[0], "argtypes") or _is_lib_attr(node.targets[0], "restype")):
                exec(compile(ast.Module([node], []), DOC, "exec"), ns)
                fn = node.targets[0].value.attr
                val = eval(compile(ast.Expression(node.value), DOC, "eval"), ns)
                (argtypes if node.targets[0].attr == "argtypes" else restypes)[fn] = val
    return argtypes, restypes, structs


@pytest.fixture(scope="module")
def lib():
    from atlaspatch_amd import _lib
    return _lib.load()


# ----------------------------------------------------------------------------- CPU: document == header == library
def test_header_prototypes_parse_and_cover_every_symbol():
    from atlaspatch_amd import _lib
    protos = header_prototypes()
    assert set(protos) == set(_lib.SIGNATURES), set(protos) ^ set(_lib.SIGNATURES)


def test_python_mirror_matches_the_header_parameter_by_parameter():
    """atlaspatch_amd/_lib.py::SIGNATURES (what the product binds) against the header: arity, and per parameter pointer /
    integer / floating kind and width."""
    from atlaspatch_amd import _lib
    protos = header_prototypes()
    for name, (res, args) in _lib.SIGNATURES.items():
        rkind, kinds = protos[name]
        assert len(args) == len(kinds), (name, len(args), len(kinds))
        assert [_ctypes_kind(a) for a in args] == kinds, (name, [_ctypes_kind(a) for a in args], kinds)
        assert _ctypes_kind(res) == rkind, (name, res, rkind)


def test_every_binding_in_the_document_matches_the_header(lib):
    argtypes, restypes, structs = doc_bindings(lib)
    protos = header_prototypes()
    assert len(argtypes) >= 15 and "ap_vit_create" in argtypes and "ap_grid_coords" in argtypes   # the scan found the stubs
    for fn, args in argtypes.items():
        assert fn in protos, f"INTEGRATION.md binds {fn}, which the header does not declare"
        assert hasattr(lib, fn)
        rkind, kinds = protos[fn]
        assert len(args) == len(kinds), f"INTEGRATION.md: {fn}.argtypes has {len(args)} entries, the header's prototype {len(kinds)}"
        got = [_ctypes_kind(a) for a in args]
        assert got == kinds, f"INTEGRATION.md: {fn}.argtypes kinds {got} != header {kinds}"
    for fn, res in restypes.items():
        assert fn in protos and _ctypes_kind(res) == protos[fn][0], (fn, res, protos[fn][0])
    # a function whose C return type is not int must have its restype declared wherever its argtypes are
    for fn in argtypes:
        if protos[fn][0] not in (("int", 4), None):
            assert fn in restypes, f"INTEGRATION.md binds {fn} without a restype; it returns {protos[fn][0]}"


def test_document_structures_have_the_librarys_size_and_layout(lib):
    from atlaspatch_amd import _lib
    _, _, structs = doc_bindings(lib)
    assert "VitConfig" in structs
    doc_cfg = structs["VitConfig"]
    assert C.sizeof(doc_cfg) == lib.ap_sizeof_vit_config() == C.sizeof(_lib.VitConfig)
    want = [(n, getattr(_lib.VitConfig, n).offset, getattr(_lib.VitConfig, n).size) for n, _ in _lib.VitConfig._fields_]
    got = [(n, getattr(doc_cfg, n).offset, getattr(doc_cfg, n).size) for n, _ in doc_cfg._fields_]
    assert got == want
    # ... and the header's own field list, in order
    text = _strip_comments(open(HEADER).read())
    body = re.search(r"typedef struct ap_vit_config \{(.*?)\} ap_vit_config;", text, flags=re.S).group(1)
    fields = re.findall(r"\b(?:uint32_t|int|float)\s+([a-z_0-9]+)\s*;", body)
    assert fields == [n for n, _ in doc_cfg._fields_]
    m = re.search(r"#define AP_ABI_VERSION (\d+)", open(HEADER).read())
    assert int(m.group(1)) == lib.ap_abi_version() == _lib.ABI_VERSION
    assert f"AP_ABI_VERSION = {m.group(1)}" in open(DOC).read()


def test_vit_create_never_reads_past_the_declared_size(lib):
    """The size rules run before any HIP call, so they are checked here: a binder written for ABI <= 19 (no struct_size, image_size
    = 224 first), a zero, a truncated and an odd size are all refused with a message and WITHOUT reading past the declared
    object (each buffer below is exactly as long as a v20 structure; under a sanitizer an over-read would trap)."""
    from atlaspatch_amd import _lib
    full = lib.ap_sizeof_vit_config()
    assert full >= 92

    def create(raw: bytes):
        buf = C.create_string_buffer(raw, len(raw))
        h = C.c_void_p()
        rc = lib.ap_vit_create(C.cast(buf, C.POINTER(_lib.VitConfig)), C.byref(h))
        assert rc != 0 or h.value
        if rc == 0:
            lib.ap_vit_destroy(h)
        return rc, lib.ap_last_error().decode()

    good = _lib.VitConfig(224, 16, 768, 2, 12, 3072, 1e-6, 0, 1, 0, 0, 0, 1e-5)
    assert good.struct_size == full
    raw = bytes(good)
    v19 = raw[4:]                                                    # the pre-v20 layout: starts with image_size
    rc, msg = create(v19 + b"\0" * 4)                                # 92 bytes in all: a read of "224 bytes" would run past them
    assert rc == _lib.AP_ERR_INVALID and "struct_size" in msg and "224" in msg
    for bad in (0, 4, 52, 88, full + 2, 448, 518, 1 << 20):
        rc, msg = create(int(bad).to_bytes(4, "little") + raw[4:])
        assert rc == _lib.AP_ERR_INVALID and "struct_size" in msg, (bad, rc, msg)
    # a newer caller (8 more bytes than this library knows) on this library: refused as such, tail unread
    rc, msg = create((full + 8).to_bytes(4, "little") + raw[4:])
    assert rc == _lib.AP_ERR_UNSUPPORTED and "newer" in msg
    # ap_vit_config_init: zero-fill + size
    cfg = _lib.VitConfig(1, 2, 3)
    assert lib.ap_vit_config_init(C.byref(cfg), C.sizeof(cfg)) == 0
    assert cfg.struct_size == full and cfg.image_size == 0 and cfg.dim == 0
    assert lib.ap_vit_config_init(C.byref(cfg), 52) == _lib.AP_ERR_INVALID
    assert lib.ap_vit_config_init(None, full) == _lib.AP_ERR_INVALID


# ----------------------------------------------------------------------------- GPU: the plugin of section 2.1, as written
def _plugin_source():
    for block in doc_blocks():
        if block.startswith("# hip_vit_plugin.py"):
            return block
    raise AssertionError("INTEGRATION.md section 2.1: the plugin block is gone")


def _torchvision_state_dict(canon, depth):
    """Canonical names -> torchvision VisionTransformer names (what `vit_b_16(...).state_dict()` returns)."""
    sd = {"conv_proj.weight": canon["patch_embed.weight"], "conv_proj.bias": canon["patch_embed.bias"],
          "class_token": canon["cls_token"].reshape(1, 1, -1), "encoder.pos_embedding": canon["pos_embed"][None],
          "encoder.ln.weight": canon["norm.weight"], "encoder.ln.bias": canon["norm.bias"]}
    for i in range(depth):
        p, b = f"encoder.layers.encoder_layer_{i}.", f"blocks.{i}."
        sd.update({p + "ln_1.weight": canon[b + "ln1.weight"], p + "ln_1.bias": canon[b + "ln1.bias"],
                   p + "self_attention.in_proj_weight": canon[b + "qkv.weight"], p + "self_attention.in_proj_bias": canon[b + "qkv.bias"],
                   p + "self_attention.out_proj.weight": canon[b + "proj.weight"], p + "self_attention.out_proj.bias": canon[b + "proj.bias"],
                   p + "ln_2.weight": canon[b + "ln2.weight"], p + "ln_2.bias": canon[b + "ln2.bias"],
                   p + "mlp.0.weight": canon[b + "fc1.weight"], p + "mlp.0.bias": canon[b + "fc1.bias"],
                   p + "mlp.3.weight": canon[b + "fc2.weight"], p + "mlp.3.bias": canon[b + "fc2.bias"]})
    return sd


@pytest.mark.gpu
@pytest.mark.parametrize("dtype_name", ["float32", "float16"])
def test_section_2_1_plugin_runs_verbatim_and_matches_g1(dtype_name, tmp_path, monkeypatch, golden_dir):
    import torch
    from atlaspatch_amd import _lib
    from atlaspatch_amd.encoders import base as our_base
    from atlaspatch_amd.encoders.custom import register_feature_extractors_from_module
    from atlaspatch_amd.encoders.registry import PatchFeatureExtractorRegistry
    from atlaspatch_amd.encoders.vit import canonical_state_dict
    from oracle import vit_oracle
    from tests import helpers

    dtype = getattr(torch, dtype_name)
    plugin = tmp_path / "hip_vit_plugin.py"
    plugin.write_text(_plugin_source())                              # not one character changed

    # stand-ins for what the GPU box lacks: the reference package's ABC and torchvision's pretrained weights (here: the
    # seeded model golden G1 was generated with, under torchvision's key names)
    ref_base = types.ModuleType("atlas_patch.models.patch.base")
    ref_base.FeatureExtractor = our_base.FeatureExtractor
    for name in ("atlas_patch", "atlas_patch.models", "atlas_patch.models.patch"):
        monkeypatch.setitem(sys.modules, name, types.ModuleType(name))
    monkeypatch.setitem(sys.modules, "atlas_patch.models.patch.base", ref_base)
    canon = canonical_state_dict(dict(vit_oracle.make_hf_vit(layers=12).state_dict()), depth=12, layer_scale=False, source="hf")
    tv_sd = _torchvision_state_dict(canon, 12)

    class _Model:
        def state_dict(self):
            return tv_sd

    tv, tvm = types.ModuleType("torchvision"), types.ModuleType("torchvision.models")
    tvm.vit_b_16 = lambda weights=None: _Model()
    tvm.ViT_B_16_Weights = types.SimpleNamespace(IMAGENET1K_V1="IMAGENET1K_V1")
    tv.models = tvm
    monkeypatch.setitem(sys.modules, "torchvision", tv)
    monkeypatch.setitem(sys.modules, "torchvision.models", tvm)
    real_cdll = C.CDLL
    monkeypatch.setattr(C, "CDLL", lambda name, *a, **k: real_cdll(_lib.library_path() if name == "libatlaspatch_hip.so" else name, *a, **k))

    registry = PatchFeatureExtractorRegistry()
    register_feature_extractors_from_module(plugin, registry, device=torch.device("cuda:0"), dtype=dtype, num_workers=0)
    assert registry.available() == ["hip_vit_b_16"]
    ex = registry.create("hip_vit_b_16")
    try:
        assert isinstance(ex, our_base.FeatureExtractor) and ex.embedding_dim == 768
        empty = ex.extract_batch([])
        assert empty.shape == (0, 768) and empty.dtype == np.float32
        g = np.load(os.path.join(golden_dir, "extract_batch.npz"))
        patches = helpers.golden_patches((5,))[5]
        got = ex.extract_batch(patches, batch_size=32)
        want = g["L12_n5_out"]
        assert got.shape == want.shape and got.dtype == np.float32 and got.flags["C_CONTIGUOUS"]
        rel = float(np.linalg.norm(got - want) / np.linalg.norm(want))
        print(f"PARITY section-2.1 plugin {dtype_name}: {rel:.3e} vs the reference's extract_batch (G1 L12)")
        assert rel <= (2e-5 if dtype == torch.float32 else 1e-3)       # the north star's bound; measured 2e-6 / 8e-4
        # the same kernels as the packaged extractor: bit-equal features
        from atlaspatch_amd.encoders.vit import build_hip_vit_extractor
        ours = build_hip_vit_extractor(name="vit_b_16", arch="vit_b_16", depth=12, state_dict=tv_sd, device=torch.device("cuda:0"),
                                       dtype=dtype, source="torchvision")
        assert np.array_equal(ours.extract_batch(patches, batch_size=32), got)
        ours.cleanup()
    finally:
        ex.cleanup()
    ex.cleanup()                                                     # idempotent, as the reference's finally-block needs
