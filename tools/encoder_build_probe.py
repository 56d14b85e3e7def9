#!/usr/bin/env python3
"""Where the encoder build time goes (checkpoint file -> ready HipViT): load_file, key mapping, per-tensor upload, finalize.
usage: encoder_build_probe.py [arch=vit_b_16] [dtype=float16]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
from atlaspatch_amd import _lib
from atlaspatch_amd.encoders import vit as V
arch = sys.argv[1] if len(sys.argv) > 1 else "vit_b_16"
dt = getattr(torch, sys.argv[2] if len(sys.argv) > 2 else "float16")
torch.zeros(1, device="cuda"); lib = _lib.load()
with tempfile.TemporaryDirectory() as tmp:
    from safetensors.torch import save_file
    p = os.path.join(tmp, f"{arch}.safetensors")
    save_file(V.random_canonical_state_dict(V.ARCHS[arch], 0), p)
    print("file MB", os.path.getsize(p) / 1e6)
    for rep in range(3):
        t0 = time.perf_counter(); sd = V.load_checkpoint(V.Path(p)); t1 = time.perf_counter()
        spec = dict(V.ARCHS[arch])
        state = V.canonical_state_dict(sd, depth=spec["depth"], layer_scale=bool(spec.get("layer_scale")), source="auto"); t2 = time.perf_counter()
        vit = V.HipViT(spec, state, device=torch.device("cuda:0"), dtype=dt); torch.cuda.synchronize(); t3 = time.perf_counter()
        print(f"rep {rep}: load_file {1e3*(t1-t0):.1f} ms, canonical {1e3*(t2-t1):.1f} ms, HipViT(create+set_param+finalize) {1e3*(t3-t2):.1f} ms", flush=True)
        # split HipViT: numpy conversion vs set_param vs finalize
        cfg_t0 = time.perf_counter()
        arrs = {k: np.ascontiguousarray(v.detach().to(torch.float32).cpu().numpy()) for k, v in state.items()}
        t4 = time.perf_counter()
        print(f"        to-numpy of {len(arrs)} tensors {1e3*(t4-cfg_t0):.1f} ms ({sum(a.nbytes for a in arrs.values())/1e6:.0f} MB)")
        vit.release()
