import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atlaspatch_amd.encoders.vit import build_hip_vit_extractor
dev = torch.device("cuda:0")
ex = build_hip_vit_extractor(name="vit_b_16", arch="vit_b_16", device=dev, dtype=torch.float16, random_init_seed=0, max_batch=2048)
tiles = torch.randint(0, 256, (2048, 256, 256, 3), dtype=torch.uint8, device=dev)
out = torch.empty((2048, 768), device=dev)
for _ in range(3): ex.forward_device(tiles, out)
torch.cuda.synchronize()
ex.vit.profile(True)
for _ in range(10): ex.forward_device(tiles, out)
torch.cuda.synchronize()
p = ex.vit.profile_read()
print(os.environ.get("ATLASPATCH_HIP_LIB", "new"), {k: round(v[0] / 10, 3) for k, v in p.items()}, "sum", round(sum(v[0] for v in p.values()) / 10, 2), float(out.double().abs().sum()))
