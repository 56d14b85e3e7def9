import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from atlaspatch_amd import _lib
dev = torch.device("cuda:0"); lib = _lib.load(); stream = _lib.current_stream_ptr(dev)
torch.manual_seed(0)
def run(qkv, n, T, H):
    out = torch.full((n * T, H * 64), float("nan"), device=dev, dtype=qkv.dtype)
    _lib.check(lib.ap_attention(1, qkv.data_ptr(), out.data_ptr(), n, T, H, 64, stream)); torch.cuda.synchronize()
    return out
n, T, H = 1, 64, 1
# 1) q = k = 0 -> uniform softmax; v[t][c] = t + c/100 -> out[q][c] = mean_t + c/100
qkv = torch.zeros((T, 3 * 64), device=dev, dtype=torch.float16)
t = torch.arange(T, device=dev).float()[:, None]; c = torch.arange(64, device=dev).float()[None, :]
qkv[:, 128:] = (t * 0 + c).half()          # v[t][c] = c
o = run(qkv, n, T, H); print("v=c      : out[0,:8]", o[0, :8].float().tolist(), " out[5,60:]", o[5, 60:].float().tolist())
qkv[:, 128:] = (t + 0 * c).half()          # v[t][c] = t  -> mean = 31.5
o = run(qkv, n, T, H); print("v=t      : out[0,:4]", o[0, :4].float().tolist(), "(want 31.5)")
# 2) one-hot attention: q[i] . k[j] large when i == j  -> out[i] = v[i]
qkv = torch.zeros((T, 3 * 64), device=dev, dtype=torch.float16)
eye = torch.eye(64, device=dev).half() * 16.0
qkv[:, 0:64] = eye; qkv[:, 64:128] = eye * 8
qkv[:, 128:] = (t * 1.0 + c / 100).half()
o = run(qkv, n, T, H); print("one-hot  : out[i,0] for i<12", o[:12, 0].float().tolist(), " out[3,:4]", o[3, :4].float().tolist())
