import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atlaspatch_amd.encoders.vit import build_hip_vit_extractor
ex = build_hip_vit_extractor(name="vit_b_16", arch="vit_b_16", device=torch.device("cuda:0"), dtype=torch.float16, random_init_seed=0, max_batch=2048)
rng = np.random.default_rng(0)
patches = [rng.integers(0, 256, (256, 256, 3), dtype=np.uint8) for _ in range(32)]
for _ in range(5): a = ex.extract_batch(patches, batch_size=32)
t0 = time.perf_counter()
for _ in range(50): b = ex.extract_batch(patches, batch_size=32)
dt = (time.perf_counter() - t0) / 50
print(f"extract_batch(32): {dt*1e3:.3f} ms per call, {32/dt:.0f} patches/s, repeatable={np.array_equal(a, b)}")
# where the call's time goes (each part synchronised, so the parts add up to a little more than the call)
def tm(fn, reps=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3, r
t_prep, host = tm(lambda: torch.from_numpy(ex._prepare(patches)))
t_h2d, dev = tm(lambda: host.to(ex.device, non_blocking=True))
out = torch.empty((32, ex.embedding_dim), dtype=torch.float32, device=ex.device)
t_fwd, _ = tm(lambda: ex.forward_device(dev, out))
t_d2h, _ = tm(lambda: out.cpu().numpy())
print(f"gather into pinned {t_prep:.3f} ms, H2D {t_h2d:.3f} ms, forward {t_fwd:.3f} ms, D2H {t_d2h:.3f} ms")
# same-process A/B of the call's host side: rounds 1-5 gathered ALL patches into the pinned buffer, then copied, then ran
def old_call():
    host = torch.from_numpy(ex._prepare(patches))
    o = torch.empty((32, ex.embedding_dim), dtype=torch.float32, device=ex.device)
    ex.forward_device(host.to(ex.device, non_blocking=True), o)
    return o.cpu().numpy()
for name, fn in (("gather-all, then copy (rounds 1-5)", old_call), ("pieces of 8 (round 6)", lambda: ex.extract_batch(patches, batch_size=32))) * 2:
    for _ in range(5): fn()
    t0 = time.perf_counter()
    for _ in range(100): r = fn()
    dt = (time.perf_counter() - t0) / 100
    print(f"A/B {name}: {dt*1e3:.3f} ms per call, {32/dt:.0f} patches/s, equal={np.array_equal(r, a)}")
