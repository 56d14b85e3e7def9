#!/usr/bin/env python3
"""Rewrites the MEASURED tables of tests/test_gpu_parity.py and tests/test_encoder_zoo.py from the PARITY lines of a
`pytest -m gpu -s` log (the tests print every measured statistic beside its bound): for every (case, dtype) key that has an entry,
the entry becomes the largest measured triple among the log's lines of that key, rounded UP to three significant digits.
usage: update_parity_bounds.py <log> [--dry]"""
import math, re, sys
log = open(sys.argv[1]).read()
dry = "--dry" in sys.argv
pat = re.compile(r"^PARITY (.+?) (float32|float16|bfloat16): norm-wise ([0-9.e+-]+) \(bound [^)]*\) element-wise max ([0-9.e+-]+) \([^)]*\) q99\.9 ([0-9.e+-]+)", re.M)
meas = {}
for what, dt, a, b, c in pat.findall(log):
    key = (what.split(" n=")[0], dt)
    t = (float(a), float(b), float(c))
    meas[key] = tuple(max(x, y) for x, y in zip(meas.get(key, (0, 0, 0)), t))
def up3(x):
    if x == 0: return "0.0"
    e = math.floor(math.log10(x))
    m = math.ceil(x / 10 ** e * 100 - 1e-9) / 100
    if m >= 10: m, e = m / 10, e + 1
    return f"{m:.2f}e{e:d}"
entry = re.compile(r'\(\s*"([^"]+)",\s*"(float32|float16|bfloat16)"\s*\)\s*:\s*\(([^)]*)\)')
for path in ("tests/test_gpu_parity.py", "tests/test_encoder_zoo.py"):
    src = open(path).read()
    changed = 0
    def sub(mo):
        global changed
        key = (mo.group(1), mo.group(2))
        if key not in meas: return mo.group(0)
        old = tuple(float(x) for x in mo.group(3).split(","))
        new = meas[key]
        if all(abs(n - o) <= 0.02 * o for n, o in zip(new, old)): return mo.group(0)
        changed += 1
        print(f"{path}: {key}: {old} -> {tuple(float(up3(v)) for v in new)}")
        return f'("{key[0]}", "{key[1]}"): ({", ".join(up3(v) for v in new)})'
    out = entry.sub(sub, src)
    if not dry and changed:
        open(path, "w").write(out)
    print(f"{path}: {changed} entries changed")
