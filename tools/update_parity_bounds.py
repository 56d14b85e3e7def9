#!/usr/bin/env python3
"""Rewrites the MEASURED tables of tests/test_gpu_parity.py and tests/test_encoder_zoo.py from the PARITY lines of a
`pytest -m gpu -s` log (the tests print every measured statistic beside its bound): for every (case, dtype) key that has an entry,
the entry becomes the largest measured triple among the log's lines of that key, rounded UP to three significant digits.
Bounds only ever TIGHTEN automatically: a statistic that measures above its entry keeps the old (smaller) value -- the test then
fails, which is the point of the table.  Raising an entry needs `--allow-raise "<reason>"`; the reason is printed beside every
raised entry and belongs in the commit message.  The north star's fixed asserts (1e-3 norm-wise) are not part of the tables.
usage: update_parity_bounds.py <log> [--dry] [--allow-raise "<reason>"]"""
import math, re, sys
log = open(sys.argv[1]).read()
dry = "--dry" in sys.argv
raise_reason = sys.argv[sys.argv.index("--allow-raise") + 1] if "--allow-raise" in sys.argv and sys.argv.index("--allow-raise") + 1 < len(sys.argv) else None
if "--allow-raise" in sys.argv and not raise_reason:
    sys.exit("--allow-raise needs a reason")
refused = 0
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
        global changed, refused
        key = (mo.group(1), mo.group(2))
        if key not in meas: return mo.group(0)
        old = tuple(float(x) for x in mo.group(3).split(","))
        new = tuple(float(up3(v)) for v in meas[key])
        if all(abs(n - o) <= 0.02 * o for n, o in zip(new, old)): return mo.group(0)
        up = [i for i, (n, o) in enumerate(zip(new, old)) if n > o]
        if up and not raise_reason:
            refused += 1
            print(f"{path}: {key}: measured {new} is ABOVE the entry {old} in position(s) {up}: kept (a regression must fail the test; "
                  f"--allow-raise \"<reason>\" to record a new bound)")
            new = tuple(min(n, o) for n, o in zip(new, old))
            if new == old: return mo.group(0)
        changed += 1
        print(f"{path}: {key}: {old} -> {new}" + (f"   RAISED: {raise_reason}" if up and raise_reason else ""))
        return f'("{key[0]}", "{key[1]}"): ({", ".join(up3(v) for v in new)})'
    out = entry.sub(sub, src)
    if not dry and changed:
        open(path, "w").write(out)
    print(f"{path}: {changed} entries changed")
if refused:
    print(f"{refused} entries would have been raised and were kept")
