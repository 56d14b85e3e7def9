#!/bin/bash
out=gpurun_out/${1:-r05d}; mkdir -p $out
export TMPDIR=/tmp
ATLASPATCH_HIP_LIB=atlaspatch_amd/libatlaspatch_hip_twin_nopf.so timeout 900 python tools/gemm_twin_ab.py > $out/twin_nopf.log 2>&1
echo "== twin without the window prefetch (257) vs product (256): rc=$?"; grep -E "bit-equality|repeatability|FAIL" $out/twin_nopf.log | head -8
python - <<PY
import json
r=json.loads(open("$out/twin_nopf.log").read().strip().splitlines()[-1])
by={}
for x in r: by.setdefault(x["gemm"],{})[(x["impl"],x["ablate"])]=x["ms_median"]
for g,d in by.items(): print("   %-16s"%g, "  ".join("%s:%.4f"%(k,v) for k,v in d.items()), " with/without %.4f"%(d[(256,0)]/d[(257,0)]))
PY
python bench.py --no-cpu-baseline --no-extras > $out/bench_quick.json 2> $out/bench_quick.err; tail -1 $out/bench_quick.json | cut -c1-300
