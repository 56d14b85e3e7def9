#!/bin/bash
out=gpurun_out/${1:-r05c}; mkdir -p $out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log )
ATLASPATCH_HIP_LIB=atlaspatch_amd/libatlaspatch_hip_twin_old.so timeout 900 python tools/gemm_twin_ab.py > $out/twin_old.log 2>&1
echo "== old-loop twin (257) vs product (256): rc=$?"; grep -E "bit-equality|repeatability|FAIL" $out/twin_old.log | head -8
python - <<PY
import json
r=json.loads(open("$out/twin_old.log").read().strip().splitlines()[-1])
by={}
for x in r: by.setdefault(x["gemm"],{})[(x["impl"],x["ablate"])]=x["ms_median"]
for g,d in by.items(): print("   %-16s"%g, "  ".join("%s:%.4f"%(k,v) for k,v in d.items()), " new/old %.4f"%(d[(256,0)]/d[(257,0)]))
PY
ATLASPATCH_HIP_LIB=atlaspatch_amd/libatlaspatch_hip_twin_diag.so timeout 600 python tools/gemm_epilogue_trace.py > $out/epilogue_trace.txt 2>&1; tail -12 $out/epilogue_trace.txt
bash tools/collect_evidence.sh ${1:-r05c} > $out/evidence.log 2>&1; tail -1 $out/bench.json | cut -c1-600
