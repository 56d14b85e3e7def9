#!/bin/bash
# round 5, final tree: full GPU suite, race screens, smoke(), evidence (bench line + kernel trace + PMC passes)
out=gpurun_out/${1:-r05m}; mkdir -p $out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q -s > $out/pytest_gpu.log 2>&1; tail -2 $out/pytest_gpu.log; grep "^PARITY" $out/pytest_gpu.log > $out/parity_lines.txt )
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python tools/forward_repeat.py > $out/forward_repeat.log 2>&1; tail -1 $out/forward_repeat.log
timeout 600 python tools/sam2_repeat.py 2>&1 | tail -1
bash tools/collect_evidence.sh ${1:-r05m} > $out/evidence.log 2>&1
python - <<PY
import json
d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["clock"]["shader_clock_GHz"], json.dumps(d.get("sam2")))
PY
