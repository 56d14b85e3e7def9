#!/bin/bash
out=gpurun_out/${1:-r05e}; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python tools/attn_check.py > $out/attn_stream.log 2>&1; echo "stream rc=$?"; grep -E "FAIL|^AP_ATTN" $out/attn_stream.log; grep -c "^ok" $out/attn_stream.log
AP_ATTN_IMPL=flash timeout 600 python tools/attn_check.py > $out/attn_flash.log 2>&1; echo "flash rc=$?"; grep -E "FAIL|^AP_ATTN" $out/attn_flash.log
timeout 900 python -m pytest tests -m gpu -x -q -k "attention or attn or batch_cut or repeat" > $out/pytest_attn.log 2>&1; tail -3 $out/pytest_attn.log
timeout 600 python tools/forward_repeat.py > $out/forward_repeat.log 2>&1; tail -2 $out/forward_repeat.log
python bench.py --no-cpu-baseline --no-extras > $out/bench_quick.json 2> $out/bench_quick.err; python - <<PY
import json
d=json.loads(open("$out/bench_quick.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("kernel_ms_per_step"), d["clock"]["shader_clock_GHz"])
PY
AP_ATTN_IMPL=flash python bench.py --no-cpu-baseline --no-extras > $out/bench_quick_flash.json 2> $out/bench_quick_flash.err; python - <<PY
import json
d=json.loads(open("$out/bench_quick_flash.json").read().strip().splitlines()[-1])
print("flash:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("kernel_ms_per_step"), d["clock"]["shader_clock_GHz"])
PY
