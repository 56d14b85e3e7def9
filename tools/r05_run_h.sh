#!/bin/bash
out=gpurun_out/${1:-r05h}; mkdir -p $out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q -s > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log; grep "^PARITY" $out/pytest_gpu.log > $out/parity_lines.txt; wc -l $out/parity_lines.txt )
timeout 900 python tools/forward_repeat.py > $out/forward_repeat.log 2>&1; tail -2 $out/forward_repeat.log
timeout 600 python tools/batch_invariance.py > $out/batch_invariance.log 2>&1; tail -2 $out/batch_invariance.log
bash tools/collect_evidence.sh ${1:-r05h} > $out/evidence.log 2>&1; tail -1 $out/bench.json | cut -c1-400
