#!/bin/bash
out=gpurun_out/${1:-r05k}; mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -k "sam2 or segment" > $out/pytest_sam2.log 2>&1; tail -3 $out/pytest_sam2.log
python tools/sam2_batch_time.py 2>&1 | grep -v amdgpu > $out/sam2_batch_time_wide.txt; cat $out/sam2_batch_time_wide.txt
ATLASPATCH_SAM2_NO_WIDE_GEMM=1 python tools/sam2_batch_time.py 2>&1 | grep -v amdgpu > $out/sam2_batch_time_nowide.txt; cat $out/sam2_batch_time_nowide.txt
timeout 600 python tools/sam2_repeat.py 2>&1 | tail -2
python tools/sam2_breakdown.py 2>&1 | grep -v amdgpu > $out/sam2_breakdown.txt; head -12 $out/sam2_breakdown.txt
