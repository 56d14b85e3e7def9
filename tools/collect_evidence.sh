#!/bin/bash
# Round evidence on one MI355X: bench line, rocprofv3 kernel trace of the same command, and the two PMC passes for
# HBM traffic (each counter in its own run, kernel-trace only, as MI355X_MICROARCH.md prescribes).
# usage: tools/collect_evidence.sh <tag>      (outputs under gpurun_out/<tag>/; copy the summaries into profiles/)
set -u
TAG=${1:-r01c}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -1 "$OUT/bench.json"
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt -- $CMD > "$OUT/kt.log" 2>&1
DB=$(find "$OUT/kt" -name '*_results.db' | head -1)
if [ -n "$DB" ]; then python profiles/summarize_rocpd.py "$DB" > "$OUT/kernel_stats.txt"; else
  find "$OUT/kt" -name '*kernel_stats.csv' | head -1 | xargs -r cp -t "$OUT"; fi
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o f -- $CMD > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o w -- $CMD > "$OUT/pmc_write.log" 2>&1
python tools/pmc_traffic.py "$OUT/pmc_fetch" "$OUT/pmc_write" > "$OUT/traffic.json"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma" -o m -- $CMD > "$OUT/pmc_mfma.log" 2>&1
head -1 $(find "$OUT/pmc_mfma" -name "*counter_collection.csv" | head -1) > "$OUT/pmc_mfma_columns.txt" 2>/dev/null
python tools/pmc_mfma.py "$OUT/pmc_mfma" > "$OUT/mfma_util.json"
# SQ counters of the same command, two passes of eight (issue mix, waits, VALU beside MFMA): tools/pmc_sq.py
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d "$OUT/pmc_sqa" -o a -- $CMD > "$OUT/pmc_sqa.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pmc_sqb" -o b -- $CMD > "$OUT/pmc_sqb.log" 2>&1
python tools/pmc_sq.py "$OUT/pmc_sqa" "$OUT/pmc_sqb" > "$OUT/sq.json"
rm -rf "$OUT/kt" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_mfma" "$OUT/pmc_sqa" "$OUT/pmc_sqb"
head -30 "$OUT/kernel_stats.txt"
