#!/bin/bash
# Samples rocm-smi (power, shader clock) while bench.py runs a long timed region: direct evidence for the power-bound claim.
# usage: tools/power_clock_sample.sh <out_file>
OUT=${1:-gpurun_out/power_clock.txt}
mkdir -p "$(dirname "$OUT")"
python bench.py --steps 250 --warmup 2 --no-cpu-baseline > "$OUT.bench.json" 2>/dev/null &
BP=$!
sleep 25        # model build + slide synthesis, then the timed region (~20 s)
{
  echo "# rocm-smi samples every 0.5 s while bench.py --steps 250 runs (f16, B=2048)"
  for i in $(seq 1 30); do
    /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr -s ' ' | tr '\n' ';'
    echo
    sleep 0.5
  done
} > "$OUT"
wait $BP
tail -1 "$OUT.bench.json" | cut -c1-200 >> "$OUT"
