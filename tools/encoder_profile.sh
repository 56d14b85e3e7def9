set -u
export TMPDIR=/tmp
for enc in uni_v1:2048 conch_v1:256; do
  name=${enc%%:*}; b=${enc##*:}
  O=$PWD/gpurun_out/prof_$name; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python tools/encoder_breakdown.py $name $b > $O/run.log 2>&1
  tail -1 $O/run.log | cut -c1-600
  DB=$(find $O/kt -name '*_results.db' | head -1)
  python profiles/summarize_rocpd.py "$DB" > $O/kernel_stats.txt
  head -14 $O/kernel_stats.txt | cut -c1-150
  rm -rf $O/kt
done
