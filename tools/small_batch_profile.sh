# rocprofv3 kernel trace of extract_batch(32): true kernel durations vs the 2.25-ms forward
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof_b32; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python tools/extract_batch_probe.py > $O/run.log 2>&1
tail -2 $O/run.log
DB=$(find $O/kt -name '*_results.db' | head -1)
python profiles/summarize_rocpd.py "$DB" > $O/kernel_stats.txt
head -24 $O/kernel_stats.txt | cut -c1-140
rm -rf $O/kt
