# HBM-side traffic of one attention shape (FETCH_SIZE / WRITE_SIZE in separate passes): bash tools/pmc_attn_traffic.sh n T H
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/attn_traffic; rm -rf $O; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -o p -- python tools/attn_shape_run.py $1 $2 $3 3 > $O/$c.log 2>&1 || tail -3 $O/$c.log
  python tools/pmc_any.py $O/$c attention
done
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
