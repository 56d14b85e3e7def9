# LDS-side counters of the persistent GEMM on the ViT-B shapes (is the fragment traffic conflict-free?)
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/gemm_lds; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o p -- python tools/fused_gemm_bench.py > $O/p$i.log 2>&1 || tail -3 $O/p$i.log
  python tools/pmc_any.py $O/p$i gemm256_kernelIDF16_Li5
  python tools/pmc_any.py $O/p$i gemm256_kernelIDF16_Li4
done
rm -rf $O/p?
