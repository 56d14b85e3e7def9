set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/attn_pmc; rm -rf $O; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/sq_counters.txt
wc -l $O/sq_counters.txt
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_MFMA SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o p -- python tools/attn_shape_run.py 256 785 12 3 > $O/p$i.log 2>&1 || tail -3 $O/p$i.log
  python tools/pmc_any.py $O/p$i attention
done
rm -rf $O/p?
