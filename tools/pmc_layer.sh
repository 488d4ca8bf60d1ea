#!/bin/bash
# usage: pmc_layer.sh <layer> <outdir> ; collects PMC sets for tools/conv_bench.py --one <layer> under KBN_DEBUG=0/1/3
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; L=$1; O=$R/gpurun_out/$2; mkdir -p $O
for dbg in ${KBN_PMC_DBGS:-0 1 3}; do
 for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD" "GRBM_GUI_ACTIVE FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | cut -d" " -f1)
  KBN_DEBUG=$dbg timeout 200 rocprofv3 --pmc $set --output-format csv -d $O -o d${dbg}_$tag -- python $R/tools/conv_bench.py --one $L > $O/d${dbg}_$tag.log 2>&1
 done
done
ls $O | wc -l
