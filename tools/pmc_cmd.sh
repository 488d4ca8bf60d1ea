#!/bin/bash
# usage: pmc_cmd.sh <outdir-under-gpurun_out> <command...> ; runs the command under the PMC sets of pmc_layer.sh
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; shift; mkdir -p $O
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d" " -f1)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O -o d0_$tag -- "$@" < /dev/null > $O/d0_$tag.log 2>&1
done
ls $O | wc -l
