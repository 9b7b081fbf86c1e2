#!/bin/bash
# SQ / LDS / TA counters of one lab case:  bash tools/pmc_lab.sh <tag> <lab filter>     (run through gpurun)
tag=${1:-pmc}; filt=${2:-sq8k:pp:0:bf16}
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" \
           "TA_BUSY_avr SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VALU"; do
  i=$((i+1))
  ( cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc $set -d $R/$out/p$i -o g -- $R/tools/gemm_lab.bin 3 $filt > $R/$out/p$i.log 2>&1 )
done
python tools/rocpd_summary.py $(find $out -name "*.db" | sort) 2>&1 | grep -E "^##|^kernel|gemm_pp|gemm_stream|gemm_ring" > $out/summary.txt
find $out -name "*.db" -delete
cat $out/summary.txt
