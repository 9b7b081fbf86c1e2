#!/bin/bash
# SQ / LDS / TA counters of the GEMM kernels on one shape:  bash tools/pmc_gemm.sh <tag> [shape]   (run through gpurun)
tag=${1:-pmc}; shape=${2:-sq8k}
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" \
           "TA_BUSY_avr SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $set -d $R/$out/p$i -o g -- python $R/tools/gemm_bench.py --shapes $shape --iters 3 --dtype bf16 > $R/$out/p$i.log 2>&1 )
done
python tools/rocpd_summary.py $(find $out -name "*.db" | sort) 2>&1 | grep -E "^##|^kernel|gemm_stream|gemm_ring" > $out/summary.txt
find $out -name "*.db" -delete
cat $out/summary.txt
