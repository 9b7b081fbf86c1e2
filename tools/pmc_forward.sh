#!/bin/bash
# Matrix-pipe and VALU utilisation per kernel of the forward (profiling schedule: sub-batches back to back on one stream):
#   bash tools/pmc_forward.sh <tag> [model] [batch]      (run through gpurun; counters in their own passes, --kernel-trace only)
tag=${1:-pmcf}; model=${2:-vit_base_patch16_224}; batch=${3:-256}
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $set -d $R/$out/p$i -o f -- python $R/tools/prof_forward.py $model $batch 2 bf16 profile=1 > $R/$out/p$i.log 2>&1 )
done
python tools/rocpd_summary.py $(find $out -name "*.db" | sort) 2>&1 | grep -E "^##|^kernel|vitx" > $out/summary.txt
find $out -name "*.db" -delete
cat $out/summary.txt
