#!/bin/bash
# SQ / LDS / L2 counters of one attention micro-benchmark:  bash tools/pmc_attn.sh <tag> "<attn_bench args>"     (run through gpurun)
tag=${1:-pmca}; args=${2:-64 577 16 3 bf16}
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS" \
           "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "WRITE_SIZE SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_WAVES"; do
  i=$((i+1))
  ( cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc $set -d $R/$out/p$i -o g -- python $R/tools/attn_bench.py $args > $R/$out/p$i.log 2>&1 )
done
python tools/rocpd_summary.py $(find $out -name "*.db" | sort) 2>&1 | grep -E "^##|^kernel|attention" > $out/summary.txt
find $out -name "*.db" -delete
cat $out/summary.txt
