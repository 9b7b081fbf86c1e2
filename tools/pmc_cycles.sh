#!/bin/bash
# Unperturbed cycles per kernel (GRBM_GUI_ACTIVE) and MFMA-busy cycles for a set of lab cases:  bash tools/pmc_cycles.sh <tag> <lab filter>
tag=${1:-cyc}; filt=${2:-sq8k:pp}
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $R/$out/p1 -o g -- $R/tools/gemm_lab.bin 3 $filt > $R/$out/p1.log 2>&1 )
python tools/rocpd_summary.py $(find $out -name "*.db" | sort) 2>&1 | grep -E "^##|^kernel|gemm_pp|gemm_stream|gemm_ring" > $out/summary.txt
find $out -name "*.db" -delete
cat $out/summary.txt
