#!/bin/bash
# re-tune the launch parameters after the GEMMs moved to 16x16x32 (the energy balance between kernels shifted)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/c13; mkdir -p $out
T="python tools/time_fwd.py 256 vit_base_patch16_224 bf16 60"
run() { echo -n "$1: "; env $1 $T 2>&1 | grep -v amdgpu | sed 's/vit_base_patch16_224 f16-file batch 256: //'; }
for cfg in X=0 VITX_SPLIT=96 VITX_SPLIT=104 VITX_SPLIT=118 VITX_SPLIT=128 X=0 VITX_GROUP_M=4 VITX_GROUP_M=16 VITX_LN_FUSE=1 X=0 VITX_GEMM_BALANCE=0 VITX_STREAMS=3 VITX_STREAMS=1 VITX_ATTN_PERSIST=0 X=0; do run $cfg; done | tee $out/sweep.txt
