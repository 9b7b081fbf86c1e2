#!/bin/bash
# tail split (rows beyond whole rounds re-tiled 128x256 in a second launch) re-tested with the 16x16x32 kernels
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/c18; mkdir -p $out
T="python tools/time_fwd.py"
for r in 1 2 3; do for p in 0 1; do echo -n "GEMM_SPLIT=$p: "; VITX_GEMM_SPLIT=$p $T 256 vit_base_patch16_224 bf16 60 2>&1 | grep -v amdgpu; done; done | tee $out/fwd.txt
for p in 0 1; do echo -n "GEMM_SPLIT=$p ViT-L/384: "; VITX_GEMM_SPLIT=$p $T 128 vit_large_patch16_384 bf16 20 2>&1 | grep -v amdgpu; done | tee -a $out/fwd.txt
