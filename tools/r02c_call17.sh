#!/bin/bash
# s_setprio around the MFMA clusters (default) vs none (VITX_PP_SCHED=1), interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/c17; mkdir -p $out
T="python tools/time_fwd.py"
for r in 1 2 3; do for p in 4 1; do echo -n "PP_SCHED=$p bf16: "; VITX_PP_SCHED=$p $T 256 vit_base_patch16_224 bf16 60 2>&1 | grep -v amdgpu; done; done | tee $out/fwd.txt
for p in 4 1; do echo -n "PP_SCHED=$p ViT-L/384: "; VITX_PP_SCHED=$p $T 128 vit_large_patch16_384 bf16 20 2>&1 | grep -v amdgpu; done | tee -a $out/fwd.txt
