#!/bin/bash
# forward A/B of the 16x16x32-MFMA GEMM (VITX_PP_SCHED=16) against the 32x32x16 default, interleaved; ViT-B bs256 and ViT-L/384 bs128
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/c8; mkdir -p $out
T="python tools/time_fwd.py"
for r in 1 2 3; do for p in 4 16; do echo -n "PP_SCHED=$p bf16: "; VITX_PP_SCHED=$p $T 256 vit_base_patch16_224 bf16 60 2>&1 | grep -v amdgpu; done; done | tee $out/fwd.txt
for p in 4 16; do echo -n "PP_SCHED=$p f16: "; VITX_PP_SCHED=$p $T 256 vit_base_patch16_224 f16 60 2>&1 | grep -v amdgpu; done | tee -a $out/fwd.txt
for p in 4 16; do echo -n "PP_SCHED=$p ViT-L/384 bf16: "; VITX_PP_SCHED=$p $T 128 vit_large_patch16_384 bf16 20 2>&1 | grep -v amdgpu; done | tee -a $out/fwd.txt
