#!/bin/bash
# after the stream kernel got the staged epilogue: full GPU suite; two-burst vs four-phase schedule on 16x16x32; ViT-tiny / small (stream + ring kernels) vs the r02e build
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/c11; mkdir -p $out
( timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1 ); grep -E "passed|failed" $out/pytest_gpu.txt
tools/mfma_ceiling.bin 200 > $out/mfma_ceiling.txt 2>&1; cat $out/mfma_ceiling.txt
T="python tools/time_fwd.py"
for r in 1 2; do for p in 4 2; do echo -n "PP_SCHED=$p bf16: "; VITX_PP_SCHED=$p $T 256 vit_base_patch16_224 bf16 60 2>&1 | grep -v amdgpu; done; done | tee $out/fwd_sched.txt
for m in vit_tiny_patch16_224 vit_small_patch16_224; do for lib in "" tools/ab/libvitx_r02e.so; do echo -n "VITX_LIB=$lib: "; VITX_LIB=$lib $T 256 $m bf16 100 2>&1 | grep -v amdgpu; done; done | tee $out/fwd_tiny_small.txt
VITX_LIB= python tools/gemm_families.py bf16 50 2>&1 | grep -v amdgpu | tee $out/families.txt
