#!/bin/bash
# deep-ring skinny GEMM (cfg 129): parity + families timing + small-batch forwards with and without (VITX_DEEP_RING=0)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/c15; mkdir -p $out
( timeout 600 python -m pytest tests/test_gpu_parity_r02.py -x -q -m gpu -k "gemm" > $out/pytest.txt 2>&1 ); tail -3 $out/pytest.txt
python tools/gemm_families.py bf16 100 2>&1 | grep -v amdgpu | tee $out/families.txt
T="python tools/time_fwd.py"
for b in 1 2 4 8; do for g in 0 1; do echo -n "VITX_DEEP_RING=$g: "; VITX_DEEP_RING=$g $T $b vit_base_patch16_224 bf16 200 2>&1 | grep -v amdgpu; done; done | tee $out/fwd.txt
for m in vit_tiny_patch16_224 vit_large_patch16_384; do for g in 0 1; do echo -n "VITX_DEEP_RING=$g: "; VITX_DEEP_RING=$g $T 1 $m bf16 100 2>&1 | grep -v amdgpu; done; done | tee -a $out/fwd.txt
