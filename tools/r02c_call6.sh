#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/c6; mkdir -p $out
( timeout 600 python -m pytest tests/test_gpu_parity_r02.py -x -q -k "layernorm_fused" > $out/pytest_ln.txt 2>&1 ); grep -E "passed|failed|Error|assert" $out/pytest_ln.txt | tail -8
T="python tools/time_fwd.py"
for r in 1 2 3; do
  for f in 0 1; do echo -n "LN_FUSE=$f: "; VITX_LN_FUSE=$f $T 256 vit_base_patch16_224 bf16 60 2>&1 | grep -v amdgpu; done
done | tee $out/ln_fuse_fwd.txt
for f in 0 1; do echo -n "L LN_FUSE=$f: "; VITX_LN_FUSE=$f $T 128 vit_large_patch16_384 bf16 15 2>&1 | grep -v amdgpu; done | tee -a $out/ln_fuse_fwd.txt
