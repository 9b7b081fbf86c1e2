#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/c7; mkdir -p $out
( timeout 300 python -m pytest tests/test_gpu_parity_r02.py -x -q -k "persistent_attention" > $out/pytest_attn.txt 2>&1 ); grep -E "passed|failed|Error|assert" $out/pytest_attn.txt | tail -6
for k in 1 4 3; do python tools/attn_bench.py 128 197 12 $k bf16 2>&1 | grep -v amdgpu | head -3; done | tee $out/attn_bench.txt
for k in 1 4; do python tools/attn_bench.py 256 197 12 $k f16 2>&1 | grep -v amdgpu | head -1; done | tee -a $out/attn_bench.txt
T="python tools/time_fwd.py"
for r in 1 2; do for p in 0 1; do echo -n "ATTN_PERSIST=$p: "; VITX_ATTN_PERSIST=$p $T 256 vit_base_patch16_224 bf16 60 2>&1 | grep -v amdgpu; done; done | tee $out/fwd.txt
