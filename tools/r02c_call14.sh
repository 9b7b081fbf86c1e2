#!/bin/bash
# hipGraph cache of the single-stream forward: correctness + latency with and without (VITX_GRAPH=0)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/c14; mkdir -p $out
( timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu > $out/pytest.txt 2>&1 ); tail -5 $out/pytest.txt
T="python tools/time_fwd.py"
for ft in f16 q4_0; do for b in 1 4 8 15; do for g in 0 1; do echo -n "ftype $ft VITX_GRAPH=$g: "; TF_FTYPE=$ft VITX_GRAPH=$g $T $b vit_base_patch16_224 bf16 200 2>&1 | grep -v amdgpu; done; done; done | tee $out/graph_latency.txt
for m in vit_tiny_patch16_224 vit_large_patch16_384; do for g in 0 1; do echo -n "VITX_GRAPH=$g: "; VITX_GRAPH=$g $T 1 $m bf16 100 2>&1 | grep -v amdgpu; done; done | tee -a $out/graph_latency.txt
