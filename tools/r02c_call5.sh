#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/c5; mkdir -p $out
( timeout 600 python -m pytest tests/test_gpu_quant.py "tests/test_gpu_e2e.py::test_quantised_file_runs_dequantised" -x -q > $out/pytest_quant.txt 2>&1 ); grep -E "passed|failed|Error|error" $out/pytest_quant.txt | tail -5
T="python tools/time_fwd.py"
{
  for b in 1 8 256; do
    echo "== batch $b: f16 file / q4_0 expanded on the host / q4_0 blocks, expansion one layer ahead / inline"
    $T $b vit_base_patch16_224 bf16 40
    TF_FTYPE=q4_0 VITX_QUANT_HOST=1 $T $b vit_base_patch16_224 bf16 40
    TF_FTYPE=q4_0 $T $b vit_base_patch16_224 bf16 40
    TF_FTYPE=q4_0 VITX_QUANT_PREFETCH=0 $T $b vit_base_patch16_224 bf16 40
  done
} 2>&1 | grep -v amdgpu.ids | tee $out/quant_paths.txt
