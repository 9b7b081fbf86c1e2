#!/bin/bash
# ring / v1 kernels after the move to 16x16x32 + LDS-staged epilogue, against the r02e build (VITX_LIB=tools/ab/libvitx_r02e.so)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/c10; mkdir -p $out
( timeout 900 python -m pytest tests/test_gpu_parity_r02.py tests/test_gpu_kernels.py tests/test_gpu_quant.py -x -q -m gpu > $out/pytest.txt 2>&1 ); tail -4 $out/pytest.txt
for lib in "" tools/ab/libvitx_r02e.so; do echo "== VITX_LIB=$lib"; VITX_LIB=$lib python tools/gemm_families.py bf16 50 2>&1 | grep -v amdgpu; done | tee $out/families.txt
T="python tools/time_fwd.py"
for b in 1 8 32 64; do for lib in "" tools/ab/libvitx_r02e.so; do echo -n "batch $b VITX_LIB=$lib: "; VITX_LIB=$lib $T $b vit_base_patch16_224 bf16 100 2>&1 | grep -v amdgpu; done; done | tee $out/fwd_small.txt
