#!/bin/bash
# MFMA issue order inside a quadrant: W fragment held x4 (u outer) vs A fragment held x2 (t outer)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/c12; mkdir -p $out
T="python tools/time_fwd.py"
for r in 1 2 3; do for lib in tools/ab/libvitx_theld.so tools/ab/libvitx_uheld.so; do echo -n "VITX_LIB=$lib: "; VITX_LIB=$lib $T 256 vit_base_patch16_224 bf16 60 2>&1 | grep -v amdgpu; done; done | tee $out/fwd.txt
for lib in tools/ab/libvitx_theld.so tools/ab/libvitx_uheld.so; do echo -n "VITX_LIB=$lib: "; VITX_LIB=$lib $T 128 vit_large_patch16_384 bf16 20 2>&1 | grep -v amdgpu; done | tee -a $out/fwd.txt
