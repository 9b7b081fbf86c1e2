#!/bin/bash
# skinny-tile threshold: fewer than 64 (r02e) / 128 / 192 / 256 tiles of 128x256 -> 64x128 tiles
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/c16; mkdir -p $out
T="python tools/time_fwd.py"
for b in 2 4 8 12 16 24 32; do for g in 64 128 192 256; do echo -n "batch $b VITX_SKINNY_TILES=$g: "; VITX_SKINNY_TILES=$g $T $b vit_base_patch16_224 bf16 100 2>&1 | grep -v amdgpu | sed 's/vit_base_patch16_224 f16-file //'; done; done | tee $out/fwd.txt
