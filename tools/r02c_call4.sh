#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/c4; mkdir -p $out
for gm in 2 4 8 12 16 32; do
  echo "## GROUP_M $gm"; LAB_GROUP_M=$gm tools/gemm_lab.bin 20 'qkv:pp:0:bf16,fc1:pp:1:bf16,fc2:pp:2:bf16,proj:pp:2:bf16,sq8k:pp:0:bf16' | grep -v "^#"
done > $out/group_m_lab.txt 2>&1
cat $out/group_m_lab.txt
T="python tools/time_fwd.py"
for gm in 4 8 16 32; do echo "GROUP_M $gm"; VITX_GROUP_M=$gm $T 256 vit_base_patch16_224 bf16 40 | grep -v amdgpu; done 2>&1 | tee $out/group_m_fwd.txt
