#!/bin/bash
# rocm-smi power / clock samples while the forward loops (batch 256 ViT-B bf16):  bash tools/smi_forward.sh [iters] [extra env...]
cd "$GRAFT_REPO_ROOT" || exit 1
iters=${1:-1200}
smi() { rocm-smi --showpower --showclocks -t 2>&1 | grep -E "sclk|Power|Temperature \(Sensor junction" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';' | sed 's/=*//g'; echo; }
python tools/time_fwd.py 256 vit_base_patch16_224 bf16 $iters > /tmp/fw.txt 2>&1 &
pid=$!
for i in $(seq 1 22); do sleep 0.7; smi; done
wait $pid; grep -v amdgpu.ids /tmp/fw.txt
