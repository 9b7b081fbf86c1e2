#!/bin/bash
# Package power while ONE kernel class loops (steady state): energy per launch = mean power x mean time.   bash tools/energy_by_kernel.sh
cd "$GRAFT_REPO_ROOT" || exit 1
smi() { rocm-smi --showpower --showclocks -t 2>&1 | grep -E "sclk|Power" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';' | sed 's/=*//g; s/Power Consumption ;//; s/Current Socket Graphics Package Power (W)/W/; s/sclk clock level: 1: //'; echo; }
run() {   # name, command
  echo "## $1"
  eval "$2" > /tmp/e.txt 2>&1 &
  pid=$!
  sleep ${3:-2.5}
  for i in 1 2 3; do smi; sleep 0.5; done
  wait $pid; grep -v "amdgpu.ids\|^#\|max abs\|bit-ident" /tmp/e.txt | tail -2
}
run "gemm qkv (50432x2304x768, pp)" "tools/gemm_lab.bin 30000 'qkv:pp:0:bf16'" 2.0
run "gemm fc1+gelu" "tools/gemm_lab.bin 20000 'fc1:pp:1:bf16'" 2.0
run "gemm fc2+resid" "tools/gemm_lab.bin 20000 'fc2:pp:2:bf16'" 2.0
run "gemm proj+resid" "tools/gemm_lab.bin 50000 'proj:pp:2:bf16'" 2.0
run "gemm qkv, no epilogue" "tools/gemm_lab.bin 30000 'qkv:pp_noepi:0:bf16'" 2.0
run "gemm sq8k" "tools/gemm_lab.bin 5000 'sq8k:pp:0:bf16'" 2.0
run "attention single-pass (kernel 1) 256x12x197" "ATTN_ITERS=60000 python tools/attn_bench.py 256 197 12 1 bf16" 6.0
run "attention persistent (kernel 4) 256x12x197" "ATTN_ITERS=60000 python tools/attn_bench.py 256 197 12 4 bf16" 6.0
