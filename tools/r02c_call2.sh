#!/bin/bash
# r02c GPU batch 2: matrix-pipe ceiling under this chip's power management (+ rocm-smi samples), forward schedule variants.
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/c2; mkdir -p $out
smi() { rocm-smi --showpower --showclocks -t 2>&1 | grep -E "sclk|Power|Temperature \(Sensor (junction|edge)" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';'; echo; }
{
  echo "## idle"; smi
  rocm-smi --showmaxpower --showperflevel 2>&1 | grep -v "^=\|^$" | head -8
  echo "## tools/mfma_ceiling.bin 300"
  tools/mfma_ceiling.bin 300
  echo "## rocm-smi while mfma_ceiling runs 1.5 s cases (zero, const, random bf16, random f16, sleep 1/4/8)"
  tools/mfma_ceiling.bin 1500 > /tmp/mc.txt 2>&1 &
  pid=$!
  for i in $(seq 1 14); do sleep 0.75; smi; done
  wait $pid; cat /tmp/mc.txt
  echo "## rocm-smi while the GEMM lab loops sq8k pp (3000 launches)"
  tools/gemm_lab.bin 3000 'sq8k:pp:0:bf16' > /tmp/lab.txt 2>&1 &
  pid=$!
  for i in 1 2 3 4; do sleep 0.6; smi; done
  wait $pid; cat /tmp/lab.txt
} > $out/mfma_ceiling.txt 2>&1
T="python tools/time_fwd.py"
{
  echo "== streams 1 / 2 / 3"
  VITX_STREAMS=1 $T 256 vit_base_patch16_224 bf16 30
  VITX_STREAMS=2 $T 256 vit_base_patch16_224 bf16 30
  VITX_STREAMS=3 $T 256 vit_base_patch16_224 bf16 30
  echo "== balance off / split on"
  VITX_GEMM_BALANCE=0 $T 256 vit_base_patch16_224 bf16 30
  VITX_GEMM_SPLIT=1 $T 256 vit_base_patch16_224 bf16 30
  echo "== rocm-smi during 400 forwards"
  $T 256 vit_base_patch16_224 bf16 400 > /tmp/fw.txt 2>&1 &
  pid=$!
  sleep 6
  for i in 1 2 3 4 5; do sleep 0.5; smi; done
  wait $pid; cat /tmp/fw.txt
} > $out/forward_variants.txt 2>&1
cat $out/mfma_ceiling.txt | head -60; grep -v amdgpu.ids $out/forward_variants.txt
