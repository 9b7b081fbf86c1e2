#!/bin/bash
# Samples clocks / power with rocm-smi while a lab case loops:  bash tools/smi_during.sh <lab filter> [iters]
cd "$GRAFT_REPO_ROOT" || exit 1
filt=${1:-sq8k:pp:0:bf16}; iters=${2:-3000}
rocm-smi --showpower --showclocks --showperflevel --showmaxpower 2>&1 | grep -v "^=\|^$" | head -30
tools/gemm_lab.bin $iters $filt > /tmp/lab_bg.txt 2>&1 &
pid=$!
sleep 1.0
for i in 1 2 3 4 5; do
  rocm-smi --showpower --showclocks -t 2>&1 | grep -E "sclk|Power|fclk|mclk|Temperature \(Sensor (junction|edge)" | tr '\n' ' '; echo
  sleep 0.4
done
wait $pid
cat /tmp/lab_bg.txt
