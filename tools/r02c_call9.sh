#!/bin/bash
# all GEMM families on v_mfma_16x16x32: lab refchecks + A/B against the 32x32x16 ping-pong kernel, then the GPU test suite
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/c9; mkdir -p $out
timeout 300 tools/gemm_lab.bin 10 ':pp:,:pp32:,:pp2:,:ring945:' > $out/lab.txt 2>&1; grep -c " ok " $out/lab.txt; grep -v " ok " $out/lab.txt | head -20
grep -E "^(sq8k|qkv|proj|fc1|fc2|qkvL|fc2L)" $out/lab.txt | grep bf16
( timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1 ); tail -15 $out/pytest_gpu.txt
