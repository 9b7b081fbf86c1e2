#!/bin/bash
# r02c GPU batch 1: two-burst schedule A/B (lab + forward), quantised-weight device paths, full GPU test suite.
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/c1; mkdir -p $out
( timeout 240 tools/gemm_lab.bin 10 ':pp:,:pp2:' > $out/lab_pp_vs_pp2.txt 2>&1 )
( timeout 60 tools/gemm_lab.bin 5 'sq4k:pp2_stamp,sq4k:pp2_mfmaonly_stamp,sq4k:pp_stamp,qkv:pp2_noepi,qkv:pp_noepi' > $out/lab_stamps.txt 2>&1 )
( timeout 600 python -m pytest tests/test_gpu_quant.py -x -q > $out/pytest_quant.txt 2>&1 ); tail -3 $out/pytest_quant.txt
T="python tools/time_fwd.py"
{
  echo "== ViT-B bs256 bf16: four-phase vs two-burst (interleaved, 2 rounds)"
  for r in 1 2; do
    VITX_PP_SCHED=4 $T 256 vit_base_patch16_224 bf16 30
    VITX_PP_SCHED=2 $T 256 vit_base_patch16_224 bf16 30
  done
  echo "== q4_0 file, bs256: host-expanded vs blocks in HBM + JIT expansion"
  TF_FTYPE=q4_0 VITX_QUANT_HOST=1 $T 256 vit_base_patch16_224 bf16 30
  TF_FTYPE=q4_0 $T 256 vit_base_patch16_224 bf16 30
  echo "== batch 1 latency: f16 file / q4_0 host-expanded / q4_0 fused GEMM / q4_0 JIT"
  $T 1 vit_base_patch16_224 f16 50
  TF_FTYPE=q4_0 VITX_QUANT_HOST=1 $T 1 vit_base_patch16_224 f16 50
  TF_FTYPE=q4_0 $T 1 vit_base_patch16_224 f16 50
  TF_FTYPE=q4_0 VITX_Q4_FUSED_ROWS=0 $T 1 vit_base_patch16_224 f16 50
  echo "== batch 8: f16 file / q4_0 fused / q4_0 JIT"
  $T 8 vit_base_patch16_224 f16 50
  TF_FTYPE=q4_0 $T 8 vit_base_patch16_224 f16 50
  TF_FTYPE=q4_0 VITX_Q4_FUSED_ROWS=0 $T 8 vit_base_patch16_224 f16 50
} > $out/time_fwd.txt 2>&1
cat $out/time_fwd.txt | grep -v "^$" | tail -30
( timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1 ); tail -5 $out/pytest_gpu.txt
grep -E "sq8k|qkv |fc2 |fc1 |proj " $out/lab_pp_vs_pp2.txt | grep bf16 | head -30
