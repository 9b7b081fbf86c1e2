# forward wall-clock over sub-batch splits (ViT-B/16 bs256 bf16): bash tools/sweep_split.sh
cd $GRAFT_REPO_ROOT
run() { echo -n "$1 : "; env $1 python tools/time_fwd.py 256 vit_base_patch16_224 bf16 40 2>&1 | grep -v amdgpu | sed 's/.*: //'; }
run "VITX_GEMM_BALANCE=0"
run "VITX_GEMM_BALANCE=1"
run "VITX_STREAMS=1 VITX_GEMM_BALANCE=0"
run "VITX_STREAMS=1"
for s in 96 104 110 116 120 128 136 146; do run "VITX_STREAMS=2 VITX_SPLIT=$s"; done
run "VITX_STREAMS=3 VITX_SPLIT=85,85"
