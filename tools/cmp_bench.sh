cd $GRAFT_REPO_ROOT
B="python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-host-feed --no-profile"
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["ms_per_step"])'
$B | python -c "$P" bench
VITX_FLAT_PRIORITY=1 $B | python -c "$P" bench_flat
GPU_MAX_HW_QUEUES=8 VITX_FLAT_PRIORITY=1 $B | python -c "$P" bench_flat_q8
GPU_MAX_HW_QUEUES=8 $B | python -c "$P" bench_q8
python tools/time_fwd.py 256 vit_base_patch16_224 bf16 30
VITX_FLAT_PRIORITY=1 python tools/time_fwd.py 256 vit_base_patch16_224 bf16 30
TF_STREAM=1 python tools/time_fwd.py 256 vit_base_patch16_224 bf16 30
python tools/time_fwd.py 128 vit_large_patch16_384 bf16 10
VITX_FLAT_PRIORITY=1 python tools/time_fwd.py 128 vit_large_patch16_384 bf16 10
