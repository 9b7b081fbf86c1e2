#!/bin/bash
# Round-end measurement batch (run on the GPU box through gpurun):  bash tools/measure_round.sh <tag>
# default bench line (carries parity / sustained / f16 mode / configs 3 and 5), rocprofv3 kernel trace + two PMC passes of the same forward
# with the context's profiling schedule (sub-batches back to back on one stream: "one launch" = what bench.py's profiled step times).
tag=${1:-rXX}
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$tag; mkdir -p $out
python bench.py > $out/bench_default.json 2> $out/bench_default.err          # exactly what the driver runs
cp bench_extras.json $out/bench_default_extras.json 2>/dev/null              # the full record behind the compact line
# (r06: ONE bench run per round -- the default line carries the F16 mode, the q4_0 file and ViT-L/384 as `parity_mode` / `other_configs`; the full record is bench_default_extras.json)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$out/prof -o fwd -- python $R/tools/prof_forward.py vit_base_patch16_224 256 5 bf16 profile=1,last_layer_all_rows=1 > $R/$out/prof.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum -d $R/$out/pmc1 -o fwd -- python $R/tools/prof_forward.py vit_base_patch16_224 256 2 bf16 profile=1,last_layer_all_rows=1 > $R/$out/pmc1.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -d $R/$out/pmc2 -o fwd -- python $R/tools/prof_forward.py vit_base_patch16_224 256 2 bf16 profile=1,last_layer_all_rows=1 > $R/$out/pmc2.log 2>&1 )
python tools/rocpd_summary.py $(find $out/prof $out/pmc1 $out/pmc2 -name "*.db" | sort) > $out/rocprofv3_summary.txt 2>&1
python tools/hbm_traffic.py $(find $out/pmc1 -name "*.db" | head -1) $(find $out/pmc2 -name "*.db" | head -1) --commit "${COMMIT:-unknown}" --out $out/hbm_traffic.json > $out/hbm_traffic.log 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$out/profL -o fwd -- python $R/tools/prof_forward.py vit_large_patch16_384 128 3 bf16 profile=1,last_layer_all_rows=1 > $R/$out/profL.log 2>&1 )
python tools/rocpd_summary.py $(find $out/profL -name "*.db" | sort) > $out/rocprofv3_summary_large384.txt 2>&1
# the bench line against a kernel trace of THE SAME COMMAND (does the calibrated event clock agree with the device's dispatch stamps?)
( cd /tmp && timeout 400 rocprofv3 --kernel-trace -d $R/$out/bt -o b -- python $R/bench.py --no-extras --no-cpu-baseline > $R/$out/bench_same_run_as_rocprof.json 2> $R/$out/bt.log )
python tools/bench_trace_agreement.py $(find $out/bt -name "*.db" | head -1) $out/bench_same_run_as_rocprof.json > $out/bench_under_rocprofv3.txt 2>&1
# small-batch latency table: f16 file and q4_0 file (blocks in HBM), bf16 compute
for ft in f16 q4_0; do for b in 1 8 32 64; do TF_FTYPE=$ft python tools/time_fwd.py $b vit_base_patch16_224 bf16 100 2>&1 | grep -v amdgpu; done; done > $out/small_batches.txt
find $out -name "*.db" -size +20M -delete
python - <<PY
import json
d = json.loads(open("$out/bench_default.json").read().strip().splitlines()[-1])
print(len(json.dumps(d)), "bytes;", d["config"]["workload"][:60], d["value"], d["ms_per_step"], d.get("roofline", {}).get("frac"), "parity_mode", d.get("parity_mode", {}).get("value"))
for k, v in d.get("other_configs", {}).items(): print("  ", k, v.get("value"), v.get("frac"), (v.get("parity") or {}).get("passed"))
PY
head -14 $out/rocprofv3_summary.txt; cat $out/hbm_traffic.json | head -14; cat $out/bench_under_rocprofv3.txt
