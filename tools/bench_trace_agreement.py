"""Does the bench line agree with a kernel trace of THE SAME COMMAND?
    cd /tmp && rocprofv3 --kernel-trace -d DIR -o b -- python bench.py --no-extras --no-cpu-baseline > line.json
    python tools/bench_trace_agreement.py DIR/.../b_results.db line.json
bench.py times its steps on the production schedule (two sub-batches on two streams) and then runs ONE more forward with the context's profiling
schedule (sub-batches back to back on one stream, every launch bracketed by HIP events): `roofline` and `kernel_breakdown` come from that forward.
In the trace it is the LAST forward (all of its launches on one hardware queue).  This prints, per kernel class, the mean duration rocprofv3 stamped
for that forward's launches next to the mean the HIP events gave, and the same for the launches of the timed steps (co-running: longer)."""
import json, re, sqlite3, sys
from collections import defaultdict
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from timeline import classify

db, line = sys.argv[1], sys.argv[2]
d = json.loads(open(line).read().strip().splitlines()[-1])
if "kernel_breakdown" not in d:      # r06: the last stdout line is the compact one; the per-kernel table is in the full record the same run wrote next to bench.py
    import os
    full = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), d.get("extras", "bench_extras.json"))
    d = json.load(open(full))
rows = sqlite3.connect(db).execute("select name, queue_id, start, end from kernels order by start").fetchall()
ks, resid_n = [], defaultdict(int)
for name, q, s, e in rows:
    c = classify(name)
    if c is None: continue
    if c == "resid":
        c = "proj" if resid_n[q] % 2 == 0 else "fc2"; resid_n[q] += 1
    ks.append((c, q, s, e))
patches = [i for i, k in enumerate(ks) if k[0] == "patch"]
# forwards = pairs of patch-embedding launches (one per sub-batch); a forward's kernels run from its first patch launch to the next forward's
starts = patches[0::2] + [len(ks)]
fwd = [ks[starts[i]:starts[i + 1]] for i in range(len(starts) - 1)]
n_per = max(set(len(f) for f in fwd), key=[len(f) for f in fwd].count)
single = [i for i, f in enumerate(fwd) if len(set(k[1] for k in f[:n_per])) == 1]      # the profiled forward: every launch on ONE hardware queue
assert single, "no single-queue (profiled) forward in the trace"
prof = fwd[single[-1]][:n_per]
timed = [k for i, f in enumerate(fwd) if i not in single and i >= 5 for k in f[:n_per]]      # (skip the warm-up steps)
names = {"qkv": "gemm_qkv_bias", "proj": "gemm_proj_resid", "fc1": "gemm_fc1_gelu", "fc2": "gemm_fc2_resid", "attention": "attention", "patch": "patch_embed", "layernorm": "layernorm"}
def mean(xs): return sum(xs) / max(1, len(xs))
print(f"bench line: value {d['value']} images/s, {d['ms_per_step']} ms/step; roofline kernel {d['roofline']['kernel']}: avg_launch_ms {d['roofline']['avg_launch_ms']}")
print(f"{'class':10s} {'launches':>8s} {'rocprofv3 us (profiled fwd)':>28s} {'HIP events us (bench line)':>28s} {'ratio':>7s} {'rocprofv3 us (timed steps, two streams)':>40s}")
for c, bn in names.items():
    p = [(e - s) / 1e3 for cc, q, s, e in prof if cc == c]
    t = [(e - s) / 1e3 for cc, q, s, e in timed if cc == c]
    kb = d["kernel_breakdown"].get(bn)
    if not p or not kb: continue
    ev = kb["busy_ms_per_step"] * 1e3 / kb["launches"]
    print(f"{c:10s} {len(p):8d} {mean(p):28.1f} {ev:28.1f} {ev / mean(p):7.3f} {mean(t):40.1f}")
