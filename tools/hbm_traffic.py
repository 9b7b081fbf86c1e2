"""HBM-side traffic per launch of every kernel class of the forward, from two rocprofv3 PMC passes (rocpd SQLite).

usage: python tools/hbm_traffic.py <pmc_fetch.db> <pmc_write.db> [--commit SHA] [--out profiles/hbm_traffic.json]

pass 1 collected FETCH_SIZE, pass 2 WRITE_SIZE (separate passes, as MI355X_MICROARCH.md prescribes); both count 1 kB units and
FETCH_SIZE under-reports by 2x on gfx950, so   HBM bytes per dispatch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024.
Kernel symbols are mapped onto bench.py's class names (the engine's kProfNames, csrc/engine.cpp): the GEMM symbol carries the
epilogue (0 qkv bias, 1 fc1 GELU, 2 residual = proj AND fc2 -- with or without the fused LayerNorm --, 3 head); proj and fc2 alternate
in launch order inside a layer (proj first), which is how the residual-epilogue dispatches are split here.  bench.py reads the JSON this
writes.  Run the forward with the context's profiling switch on (tools/prof_forward.py ... profile=1): the two sub-batches then run back to
back exactly as in bench.py's profiled step, so "one launch" means the same thing in both.
"""
import argparse, json, re, sqlite3, subprocess, sys
from collections import defaultdict

EPI_CLASS = {0: "gemm_qkv_bias", 1: "gemm_fc1_gelu", 3: "gemm_head"}


def classify(name):
    """kernel symbol -> (class or None, epilogue or None)."""
    m = re.search(r"gemm_\w+_kernelIDF16[b_]Li(\d+)E", name)
    if m:
        return "gemm", int(m.group(1))
    if re.search(r"gemm_\w+_kernel<bool _Accum, int, E,", name):
        return "gemm", 1          # rocprofv3's demangler garbles <bf16, 1, ...> (DF16b Li1E) into this; every other instantiation stays mangled
    for key, cls in (("attention", "attention"), ("layernorm", "layernorm"), ("patch_embed", "patch_embed"), ("softmax", "softmax")):
        if key in name and "vitx" in name:
            return cls, None
    return None, None


def per_dispatch(db, counter):
    """[(start, kernel name, counter value)] sorted by start time."""
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(pmc_events)")]
    key = next((c for c in ("start", "dispatch_id", "event_id", "id") if c in cols), None)
    if key is None:
        raise SystemExit(f"{db}: pmc_events has no column to order dispatches by ({cols})")
    return con.execute(f"select {key}, name, counter_value from pmc_events where counter_name = ? order by {key}", (counter,)).fetchall()


def by_class(rows):
    out = defaultdict(list)
    resid = 0
    for _, name, v in rows:
        cls, epi = classify(name)
        if cls is None:
            continue
        if cls == "gemm":
            if epi == 2:
                cls = "gemm_proj_resid" if resid % 2 == 0 else "gemm_fc2_resid"
                resid += 1
            else:
                cls = EPI_CLASS.get(epi)
                if cls is None:
                    continue
        out[cls].append(v)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_db"); ap.add_argument("write_db")
    ap.add_argument("--commit", default=None)
    ap.add_argument("--out", default="profiles/hbm_traffic.json")
    ap.add_argument("--what", default="tools/prof_forward.py vit_base_patch16_224 256 2 bf16 profile=1")
    a = ap.parse_args()
    f = by_class(per_dispatch(a.fetch_db, "FETCH_SIZE"))
    w = by_class(per_dispatch(a.write_db, "WRITE_SIZE"))
    commit = a.commit
    if commit is None:
        try:
            commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()
        except Exception:
            commit = "unknown"
    gb, detail = {}, {}
    for cls in sorted(set(f) & set(w)):
        fm = sum(f[cls]) / len(f[cls]); wm = sum(w[cls]) / len(w[cls])
        gb[cls] = round((2 * fm + wm) * 1024 / 1e9, 4)
        detail[cls] = {"dispatches": len(f[cls]), "FETCH_SIZE_mean": round(fm, 1), "WRITE_SIZE_mean": round(wm, 1)}
    doc = {"gb_per_launch": gb, "detail": detail, "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 B, mean over dispatches of the class; one launch = one sub-batch (103 or 153 of the 256 images)",
           "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two passes) of {a.what}", "commit": commit}
    with open(a.out, "w") as fo:
        json.dump(doc, fo, indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
