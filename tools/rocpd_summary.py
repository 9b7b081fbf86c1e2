"""Summarise rocprofv3 rocpd SQLite output: per-kernel time stats and per-kernel mean PMC values.
usage: python tools/rocpd_summary.py <results.db> [<results.db> ...]  (prints a text table)"""
import re, sqlite3, sys
from collections import defaultdict

def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_ZN4vitx(\d+)([A-Za-z_]+)", name)
    return name[:90]

def main(paths):
    for p in paths:
        con = sqlite3.connect(p)
        print(f"## {p}")
        rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
        tot = sum(r[2] for r in rows) or 1
        print(f"{'kernel':92s} {'calls':>6s} {'total_us':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
        for n, c, s, a, mn, mx in rows:
            print(f"{short(n):92s} {c:6d} {s/1e3:10.1f} {a/1e3:9.1f} {mn/1e3:9.1f} {mx/1e3:9.1f} {100*s/tot:6.2f}")
        # proj and fc2 run the SAME kernel symbol (residual epilogue, with or without the fused LayerNorm).  In launch order proj follows an
        # attention kernel and fc2 follows the GELU GEMM: split by predecessor, so that a per-class average can be read off next to bench.py's
        # roofline.avg_launch_ms (which is per class, from the engine's own HIP events)
        try:
            seq = con.execute("select name, duration from kernels order by start").fetchall()
        except sqlite3.Error:
            seq = []
        split = defaultdict(lambda: defaultdict(list))
        prev = ""
        for n, d in seq:
            if "gemm_pp_kernel" in n and "Li2E" in n:
                cls = "proj (launched after attention)" if "attention" in prev else "fc2 (launched after the GELU GEMM)" if "gemm_pp_kernel" in prev else "other"
                split[n][cls].append(d)
            if n.startswith("_ZN4vitx") or n.startswith("vitx::") or n.startswith("void vitx::"): prev = n
        for n, parts in split.items():
            for label, part in sorted(parts.items()):
                print(f"{('  ' + label + ': ' + short(n))[:92]:92s} {len(part):6d} {sum(part)/1e3:10.1f} {sum(part)/len(part)/1e3:9.1f} {min(part)/1e3:9.1f} {max(part)/1e3:9.1f}")
        try:
            pm = con.execute("select name, counter_name, avg(counter_value), count(*) from pmc_events group by name, counter_name").fetchall()
        except sqlite3.Error:
            pm = []
        if pm:
            d = defaultdict(dict)
            for n, cn, v, c in pm: d[n][cn] = v
            ctrs = sorted({cn for _, cn, _, _ in pm})
            print(f"\n{'kernel (mean counter value per dispatch)':92s} " + " ".join(f"{c:>22s}" for c in ctrs))
            for n in sorted(d, key=lambda k: -max(d[k].values())):
                print(f"{short(n):92s} " + " ".join(f"{d[n].get(c, float('nan')):22.1f}" for c in ctrs))
        print()

if __name__ == "__main__":
    main(sys.argv[1:])
