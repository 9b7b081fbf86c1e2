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
