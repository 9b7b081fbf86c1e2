"""Timeline of the PRODUCTION (two-stream) forward from a rocprofv3 --kernel-trace run (rocpd SQLite): what runs beside what.

usage: python tools/timeline.py <results.db> [--skip N] [--forwards K]

rocprofv3 stamps every dispatch with its start / end on the device clock and the hardware queue it came through.  The engine runs
sub-batch 0 on the caller's stream and sub-batch 1 on an internal stream (engine.cpp), so the two queues of a forward are the two that
carry its patch_embed launches.  For each of the last K forwards (after skipping N warm-up ones) this prints
  * wall = first start -> last end, the sum of kernel durations per queue, and wall - max(sum per queue);
  * time with 0 / 1 / 2 kernels resident (union over both queues) and the queue gaps (next.start - prev.end inside a queue);
  * which kernel classes co-run: overlap time per pair (class of queue A, class of queue B);
  * a CU-demand estimate: sum over resident kernels of min(workgroups, 256) / 256, averaged over the wall time.
It reads no counter: run WITHOUT --pmc (PMC collection serialises dispatches).
"""
import argparse, re, sqlite3
from collections import defaultdict


def classify(name):
    m = re.search(r"gemm_\w+_kernelIDF16[b_]Li(\d+)E", name)
    if m:
        return {0: "qkv", 1: "fc1", 2: "resid", 3: "head", 4: "patch"}.get(int(m.group(1)), "gemm")
    if re.search(r"gemm_\w+_kernel<", name):
        return "fc1"          # rocprofv3's demangler garbles <bf16, 1, ...>
    for key in ("attention", "layernorm", "patch_embed", "softmax", "dequant", "topk", "spin"):
        if key in name:
            return key if key != "patch_embed" else "patch"
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--skip", type=int, default=2, help="forwards to skip at the start (warm-up)")
    ap.add_argument("--forwards", type=int, default=3)
    a = ap.parse_args()
    con = sqlite3.connect(a.db)
    rows = con.execute("select name, queue_id, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
    ks = []
    resid_n = defaultdict(int)
    for name, q, s, e, gx, wx in rows:
        c = classify(name)
        if c is None:
            continue
        if c == "resid":          # proj and fc2 share a symbol: they alternate inside a layer (proj first) on each queue
            c = "proj" if resid_n[q] % 2 == 0 else "fc2"
            resid_n[q] += 1
        ks.append(dict(cls=c, q=q, s=s, e=e, wgs=max(1, gx // max(1, wx))))
    # a forward = the kernels between one pair of patch launches (one per queue) and the next pair
    patches = [i for i, k in enumerate(ks) if k["cls"] == "patch"]
    if not patches:
        raise SystemExit("no patch_embed launches found")
    # group patch launches that start within 2 ms of each other: one forward
    groups, cur = [], [patches[0]]
    for i in patches[1:]:
        if ks[i]["s"] - ks[cur[0]]["s"] < 2_000_000:
            cur.append(i)
        else:
            groups.append(cur); cur = [i]
    groups.append(cur)
    bounds = [g[0] for g in groups] + [len(ks)]
    fwd = [ks[bounds[i]:bounds[i + 1]] for i in range(len(groups))]
    # resid parity restarts per forward
    for f in fwd:
        n = defaultdict(int)
        for k in f:
            if k["cls"] in ("proj", "fc2"):
                k["cls"] = "proj" if n[k["q"]] % 2 == 0 else "fc2"; n[k["q"]] += 1
    sel = fwd[a.skip:a.skip + a.forwards] if len(fwd) > a.skip else fwd[-a.forwards:]
    print(f"{len(fwd)} forwards in the trace; analysing {len(sel)} (skip {a.skip})")
    for fi, f in enumerate(sel):
        t0 = min(k["s"] for k in f); t1 = max(k["e"] for k in f)
        wall = (t1 - t0) / 1e3
        qs = sorted({k["q"] for k in f})
        print(f"\n== forward {fi}: wall {wall:.1f} us, {len(f)} launches on queues {qs}")
        per_q = {}
        for q in qs:
            kk = [k for k in f if k["q"] == q]
            busy = sum(k["e"] - k["s"] for k in kk) / 1e3
            gaps = [(kk[i + 1]["s"] - kk[i]["e"]) / 1e3 for i in range(len(kk) - 1)]
            per_q[q] = busy
            pos = [g for g in gaps if g > 0]
            print(f"  queue {q}: {len(kk)} launches, sum of durations {busy:.1f} us, span {(kk[-1]['e'] - kk[0]['s']) / 1e3:.1f} us, "
                  f"gaps: sum {sum(pos):.1f} us, median {sorted(pos)[len(pos) // 2] if pos else 0:.2f}, max {max(pos) if pos else 0:.1f}")
        print(f"  wall - max(sum per queue) = {wall - max(per_q.values()):.1f} us; sum over queues {sum(per_q.values()):.1f} us (x{sum(per_q.values()) / wall:.2f} of wall)")
        # sweep line
        ev = []
        for i, k in enumerate(f):
            ev.append((k["s"], 1, i)); ev.append((k["e"], 0, i))
        ev.sort()
        live = set(); last = t0
        resident = defaultdict(float); pair = defaultdict(float); solo = defaultdict(float); demand = 0.0
        for t, kind, i in ev:
            dt = (t - last) / 1e3
            if dt > 0:
                resident[min(len(live), 3)] += dt
                demand += dt * min(1.0, sum(min(f[j]["wgs"], 256) for j in live) / 256.0)
                if len(live) == 1:
                    solo[f[next(iter(live))]["cls"]] += dt
                elif len(live) >= 2:
                    cl = sorted(f[j]["cls"] for j in live)[:2]
                    pair[(cl[0], cl[1])] += dt
            last = t
            if kind: live.add(i)
            else: live.discard(i)
        print("  resident kernels: " + ", ".join(f"{n}{'+' if n == 3 else ''}: {resident[n]:.1f} us ({100 * resident[n] / wall:.1f} %)" for n in sorted(resident)))
        print(f"  CU demand (sum of min(workgroups, 256) / 256 over resident kernels, capped at 1): {100 * demand / wall:.1f} % of the wall")
        print("  alone:   " + ", ".join(f"{c} {v:.0f}" for c, v in sorted(solo.items(), key=lambda x: -x[1])))
        print("  co-run:  " + ", ".join(f"{a_}|{b_} {v:.0f}" for (a_, b_), v in sorted(pair.items(), key=lambda x: -x[1])[:14]))
        cls_t = defaultdict(float); cls_n = defaultdict(int)
        for k in f:
            cls_t[k["cls"]] += (k["e"] - k["s"]) / 1e3; cls_n[k["cls"]] += 1
        print("  per class (sum us / launches / mean us): " + ", ".join(f"{c} {cls_t[c]:.0f}/{cls_n[c]}/{cls_t[c] / cls_n[c]:.1f}" for c in sorted(cls_t, key=lambda c: -cls_t[c])))


if __name__ == "__main__":
    main()
