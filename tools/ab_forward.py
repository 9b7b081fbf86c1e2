"""Interleaved A/B timing of context options in ONE process (boxes and DVFS drift make separate runs incomparable):
    python tools/ab_forward.py [--model M] [--batch B] [--dtype bf16|f16] [--ftype f16|q4_0] [--rounds R] [--steps S] name[:opt=val,...] ...
Every variant gets its own context; rounds of S forwards are run variant after variant, R times; prints median / min ms per forward."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _pkg; pkg = _pkg.load()
from vitcpp_amd import binding as B

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="vit_base_patch16_224"); ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--dtype", default="bf16"); ap.add_argument("--ftype", default="f16")
ap.add_argument("--rounds", type=int, default=5); ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--breakdown", action="store_true", help="also one profiled step per variant: per-kernel-class busy ms (sub-batches serialised)")
ap.add_argument("variants", nargs="+")
a = ap.parse_args()
ftype = {"f16": 1, "q4_0": 2, "q4_1": 3, "q5_0": 6, "q5_1": 7, "q8_0": 8}[a.ftype]
path = pkg.synth.cached_synthetic(a.model, ftype=ftype, head_scale=8.0)
hp = pkg.synth.hparams_for(a.model)
m = B.Model(path)
dt = B.F16 if a.dtype == "f16" else B.BF16
imgs = torch.randn((a.batch, hp.img_size, hp.img_size, 3), device="cuda"); probs = torch.empty((a.batch, hp.num_classes), device="cuda")
st = torch.cuda.Stream(); s = st.cuda_stream
ctxs = []
for v in a.variants:
    name, _, opts = v.partition(":")
    kw = {k: int(x) for k, x in (o.split("=") for o in opts.split(",") if o)}
    ctxs.append((name, B.Context(m, 0, a.batch, dt, **kw), []))
for name, c, _ in ctxs:
    for _ in range(3): c.forward_device(imgs.data_ptr(), a.batch, probs.data_ptr(), 0, s)
torch.cuda.synchronize()
for r in range(a.rounds):
    for name, c, ts in ctxs:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.steps): c.forward_device(imgs.data_ptr(), a.batch, probs.data_ptr(), 0, s)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / a.steps * 1e3)
for name, c, ts in ctxs:
    ts.sort()
    if a.breakdown:
        c.profile_enable(True); c.forward_device(imgs.data_ptr(), a.batch, probs.data_ptr(), 0, s); torch.cuda.synchronize()
        pr = c.profile_read(); c.profile_enable(False)
        print("   " + "  ".join(f"{p['name']}={p['busy_ms']:.3f}/{p['launches']}" for p in pr) + f"  sum={sum(p['busy_ms'] for p in pr):.3f}")
    print(f"{a.model} b{a.batch} {a.dtype} {a.ftype}-file  {name:24s} median {ts[len(ts)//2]:.3f} ms  min {ts[0]:.3f}  max {ts[-1]:.3f}  ({a.batch / ts[len(ts)//2] * 1e3:.0f} img/s)  ln_fallbacks {c.ln_fallbacks()}")
    c.close()
