"""Interleaved A/B timing of several BUILDS of libvitx.so in ONE process (boxes and DVFS drift make separate runs incomparable):
    python tools/ab_libs.py [--model M] [--batch B] [--dtype bf16|f16] [--rounds R] [--steps S] [--breakdown] name=path/to/lib.so[:opt=val,...] ...
Each library is loaded through its own copy of the binding module; rounds of S forwards run build after build, R times."""
import argparse, importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import _pkg; pkg = _pkg.load()

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="vit_base_patch16_224"); ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--dtype", default="bf16"); ap.add_argument("--rounds", type=int, default=5); ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--breakdown", action="store_true"); ap.add_argument("--check", action="store_true", help="compare every build's probabilities with the first one's")
ap.add_argument("variants", nargs="+")
a = ap.parse_args()
path = pkg.synth.cached_synthetic(a.model, head_scale=8.0)
hp = pkg.synth.hparams_for(a.model)
imgs = torch.randn((a.batch, hp.img_size, hp.img_size, 3), device="cuda")
st = torch.cuda.Stream(); s = st.cuda_stream
runs = []
for i, v in enumerate(a.variants):
    name, _, rest = v.partition("=")
    lib, _, opts = rest.partition(":")
    os.environ["VITX_LIB"] = os.path.abspath(lib)
    spec = importlib.util.spec_from_file_location(f"binding_ab{i}", os.path.join(ROOT, "vit.cpp_amd", "binding.py"))
    B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
    kw = {k: int(x) for k, x in (o.split("=") for o in opts.split(",") if o)}
    m = B.Model(path)
    c = B.Context(m, 0, a.batch, B.F16 if a.dtype == "f16" else B.BF16, **kw)
    runs.append(dict(name=name, B=B, m=m, c=c, ts=[], probs=torch.empty((a.batch, hp.num_classes), device="cuda")))
for r in runs:
    for _ in range(3): r["c"].forward_device(imgs.data_ptr(), a.batch, r["probs"].data_ptr(), 0, s)
torch.cuda.synchronize()
for _ in range(a.rounds):
    for r in runs:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.steps): r["c"].forward_device(imgs.data_ptr(), a.batch, r["probs"].data_ptr(), 0, s)
        torch.cuda.synchronize(); r["ts"].append((time.perf_counter() - t0) / a.steps * 1e3)
for r in runs:
    ts = sorted(r["ts"]); c = r["c"]
    if a.breakdown:
        c.profile_enable(True); c.forward_device(imgs.data_ptr(), a.batch, r["probs"].data_ptr(), 0, s); torch.cuda.synchronize()
        pr = c.profile_read(); c.profile_enable(False)
        print("   " + "  ".join(f"{p['name']}={p['busy_ms']:.3f}/{p['launches']}" for p in pr) + f"  sum={sum(p['busy_ms'] for p in pr):.3f}")
    extra = ""
    if a.check:
        d = (r["probs"] - runs[0]["probs"]).abs().max().item()
        extra = f"  max|dp| vs {runs[0]['name']} {d:.3e}" + ("  (bit-identical)" if torch.equal(r["probs"], runs[0]["probs"]) else "")
    print(f"{a.model} b{a.batch} {a.dtype}  {r['name']:20s} median {ts[len(ts)//2]:.3f} ms  min {ts[0]:.3f}  max {ts[-1]:.3f}  ({a.batch / ts[len(ts)//2] * 1e3:.0f} img/s)  ln_fallbacks {c.ln_fallbacks()}{extra}")
