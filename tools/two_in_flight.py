"""Throughput with TWO forwards in flight: two contexts without an internal sub-batch split (streams = 1), each fed a whole batch from its
own caller stream, against one context that splits every batch into two sub-batches on two streams (the default).
    python tools/two_in_flight.py [--model M] [--batch B] [--rounds R] [--steps S]
Same images/s accounting for both: forwards completed * batch / wall time; rounds are interleaved in one process."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _pkg; pkg = _pkg.load()
from vitcpp_amd import binding as B

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="vit_base_patch16_224"); ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--dtype", default="bf16"); ap.add_argument("--rounds", type=int, default=4); ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--ctxs", type=int, default=2, help="contexts = forwards in flight"); ap.add_argument("--streams", type=int, default=1, help="sub-batch streams of each of them")
a = ap.parse_args()
path = pkg.synth.cached_synthetic(a.model, head_scale=8.0)
hp = pkg.synth.hparams_for(a.model)
m = B.Model(path)
dt = B.F16 if a.dtype == "f16" else B.BF16
NC = a.ctxs
imgs = [torch.randn((a.batch, hp.img_size, hp.img_size, 3), device="cuda") for _ in range(NC)]
probs = [torch.empty((a.batch, hp.num_classes), device="cuda") for _ in range(NC)]
sts = [torch.cuda.Stream() for _ in range(NC)]
split = B.Context(m, 0, a.batch, dt)
pair = [B.Context(m, 0, a.batch, dt, streams=a.streams) for _ in range(NC)]
def run_split(n):
    for _ in range(n): split.forward_device(imgs[0].data_ptr(), a.batch, probs[0].data_ptr(), 0, sts[0].cuda_stream)
def run_pair(n):
    for i in range(n): pair[i % NC].forward_device(imgs[i % NC].data_ptr(), a.batch, probs[i % NC].data_ptr(), 0, sts[i % NC].cuda_stream)
run_split(3); run_pair(2 * NC); torch.cuda.synchronize()
res = {"one context, two sub-batches": [], f"{NC} contexts x {a.streams} stream(s) in flight": []}
for r in range(a.rounds):
    for name, fn in (("one context, two sub-batches", run_split), (f"{NC} contexts x {a.streams} stream(s) in flight", run_pair)):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(a.steps); torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / a.steps * 1e3)
for name, ts in res.items():
    ts.sort(); med = ts[len(ts) // 2]
    print(f"{a.model} b{a.batch} {a.dtype}  {name:40s} {med:.3f} ms per forward (min {ts[0]:.3f} max {ts[-1]:.3f})  {a.batch / med * 1e3:.0f} img/s")
# the two schedules compute the same thing
split.forward_device(imgs[1].data_ptr(), a.batch, probs[0].data_ptr(), 0, sts[0].cuda_stream); torch.cuda.synchronize(); ref = probs[0].clone()
pair[1].forward_device(imgs[1].data_ptr(), a.batch, probs[1].data_ptr(), 0, sts[1].cuda_stream); torch.cuda.synchronize()
print("bit-identical probabilities:", torch.equal(ref, probs[1]))
