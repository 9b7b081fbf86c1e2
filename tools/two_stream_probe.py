"""Probe: does running two half-batches on two HIP streams (tails of one fill under the other) beat one full batch?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _pkg; pkg = _pkg.load()
from vitcpp_amd import binding as B
name = "vit_base_patch16_224"; hp = pkg.synth.hparams_for(name)
path = pkg.synth.cached_synthetic(name, head_scale=8.0)
m = B.Model(path)
def run(nctx, per, steps=10):
    ctxs = [B.Context(m, 0, per, B.BF16) for _ in range(nctx)]
    streams = [torch.cuda.Stream() for _ in range(nctx)]
    imgs = [torch.randn((per, 224, 224, 3), device="cuda") for _ in range(nctx)]
    probs = [torch.empty((per, 1000), device="cuda") for _ in range(nctx)]
    def step():
        for c, s, i, p in zip(ctxs, streams, imgs, probs):
            c.forward_device(i.data_ptr(), per, p.data_ptr(), 0, s.cuda_stream)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{nctx} stream(s) x {per} images: {nctx*per*steps/dt:.0f} img/s ({dt/steps*1e3:.2f} ms per {nctx*per})")
    for c in ctxs: c.close()
run(1, 256); run(2, 128); run(2, 256); run(4, 64); run(1, 256)
