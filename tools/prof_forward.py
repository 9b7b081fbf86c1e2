"""Run a few ViT forwards (for rocprofv3): python tools/prof_forward.py [model] [batch] [steps] [dtype] [opt=val,...]
(context options, e.g. streams=1; profile=1 = the context's profiling schedule: the two sub-batches back to back on one stream, exclusive kernel durations)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _pkg; pkg = _pkg.load()
from vitcpp_amd import binding as B
name = sys.argv[1] if len(sys.argv) > 1 else "vit_base_patch16_224"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dt = B.BF16 if (len(sys.argv) > 4 and sys.argv[4] == "bf16") else B.F16
path = pkg.synth.cached_synthetic(name, head_scale=8.0)
hp = pkg.synth.hparams_for(name)
opts = {k: int(v) for k, v in (o.split("=") for o in (sys.argv[5].split(",") if len(sys.argv) > 5 else []) if o)}
profile = opts.pop("profile", 0)
m = B.Model(path); ctx = B.Context(m, 0, batch, dt, **opts)
if profile: ctx.profile_enable(True)          # sub-batches back to back on one stream, as in bench.py's profiled step
imgs = torch.randn((batch, hp.img_size, hp.img_size, 3), device="cuda")
probs = torch.empty((batch, hp.num_classes), device="cuda")
for _ in range(steps):
    ctx.forward_device(imgs.data_ptr(), batch, probs.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("done", float(probs.sum()))
