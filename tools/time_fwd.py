"""Wall-clock of the forward only (no per-kernel events): python tools/time_fwd.py [batch] [model] [dtype] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _pkg; pkg = _pkg.load()
from vitcpp_amd import binding as B
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
name = sys.argv[2] if len(sys.argv) > 2 else "vit_base_patch16_224"
dt = B.F16 if (len(sys.argv) > 3 and sys.argv[3] == "f16") else B.BF16
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 30
ftype = {"f16": 1, "q4_0": 2, "q4_1": 3, "q5_0": 6, "q5_1": 7, "q8_0": 8}[os.environ.get("TF_FTYPE", "f16")]      # weight file type
path = pkg.synth.cached_synthetic(name, ftype=ftype, head_scale=8.0)
hp = pkg.synth.hparams_for(name)
m = B.Model(path); ctx = B.Context(m, 0, batch, dt)
imgs = torch.randn((batch, hp.img_size, hp.img_size, 3), device="cuda"); probs = torch.empty((batch, hp.num_classes), device="cuda")
s = torch.cuda.current_stream().cuda_stream
if os.environ.get("TF_STREAM"):          # an explicit torch stream, as bench.py uses
    _st = torch.cuda.Stream(); s = _st.cuda_stream
for _ in range(5): ctx.forward_device(imgs.data_ptr(), batch, probs.data_ptr(), 0, s)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(iters): ctx.forward_device(imgs.data_ptr(), batch, probs.data_ptr(), 0, s)
torch.cuda.synchronize(); dt_ = (time.perf_counter() - t0) / iters
# latency of one isolated call (enqueue + sync)
lat = []
for _ in range(20):
    t1 = time.perf_counter(); ctx.forward_device(imgs.data_ptr(), batch, probs.data_ptr(), 0, s); torch.cuda.synchronize(); lat.append(time.perf_counter() - t1)
print(f"{name} {os.environ.get('TF_FTYPE', 'f16')}-file batch {batch}: {dt_*1e3:.3f} ms/step pipelined ({batch/dt_:.0f} img/s); isolated call median {sorted(lat)[10]*1e3:.3f} ms")
ctx.close(); m.close()
