import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import _pkg; pkg = _pkg.load()
from vitcpp_amd import binding as B
path = pkg.synth.cached_synthetic("vit_base_patch16_224", head_scale=8.0)
m = B.Model(path); ctx = B.Context(m, 0, 256, B.BF16)
imgs = torch.randn((256, 224, 224, 3), device="cuda"); probs = torch.empty((256, 1000), device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(5): ctx.forward_device(imgs.data_ptr(), 256, probs.data_ptr(), 0, s)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): ctx.forward_device(imgs.data_ptr(), 256, probs.data_ptr(), 0, s)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
print(f"{dt*1e3:.3f} ms/step  {256/dt:.0f} img/s")
