"""Per-kernel-class HIP-event breakdown at small batch: python tools/small_batch_profile.py [batch] [model] [dtype]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _pkg; pkg = _pkg.load()
from vitcpp_amd import binding as B
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
name = sys.argv[2] if len(sys.argv) > 2 else "vit_base_patch16_224"
dt = B.F16 if (len(sys.argv) > 3 and sys.argv[3] == "f16") else B.BF16
path = pkg.synth.cached_synthetic(name, head_scale=8.0)
hp = pkg.synth.hparams_for(name)
m = B.Model(path); ctx = B.Context(m, 0, batch, dt)
imgs = torch.randn((batch, hp.img_size, hp.img_size, 3), device="cuda"); probs = torch.empty((batch, hp.num_classes), device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(5): ctx.forward_device(imgs.data_ptr(), batch, probs.data_ptr(), 0, s)
torch.cuda.synchronize()
ctx.profile_enable(True)
for _ in range(10): ctx.forward_device(imgs.data_ptr(), batch, probs.data_ptr(), 0, s)
prof = ctx.profile_read(); ctx.profile_enable(False)
tot = sum(p["total_ms"] for p in prof)
print(f"{name} batch {batch}: sum of kernel times {tot/10*1e3:.1f} us per forward")
for p in prof:
    print(f"  {p['name']:18s} {p['launches']//10:3d} launches  {p['total_ms']/p['launches']*1e3:7.1f} us each  {p['total_ms']/10*1e3:8.1f} us per forward")
ctx.close(); m.close()
