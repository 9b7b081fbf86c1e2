"""Race hunt: repeat the forward many times and require bit-identical probabilities every time.
python tools/soak_determinism.py [iters] [model]   -- batches 256 / 203 / 64 / 8 / 1, fp16 and bf16; default ViT-B/16
(ViT-tiny, K = 192, additionally takes the persistent stream kernel for its wide shapes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _pkg; pkg = _pkg.load()
from vitcpp_amd import binding as B
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
name = sys.argv[2] if len(sys.argv) > 2 else "vit_base_patch16_224"
path = pkg.synth.cached_synthetic(name, head_scale=8.0)
hp = pkg.synth.hparams_for(name)
m = B.Model(path)
bad = 0
for dt, dn in ((B.BF16, "bf16"), (B.F16, "f16")):
    for batch in (256, 203, 64, 8, 1):
        ctx = B.Context(m, 0, batch, dt)
        g = torch.Generator(device="cpu").manual_seed(batch)
        imgs = torch.randn((batch, hp.img_size, hp.img_size, 3), generator=g).cuda()
        probs = torch.empty((batch, hp.num_classes), device="cuda"); s = torch.cuda.current_stream().cuda_stream
        ctx.forward_device(imgs.data_ptr(), batch, probs.data_ptr(), 0, s); torch.cuda.synchronize()
        ref = probs.clone()
        n = iters if batch >= 64 else iters * 4
        mism = 0
        t0 = time.perf_counter()
        for i in range(n):
            probs.zero_()
            ctx.forward_device(imgs.data_ptr(), batch, probs.data_ptr(), 0, s)
            if i % 8 == 7 or i == n - 1:
                torch.cuda.synchronize()
                if not torch.equal(probs, ref): mism += 1
        torch.cuda.synchronize()
        print(f"{dn} batch {batch:3d}: {n} forwards, {mism} mismatching checks, {time.perf_counter()-t0:.1f} s", flush=True)
        bad += mism
        ctx.close()
print("SOAK", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
