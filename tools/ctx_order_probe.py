"""Does the two-stream overlap depend on WHICH context of a process runs?  (r03: the 2nd and 6th / 7th context created in one process ran
11.8 instead of 9.9 ms per ViT-B forward, serial per-kernel times unchanged -- i.e. the sub-batch streams did not overlap.)
    python tools/ctx_order_probe.py [n_contexts] [keep|close]     keep: all contexts stay alive; close: each is closed before the next is made"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _pkg; pkg = _pkg.load()
from vitcpp_amd import binding as B
n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 8
keep = (sys.argv[2] if len(sys.argv) > 2 else "keep") == "keep"
path = pkg.synth.cached_synthetic("vit_base_patch16_224", head_scale=8.0)
m = B.Model(path)
imgs = torch.randn((256, 224, 224, 3), device="cuda"); probs = torch.empty((256, 1000), device="cuda")
st = torch.cuda.Stream(); s = st.cuda_stream
alive = []
for k in range(n_ctx):
    c = B.Context(m, 0, 256, B.BF16)
    for _ in range(3): c.forward_device(imgs.data_ptr(), 256, probs.data_ptr(), 0, s)
    ts = []
    for r in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): c.forward_device(imgs.data_ptr(), 256, probs.data_ptr(), 0, s)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e3)
    print(f"context #{k + 1} ({'others alive' if keep else 'previous closed'}; GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')}): {min(ts):.3f} ms  (internal streams re-created: {c.stream_retries()})", flush=True)
    if keep: alive.append(c)
    else: c.close()
