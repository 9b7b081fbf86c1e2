"""The fused LayerNorm under contention (r04 verdict item 7): the bf16 forward beside a SECOND PROCESS that keeps the device busy.
    python tools/ln_contention.py [--seconds S]
Co-tenants: (a) none, (b) another process looping the same ViT-B/16 batch-256 forward, (c) another process looping vitx_probe_mfma
(back-to-back MFMAs on every CU).  For each: ms per forward with the fusion on and off (interleaved rounds), fall-back tiles per forward
(vitx_ctx_ln_fallbacks), and what the fall-back budget did (vitx_ctx_ln_fusion_active sampled every forward: forwards until it first reads -1,
re-arms seen)."""
import argparse, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=6.0)
ap.add_argument("--role", default="parent")
a = ap.parse_args()

import torch
import _pkg; pkg = _pkg.load()
from vitcpp_amd import binding as B

name = "vit_base_patch16_224"
path = pkg.synth.cached_synthetic(name, head_scale=8.0)
hp = pkg.synth.hparams_for(name)
n = 256

if a.role in ("forward", "mfma"):          # co-tenant: runs until killed
    if a.role == "forward":
        m = B.Model(path); c = B.Context(m, 0, n, B.BF16)
        imgs = torch.randn((n, 224, 224, 3), device="cuda"); probs = torch.empty((n, hp.num_classes), device="cuda")
        st = torch.cuda.Stream()
        print("ready", flush=True)
        while True:
            for _ in range(20): c.forward_device(imgs.data_ptr(), n, probs.data_ptr(), 0, st.cuda_stream)
            torch.cuda.synchronize()
    else:
        print("ready", flush=True)
        while True:
            B.probe_mfma(0, B.BF16, 2, 200.0)

m = B.Model(path)
imgs = torch.randn((n, 224, 224, 3), device="cuda"); probs = torch.empty((n, hp.num_classes), device="cuda")
st = torch.cuda.Stream(); s = st.cuda_stream

def measure(label):
    ctxs = {"fused": B.Context(m, 0, n, B.BF16), "unfused": B.Context(m, 0, n, B.BF16, no_ln_fusion=1)}
    for c in ctxs.values():
        for _ in range(3): c.forward_device(imgs.data_ptr(), n, probs.data_ptr(), 0, s)
    torch.cuda.synchronize()
    ts = {k: [] for k in ctxs}; fwd = {k: 0 for k in ctxs}
    fb0 = ctxs["fused"].ln_fallbacks()
    first_off, rearms, state_prev, nf = None, 0, 1, 0
    t_end = time.perf_counter() + a.seconds
    while time.perf_counter() < t_end:
        for k, c in ctxs.items():
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(16):
                c.forward_device(imgs.data_ptr(), n, probs.data_ptr(), 0, s)
                if k == "fused":
                    nf += 1
                    stt = c.ln_fusion_active()
                    if stt == -1 and first_off is None: first_off = nf
                    if stt == 1 and state_prev == -1: rearms += 1
                    state_prev = stt
            torch.cuda.synchronize(); ts[k].append((time.perf_counter() - t0) / 16 * 1e3); fwd[k] += 16
    fb = ctxs["fused"].ln_fallbacks() - fb0
    med = {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
    print(f"{label:34s} fused {med['fused']:7.3f} ms  un-fused {med['unfused']:7.3f} ms  | fused forwards {fwd['fused']}, fall-back tiles {fb} ({fb / max(fwd['fused'], 1):.2f} per forward), "
          f"budget tripped at forward {first_off}, re-arms {rearms}, state now {ctxs['fused'].ln_fusion_active()}", flush=True)
    for c in ctxs.values(): c.close()

measure("alone")
for role, label in (("forward", "beside another ViT-B forward loop"), ("mfma", "beside an MFMA burner (all CUs)")):
    p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--role", role], stdout=subprocess.PIPE, text=True)
    try:
        line = p.stdout.readline()
        assert "ready" in line, line
        time.sleep(1.0)
        measure(label)
    finally:
        p.kill(); p.wait()
