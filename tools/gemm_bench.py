"""Micro-benchmark of vitx_op_gemm on the ViT GEMM shapes (HIP-event timing, random data)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import _pkg; _pkg.load()
from vitcpp_amd import binding as B

SHAPES = {  # name: (M, N, K, epi)
    "qkv": (50432, 2304, 768, 0), "proj": (50432, 768, 768, 2), "fc1": (50432, 3072, 768, 1), "fc2": (50432, 768, 3072, 2),
    "sq4k": (4096, 4096, 4096, 0), "sq8k": (8192, 8192, 8192, 0),
}
ap = argparse.ArgumentParser(); ap.add_argument("--shapes", default="qkv,proj,fc1,fc2"); ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--dtype", default="f16"); ap.add_argument("--check", action="store_true")
a = ap.parse_args()
dt = B.F16 if a.dtype == "f16" else B.BF16
tdt = torch.float16 if a.dtype == "f16" else torch.bfloat16
L = B.lib()
for name in a.shapes.split(","):
    M, N, K, epi = SHAPES[name]
    g = torch.Generator(device="cuda").manual_seed(1)
    A = (torch.randn((M, K), device="cuda", generator=g) * 0.5).to(tdt)
    W = (torch.randn((N, K), device="cuda", generator=g) * 0.05).to(tdt)
    bias = torch.randn(N, device="cuda", generator=g) * 0.1
    out = torch.zeros((M, N), device="cuda", dtype=torch.float32 if epi >= 2 else tdt)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3): B.check(L.vitx_op_gemm(dt, epi, A.data_ptr(), W.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, s))
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters): L.vitx_op_gemm(dt, epi, A.data_ptr(), W.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    print(f"{name:6s} M={M} N={N} K={K} epi={epi} {a.dtype}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF/s", flush=True)

    if a.check:
        out.zero_(); L.vitx_op_gemm(dt, epi if epi != 2 else 3, A.data_ptr(), W.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, s); torch.cuda.synchronize()
        ref = (A[:512].float() @ W.float().T + bias)
        if epi == 1: ref = torch.nn.functional.gelu(ref.to(tdt).float(), approximate="tanh")
        err = (out[:512].float() - ref).abs().max().item(); print(f"        max abs err vs torch (first 512 rows): {err:.3e}")
