"""Time one GEMM shape on every kernel family (vitx_op_gemm_ex kernel ids): python tools/gemm_families.py [dtype] [iters]
Shapes are the encoder GEMMs of ViT-B at 16 and 64 images (the batch sizes that select the 64x128 and 128x256 ring tiles)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _pkg; _pkg.load()
from vitcpp_amd import binding as B
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dt = B.F16 if dtype == "f16" else B.BF16
tdt = torch.float16 if dtype == "f16" else torch.bfloat16
L = B.lib()
SHAPES = [("qkv", 2304, 768, 0), ("proj", 768, 768, 2), ("fc1", 3072, 768, 1), ("fc2", 768, 3072, 2)]
for n_img in (1, 4, 16, 64):
    M = (n_img * 197 + 255) // 256 * 256
    for name, N, K, epi in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(1)
        A = (torch.randn((M, K), device="cuda", generator=g) * 0.5).to(tdt)
        W = (torch.randn((N, K), device="cuda", generator=g) * 0.05).to(tdt)
        bias = torch.randn(N, device="cuda", generator=g) * 0.1
        out = torch.zeros((M, N), device="cuda", dtype=torch.float32 if epi >= 2 else tdt)
        s = torch.cuda.current_stream().cuda_stream
        row = f"{n_img:3d} img {name:5s} M={M:6d} N={N:5d} K={K:5d} {dtype}:"
        for kernel in (0, 122, 245, 945, 1):
            call = lambda: L.vitx_op_gemm_ex(dt, epi, kernel, A.data_ptr(), W.data_ptr(), bias.data_ptr(), out.data_ptr(), None, M, M - 59, N, K, 0, s)
            if call() != 0: row += f"  {kernel:>4d}: unsupported"; continue
            for _ in range(3): call()
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): call()
            e1.record(); torch.cuda.synchronize()
            row += f"  {kernel:>4d}: {e0.elapsed_time(e1) / iters * 1e3:7.1f} us"
        print(row, flush=True)
