"""What the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on the ViT shapes, next to vitx_op_gemm on the same operands:
a reference point for the hand-written kernels, not a dependency (nothing in the product calls a BLAS).  Plain GEMMs: the vendor side
has no fused epilogue; the vitx side is timed with the epilogue the forward uses AND with the plain bias epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _pkg; _pkg.load()
from vitcpp_amd import binding as B

SHAPES = {"qkv": (2304, 768, 0), "proj": (768, 768, 2), "fc1": (3072, 768, 1), "fc2": (768, 3072, 2)}
L = B.lib()
def timed(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for M in (20480, 25344, 30208, 50432):      # 80 / 99 / 118 / 197 row blocks of 256 (the forward pads its sub-batches to whole blocks)
    for name, (N, K, epi) in SHAPES.items():
        g = torch.Generator(device="cuda").manual_seed(1)
        A = (torch.randn((M, K), device="cuda", generator=g) * 0.5).to(torch.bfloat16)
        W = (torch.randn((N, K), device="cuda", generator=g) * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda", generator=g) * 0.1
        out16 = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
        out32 = torch.zeros((M, N), device="cuda", dtype=torch.float32)
        s = torch.cuda.current_stream().cuda_stream
        Wt = W.t()
        t_blas = timed(lambda: torch.matmul(A, Wt, out=out16))
        t_plain = timed(lambda: L.vitx_op_gemm(B.BF16, 0, A.data_ptr(), W.data_ptr(), bias.data_ptr(), out16.data_ptr(), M, N, K, s))
        o = out32 if epi >= 2 else out16
        t_epi = timed(lambda: L.vitx_op_gemm(B.BF16, epi, A.data_ptr(), W.data_ptr(), bias.data_ptr(), o.data_ptr(), M, N, K, s))
        fl = 2.0 * M * N * K / 1e9
        print(f"M={M:6d} {name:5s} N={N:5d} K={K:5d}: vendor {t_blas*1e3:7.1f} us {fl/t_blas:7.1f} TF/s | vitx bias->bf16 {t_plain*1e3:7.1f} us {fl/t_plain:7.1f} TF/s | vitx forward epilogue {epi} {t_epi*1e3:7.1f} us {fl/t_epi:7.1f} TF/s", flush=True)
