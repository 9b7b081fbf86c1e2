"""Micro-benchmark of vitx_op_attention (ViT-B: 256 images x 12 heads x 197 tokens)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _pkg; _pkg.load()
from vitcpp_amd import binding as B
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 197
H = int(sys.argv[3]) if len(sys.argv) > 3 else 12
D = H * 64
dt = torch.float16
qkv = (torch.randn((n_img * N, 3 * D), device="cuda") * 0.8).to(dt)
out = torch.zeros((n_img * N, D), device="cuda", dtype=dt)
L = B.lib(); s = torch.cuda.current_stream().cuda_stream
for _ in range(3): B.check(L.vitx_op_attention(B.F16, qkv.data_ptr(), out.data_ptr(), n_img, N, D, H, s))
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): L.vitx_op_attention(B.F16, qkv.data_ptr(), out.data_ptr(), n_img, N, D, H, s)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"attention {n_img}x{H}x{N}: {ms*1e3:.1f} us  {4.0*n_img*H*N*N*64/ms/1e9:.1f} TF/s")
# check vs torch
q, k, v = qkv.float().view(n_img, N, 3, H, 64).permute(2, 0, 3, 1, 4)
ref = torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v
ref = ref.permute(0, 2, 1, 3).reshape(n_img * N, D)
print("max abs err vs torch f32:", (out.float() - ref).abs().max().item())
