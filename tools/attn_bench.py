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
kern = int(sys.argv[4]) if len(sys.argv) > 4 else 0        # 0 auto, 1 single-pass, 2 streaming
bf = len(sys.argv) > 5 and sys.argv[5] == "bf16"
dt = torch.bfloat16 if bf else torch.float16
DT = B.BF16 if bf else B.F16
qkv = (torch.randn((n_img * N, 3 * D), device="cuda") * 0.8).to(dt)
out_all = torch.zeros((n_img * N + 16, D), device="cuda", dtype=dt)       # + room for the lab build's clock stamps
out = out_all[:n_img * N]
L = B.lib(); s = torch.cuda.current_stream().cuda_stream
for _ in range(3): B.check(L.vitx_op_attention_ex(DT, kern, qkv.data_ptr(), out.data_ptr(), n_img, N, D, H, s))
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
ITERS = int(os.environ.get('ATTN_ITERS', '20'))
for _ in range(ITERS): L.vitx_op_attention_ex(DT, kern, qkv.data_ptr(), out.data_ptr(), n_img, N, D, H, s)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / ITERS
print(f"attention kernel {kern} {'bf16' if bf else 'f16'} {n_img}x{H}x{N}: {ms*1e3:.1f} us  {4.0*n_img*H*N*N*64/ms/1e9:.1f} TF/s")
# check vs torch
q, k, v = qkv.float().view(n_img, N, 3, H, 64).permute(2, 0, 3, 1, 4)
ref = torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v
ref = ref.permute(0, 2, 1, 3).reshape(n_img * N, D)
print("max abs err vs torch f32:", (out.float() - ref).abs().max().item())
if kern not in (1,) and kern < 16:      # bit-compare with the single-pass kernel where it is instantiated
    out1 = torch.zeros_like(out)
    if L.vitx_op_attention_ex(DT, 1, qkv.data_ptr(), out1.data_ptr(), n_img, N, D, H, s) == 0:
        torch.cuda.synchronize()
        print("bit-identical to single-pass:", torch.equal(out1.view(torch.int16), out.view(torch.int16)))

if kern >= 16 and (kern >> 4) & 64:      # lab build: per-phase shader-clock stamps of workgroup 0, [wave][item][6]
    torch.cuda.synchronize()
    NW = 16
    st = out_all[n_img * N:].view(torch.int64).flatten()[:NW * 6 * 6].view(NW, 6, 6).cpu()
    t00 = int(st[0, 0, 0])
    names = ["QK^T", "softmax", "PV", "stores+wait", "barrier"] if (kern >> 4) < 128 else ["even interval", "wait+barrier", "odd interval", "wait", "-"]
    for w in range(NW):
        for it in range(2, 4):
            d = [int(st[w, it, i + 1] - st[w, it, i]) for i in range(5)]
            print(f"wave {w} item {it}: start {int(st[w, it, 0]) - t00:7d}  " + "  ".join(f"{n} {x:5d}" for n, x in zip(names, d)) + f"   item {int(st[w, it, 5] - st[w, it, 0])}")
