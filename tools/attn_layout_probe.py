"""Is the headline attention bound by the QKV layout?  Same number of (image, head) items, same bytes per item, two row strides:
    [n_img = 128][197][3 x 768]  (row stride 4608 B: an item's 128-byte pieces lie 4.6 KB apart, the other 11 heads' pieces in between)
    [n_img = 1536][197][3 x 64]  (H = 1: row stride 384 B, every byte of a row belongs to the item)
bf16 persistent kernel (vitx_op_attention_ex kernel 4) and the F16 parity mode's precise kernel (vitx_op_attention_planes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _pkg; _pkg.load()
from vitcpp_amd import binding as B
L = B.lib(); s = torch.cuda.current_stream().cuda_stream
N = 197
def t(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for rnd in range(3):
    for n_img, H in ((128, 12), (1536, 1)):
        D = H * 64
        qkv = (torch.randn((n_img * N, 3 * D), device="cuda") * 0.8).to(torch.bfloat16)
        out = torch.zeros((n_img * N, D), device="cuda", dtype=torch.bfloat16)
        us = t(lambda: B.check(L.vitx_op_attention_ex(B.BF16, B.ATTN_PERSIST, qkv.data_ptr(), out.data_ptr(), n_img, N, D, H, s)))
        gb = n_img * N * 4 * D * 2 / 1e9
        planes = (torch.randn((2 * n_img * N, 3 * D), device="cuda") * 0.5).to(torch.float16)
        out16 = torch.zeros((n_img * N, D), device="cuda", dtype=torch.float16)
        usp = t(lambda: B.check(L.vitx_op_attention_planes(planes.data_ptr(), n_img * N * 3 * D, out16.data_ptr(), n_img, N, D, H, s)))
        gbp = n_img * N * 7 * D * 2 / 1e9
        print(f"round {rnd}: {n_img:5d} images x {H:2d} heads (row stride {3 * D * 2:5d} B): bf16 persistent {us:6.1f} us = {gb / us * 1e3:5.2f} TB/s | F16 precise {usp:6.1f} us = {gbp / usp * 1e3:5.2f} TB/s", flush=True)
