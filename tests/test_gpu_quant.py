"""Block-quantised weights ON THE DEVICE (BASELINE.json config 5; reference: ggml keeps q4_0 ... q8_0 tensors in block form
through compute, /root/reference/vit.cpp:384-414, 645-678; quantize.cpp:271-303 writes them).

 * vitx_op_dequant (quant.hip) is bit-exact against the numpy restatement of ggml's dequantize_row_* rounded once to the operand
   type, for every block type and both operand types;
 * the fused q4_0 GEMM (blocks expanded in the GEMM's LDS-fill path) matches a float64 product of the dequantised operands;
 * a context built from a quantised file holds blocks (4.5 bits per weight for q4_0), gives the same answers as the context
   that expands on the host at upload (bit-identical where the same GEMM kernels run), and stays within the north_star tolerance
   of the oracle.
"""
import dataclasses
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BLOCK_BYTES = {2: 18, 3: 20, 6: 22, 7: 24, 8: 34}


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _round_to(torch, f32: np.ndarray, dtype) -> np.ndarray:
    """f32 -> operand type (round to nearest even) -> bits as u16."""
    t = torch.from_numpy(np.ascontiguousarray(f32)).to(torch.float16 if dtype == 0 else torch.bfloat16)
    return t.view(torch.int16).numpy().view(np.uint16)


def _split_q4_0(raw: bytes, n_rows: int, n_pad: int, nbk: int):
    """File layout (18-byte blocks) -> nibble plane [n_pad][nbk][16] + f16 scale plane [n_pad][nbk], zero pad rows."""
    b = np.frombuffer(raw, np.uint8).reshape(n_rows, nbk, 18)
    qs = np.zeros((n_pad, nbk, 16), np.uint8); ds = np.zeros((n_pad, nbk), np.uint16)
    qs[:n_rows] = b[:, :, 2:]
    ds[:n_rows] = b[:, :, 0:2].copy().view(np.uint16)[:, :, 0]
    return qs, ds


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("qtype", [2, 3, 6, 7, 8])
def test_op_dequant_bit_exact(pkg, binding, torch_gpu, qtype, dtype):
    torch = torch_gpu
    G = pkg.ggml_file
    N, n_pad, K = 200, 256, 448                      # ragged row count, pad rows, K not a power of two
    rng = np.random.default_rng(qtype * 10 + dtype)
    w = (rng.standard_normal((N, K)) * rng.uniform(0.01, 2.0, (N, 1))).astype(np.float32)
    w[3, :32] = 0.0                                   # an all-zero block (d = 0)
    raw = G.QUANTIZERS[qtype](w)
    want = _round_to(torch, G.dequantize(qtype, raw, N * K).reshape(N, K), dtype)
    if qtype == 2:
        qs, ds = _split_q4_0(raw, N, N, K // 32)
        d_blocks, d_scales = _dev(torch, qs), _dev(torch, ds)
    else:
        d_blocks, d_scales = _dev(torch, np.frombuffer(raw, np.uint8)), None
    out = torch.full((n_pad, K), 7.0, device="cuda").to(torch.float16 if dtype == 0 else torch.bfloat16)
    binding.check(binding.lib().vitx_op_dequant(dtype, qtype, d_blocks.data_ptr(), d_scales.data_ptr() if d_scales is not None else None,
                                                out.data_ptr(), N, n_pad, K, None), "vitx_op_dequant")
    torch.cuda.synchronize()
    got = out.view(torch.int16).cpu().numpy().view(np.uint16)
    assert np.array_equal(got[:N], want), f"{np.count_nonzero(got[:N] != want)} of {want.size} values differ"
    assert not got[N:].any()                          # pad rows are written as zeros


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("M,N,K,epi", [(256, 576, 192, 0), (640, 768, 768, 1), (256, 192, 768, 2), (128, 1000, 192, 3), (384, 2304, 768, 0)])
def test_op_gemm_q4_fused(pkg, binding, torch_gpu, M, N, K, epi, dtype):
    """q4_0 blocks expanded inside the GEMM (gemm_nt_kernel<.., Q4>): against the float64 product of the SAME dequantised,
    operand-rounded weights (products of two 16-bit operands are exact in f64; only the f32 summation order is the kernel's)."""
    torch = torch_gpu
    G = pkg.ggml_file
    rng = np.random.default_rng(M + 3 * N + 5 * K + epi + dtype)
    tdt = torch.float16 if dtype == 0 else torch.bfloat16
    a = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(tdt)
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    raw = G.QUANTIZERS[2](w)
    n_pad = (N + 127) // 128 * 128
    qs, ds = _split_q4_0(raw, N, n_pad, K // 32)
    wq = torch.from_numpy(G.dequantize(2, raw, N * K).reshape(N, K)).to(tdt)          # what the kernel must be multiplying with
    bias = np.zeros(n_pad, np.float32); bias[:N] = (rng.standard_normal(N) * 0.1).astype(np.float32)
    M_real = M - 5
    acc = a.double().numpy() @ wq.double().numpy().T + bias[:N]
    out_t = tdt if epi in (0, 1) else torch.float32
    resid = rng.standard_normal((M, N)).astype(np.float32)
    out = (torch.from_numpy(resid) if epi == 2 else torch.full((M, N), 3.0)).to(out_t).cuda()
    da, dq, dd, db = a.cuda(), _dev(torch, qs), _dev(torch, ds), _dev(torch, bias)
    binding.check(binding.lib().vitx_op_gemm_q4(dtype, epi, da.data_ptr(), dq.data_ptr(), dd.data_ptr(), db.data_ptr(), out.data_ptr(), M, M_real, N, K, None), "vitx_op_gemm_q4")
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    ulp = 2.0 ** -10 if dtype == 0 else 2.0 ** -7
    if epi == 0:
        want = acc; tol = 2e-4 * (1 + np.abs(want)) + ulp * np.abs(want)
    elif epi == 1:
        x = torch.from_numpy(acc.astype(np.float32)).to(tdt).double().numpy()
        want = 0.5 * x * (1 + np.tanh(0.7978845608028654 * x * (1 + 0.044715 * x * x)))
        tol = 3 * ulp * (np.abs(want) + 0.02)          # input and output rounding of the activation + the accumulation order
    elif epi == 2:
        want = acc + resid; tol = 2e-4 * (1 + np.abs(want))
    else:
        want = acc; tol = 2e-4 * (1 + np.abs(want))
    err = np.abs(got[:M_real] - want[:M_real])
    assert (err <= tol[:M_real]).all(), f"worst {float((err / tol[:M_real]).max()):.2f} tol"
    untouched = resid[M_real:] if epi == 2 else np.full((M - M_real, N), 3.0, np.float32)
    assert np.array_equal(got[M_real:], torch.from_numpy(untouched).to(out_t).float().numpy())     # rows beyond M_real are never stored


def _ctx_forward(binding, path, imgs, dtype, max_batch, **options):
    """options = fields of vitx_ctx_options (the library reads no environment variable)."""
    model = binding.Model(path)
    ctx = binding.Context(model, device=0, max_batch=max_batch, dtype=dtype, **options)
    probs, logits = ctx.forward(imgs, want_logits=True)
    wb = ctx.weight_bytes()
    ctx.close(); model.close()
    return probs, logits, wb


@pytest.mark.parametrize("ftype,ratio", [(2, 0.34), (3, 0.33), (6, 0.36), (7, 0.39), (8, 0.55)])
def test_quantised_context_keeps_blocks_in_hbm(pkg, binding, oracle, torch_gpu, tmp_path, ftype, ratio):
    """Device-resident blocks vs expand-at-upload (vitx_ctx_options::quant_on_host): the just-in-time expansion feeds the SAME GEMM kernels
    the same operand bits -> bit-identical logits; the fused q4_0 GEMM only changes the summation order; both within 1e-3 of the
    oracle on the same dequantised weights (quant_act=0, as in test_gpu_e2e.test_quantised_file_runs_dequantised)."""
    name = "vit_tiny_patch16_224"
    p = str(tmp_path / "q.gguf")
    pkg.synth.write_synthetic(p, name, ftype=ftype, head_scale=4.0)
    imgs = pkg.synth.normalize_u8(pkg.synth.synthetic_images_u8(24, 224))
    host_p, host_l, host_b = _ctx_forward(binding, p, imgs, binding.F16, 24, quant_on_host=1)
    jit_p, jit_l, jit_b = _ctx_forward(binding, p, imgs, binding.F16, 24, q4_fused_rows=0)          # every matrix expanded just in time
    assert np.array_equal(jit_l, host_l) and np.array_equal(jit_p, host_p)
    # the patch-embedding kernel stays f16 in a quantised file (4-D tensor, quantize.cpp:207-223), so the ratio is a little above bits/16
    assert jit_b <= ratio * host_b, (jit_b, host_b)
    if ftype == 2:
        fus_p, fus_l, fus_b = _ctx_forward(binding, p, imgs[:3], binding.F16, 3, q4_fused_rows=4096)  # 3 images: 768 rows -> fused kernel
        assert fus_b == jit_b
        assert np.abs(fus_p - host_p[:3]).max() <= 2e-4
        _, want = oracle.OracleModel(p).forward(imgs[:3], dataclasses.replace(oracle.REF, quant_act=0))
        assert np.abs(fus_p - want).max() <= 1e-3
        assert np.abs(jit_p[:3] - want).max() <= 1e-3


def test_q4_0_base_model_bf16_wide_path(pkg, binding, torch_gpu, tmp_path):
    """ViT-B q4_0 at a batch that takes the wide-tile kernels (BASELINE config 5's shape class): blocks in HBM + per-layer expansion
    is bit-identical to expand-at-upload, in the bench's dtype."""
    name = "vit_base_patch16_224"
    src = pkg.synth.cached_synthetic(name, head_scale=8.0)
    p = str(tmp_path / "b_q4_0.gguf")
    binding.quantize_file(src, p, 2)
    imgs = pkg.synth.normalize_u8(pkg.synth.synthetic_images_u8(32, 224))
    host_p, host_l, host_b = _ctx_forward(binding, p, imgs, binding.BF16, 32, quant_on_host=1)
    dev_p, dev_l, dev_b = _ctx_forward(binding, p, imgs, binding.BF16, 32, q4_fused_rows=0)
    assert np.array_equal(dev_l, host_l)
    assert dev_b < 0.30 * host_b
