"""Per-kernel parity of the HIP path vs the CPU oracle, through the C ABI (include/vitx.h).

Every test feeds the same seeded inputs to one vitx_op_* entry point (device pointers
from torch tensors) and to the oracle's restatement of the same ggml op.
Tolerances: f32 accumulate-order noise only for GEMM (products are exact in f32 for
fp16 operands), plus at most one operand-type ulp where an output is rounded to fp16.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F16_ULP = 2.0 ** -10       # relative spacing of fp16 (11-bit significand)


def _dev(torch, a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


def _sync(torch):
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,D", [(197 * 3, 768), (197, 192), (64, 384), (130, 1024), (17 * 2, 128)])
def test_layernorm(binding, oracle, torch_gpu, M, D):
    torch = torch_gpu
    rng = np.random.default_rng(M * 7 + D)
    x = (rng.standard_normal((M, D)) * 0.7 + 0.1).astype(np.float32)
    w = (1 + 0.02 * rng.standard_normal(D)).astype(np.float32)
    b = (0.02 * rng.standard_normal(D)).astype(np.float32)
    ref = oracle.layernorm(x, w, b, 1e-6)
    dx, dw, db = _dev(torch, x), _dev(torch, w), _dev(torch, b)
    y = torch.empty((M, D), dtype=torch.float16, device="cuda")
    binding.check(binding.lib().vitx_op_layernorm(binding.F16, dx.data_ptr(), dw.data_ptr(), db.data_ptr(), y.data_ptr(), M, D, 1e-6, None))
    _sync(torch)
    got = y.float().cpu().numpy()
    ref16 = ref.astype(np.float16).astype(np.float32)
    # fp16 output: identical up to one fp16 ulp (f32 sum order of the mean/variance may flip a rounding)
    assert np.abs(got - ref16).max() <= np.abs(ref).max() * F16_ULP
    assert (got != ref16).mean() < 0.01


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (128, 128, 768), (384, 2304, 768), (256, 768, 3072), (128, 192, 192), (128, 576, 192)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3])
def test_gemm_epilogues_f16(binding, oracle, torch_gpu, M, N, K, epi):
    """C = A.W^T (+bias, gelu, residual) vs the oracle's dot product (ggml_vec_dot_f16 order)."""
    torch = torch_gpu
    rng = np.random.default_rng(M + N * 3 + K * 5 + epi)
    a = rng.standard_normal((M, K)).astype(np.float16)
    Npad = (N + 127) // 128 * 128
    w = np.zeros((Npad, K), np.float16); w[:N] = (rng.standard_normal((N, K)) * 0.05).astype(np.float16)
    bias = np.zeros(Npad, np.float32); bias[:N] = (rng.standard_normal(N) * 0.1).astype(np.float32)
    resid = rng.standard_normal((M, N)).astype(np.float32)
    # exact reference in float64 (fp16 products are exact; only the sum order differs)
    acc = a.astype(np.float64) @ w[:N].astype(np.float64).T
    v = (acc + bias[:N]).astype(np.float32)
    da, dw, db = _dev(torch, a), _dev(torch, w), _dev(torch, bias)
    if epi in (0, 1):
        out = torch.zeros((M, N), dtype=torch.float16, device="cuda")
    elif epi == 2:
        out = _dev(torch, resid)
    else:
        out = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    binding.check(binding.lib().vitx_op_gemm(binding.F16, epi, da.data_ptr(), dw.data_ptr(), db.data_ptr(), out.data_ptr(), M, N, K, None), "gemm")
    _sync(torch)
    got = out.float().cpu().numpy()
    scale = np.abs(a.astype(np.float64)) @ np.abs(w[:N].astype(np.float64)).T      # sum |a||w|: bound for f32 accumulation error
    tol_acc = (scale * 2e-6 + 1e-6).astype(np.float32)
    if epi == 0:
        ref = v
        assert (np.abs(got - ref) <= tol_acc + np.abs(ref) * F16_ULP).all()
    elif epi == 1:
        ref = oracle.gelu(v, lut=1)
        # one fp16 ulp on the input rounding can move gelu by ~1 ulp of |x|; allow 2 ulp of max(|x|,|y|) + accumulate noise
        assert (np.abs(got - ref) <= tol_acc * 2 + np.maximum(np.abs(ref), np.abs(v)) * 2 * F16_ULP + 1e-6).all()
    elif epi == 2:
        ref = v + resid
        assert (np.abs(got - ref) <= tol_acc + np.abs(ref) * 2e-7 + 1e-7).all()
    else:
        assert (np.abs(got - v) <= tol_acc).all()


def test_gemm_transpose_detecting(binding, torch_gpu):
    """A = I-like selector with an ASYMMETRIC W: catches row/col swaps of the MFMA C layout."""
    torch = torch_gpu
    M, N, K = 128, 128, 128
    a = np.zeros((M, K), np.float16); a[np.arange(M), np.arange(M) % K] = 1
    w = (np.arange(N)[:, None] * 0.5 + np.arange(K)[None, :] * 0.001953125).astype(np.float16)
    out = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    zb = torch.zeros(N, dtype=torch.float32, device="cuda")
    da, dw = _dev(torch, a), _dev(torch, w)      # keep the device tensors alive across the launch
    binding.check(binding.lib().vitx_op_gemm(binding.F16, 3, da.data_ptr(), dw.data_ptr(), zb.data_ptr(), out.data_ptr(), M, N, K, None))
    _sync(torch)
    ref = a.astype(np.float32) @ w.astype(np.float32).T
    assert np.array_equal(out.cpu().numpy(), ref)


@pytest.mark.parametrize("n_img,N,H", [(2, 197, 3), (1, 197, 12), (3, 17, 2), (1, 257, 2), (1, 577, 2), (2, 50, 1)])
def test_attention_f16(binding, oracle, torch_gpu, n_img, N, H):
    """Fused attention vs the oracle's f32 attention on fp16-representable q,k,v
    (so the only differences are sum order and the unnormalised-P formulation)."""
    torch = torch_gpu
    D = H * 64
    rng = np.random.default_rng(n_img * 100 + N + H)
    qkv = (rng.standard_normal((n_img * N, 3 * D)) * 0.8).astype(np.float16)
    ref = oracle.attention(qkv.astype(np.float32), n_img, N, D, H, oracle.REF)
    dq = _dev(torch, qkv)
    out = torch.zeros((n_img * N, D), dtype=torch.float16, device="cuda")
    binding.check(binding.lib().vitx_op_attention(binding.F16, dq.data_ptr(), out.data_ptr(), n_img, N, D, H, None), "attention")
    _sync(torch)
    got = out.float().cpu().numpy()
    # outputs are convex combinations of v rows (|v| ~ 0.8): absolute tolerance 2 fp16 ulp of the value scale
    assert np.abs(got - ref).max() <= 3e-3
    assert np.abs(got - ref).mean() <= 3e-4


def test_attention_forced_spike(binding, oracle, torch_gpu):
    """One key dominates one query (softmax ~ one-hot) and padded keys must not leak."""
    torch = torch_gpu
    n_img, N, H = 1, 197, 1; D = 64
    rng = np.random.default_rng(5)
    qkv = (rng.standard_normal((N, 3 * D)) * 0.3).astype(np.float16)
    qkv[10, :64] = 4.0; qkv[150, 64:128] = 4.0          # q10 . k150 = 1024 -> *0.125 = 128
    ref = oracle.attention(qkv.astype(np.float32), n_img, N, D, H, oracle.REF)
    out = torch.zeros((N, D), dtype=torch.float16, device="cuda")
    dq = _dev(torch, qkv)
    binding.check(binding.lib().vitx_op_attention(binding.F16, dq.data_ptr(), out.data_ptr(), n_img, N, D, H, None))
    _sync(torch)
    got = out.float().cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got[10] - qkv[150, 128:].astype(np.float32)).max() <= 2e-3
    assert np.abs(got - ref).max() <= 3e-3


def test_softmax_matches_ggml_lut(binding, oracle, torch_gpu):
    torch = torch_gpu
    rng = np.random.default_rng(11)
    x = (rng.standard_normal((37, 1000)) * 4).astype(np.float32)
    ref = oracle.softmax_rows(x, lut=1)
    out = torch.zeros((37, 1000), dtype=torch.float32, device="cuda")
    dx = _dev(torch, x)
    binding.check(binding.lib().vitx_op_softmax(dx.data_ptr(), out.data_ptr(), 37, 1000, 1000, None))
    _sync(torch)
    got = out.cpu().numpy()
    assert np.abs(got - ref).max() <= 2e-6
    assert np.abs(got.sum(1) - 1).max() <= 1e-5


@pytest.mark.parametrize("shape", [(37, 53), (224, 224), (500, 31), (1, 1), (640, 480), (408, 612)])
def test_device_preprocess_is_bit_identical_to_host_and_oracle(binding, oracle, torch_gpu, shape):
    """vitx_preprocess_u8_device (HIP) == vitx_preprocess_u8 (host) == oracle restatement of vit_image_preprocess
    (vit.cpp:130-305), bit for bit: up- and down-scaling, both interpolations, 1x1 and extreme aspect ratios, and a
    batch of 3 same-sized images in one launch."""
    torch = torch_gpu
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    imgs = rng.integers(0, 256, size=(3, shape[0], shape[1], 3), dtype=np.uint8)
    d_in = _dev(torch, imgs)
    for mode, code in (("bicubic", binding.BICUBIC), ("bilinear", binding.BILINEAR)):
        for S in (64, 224, 384):
            d_out = torch.empty((3, S, S, 3), dtype=torch.float32, device="cuda")
            binding.preprocess_device(d_in.data_ptr(), 3, shape[1], shape[0], S, d_out.data_ptr(), code)
            _sync(torch)
            got = d_out.cpu().numpy()
            for b in range(3):
                assert np.array_equal(got[b], binding.preprocess(imgs[b], S, code)), (mode, S, b)
            assert np.array_equal(got[0], oracle.preprocess(imgs[0], S, mode)), (mode, S)
    with pytest.raises(binding.VitxError):
        binding.preprocess_device(d_in.data_ptr(), 3, shape[1], shape[0], 224, d_in.data_ptr(), 7)
