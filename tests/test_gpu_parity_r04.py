"""Round-4 parity tests (VERDICT r03 "Next round" items 1-3).

  * the F16 parity mode multiplies f32-GRADE q, k, v, as the reference does (/root/reference/vit.cpp:848,858: ggml_mul_mat on f32 views):
    the QKV GEMM emits two fp16 planes (epilogue 5, hi = round(x), lo = round((x - hi) * 2048)) -- checked on every GEMM kernel family
    against float64 products -- and the precise streaming attention kernel forms hi.hi + (hi.lo + lo.hi) / 2048 -- checked against
    the oracle's REF mode on inputs that are NOT fp16-representable, at the level the oracle-rounds-too comparison had in r03;
  * the streaming two-pass kernel (attention_stream.hip) in its fast build, both operand types, any token count, against the oracle,
    with NaNs planted behind the tensor;
  * whole forwards: F16 contexts now run the precise kernel at every token count (ViT-tiny / ViT-B / 577 tokens / 785 tokens).
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RECORD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_r04.jsonl")


def _record(**kw):
    print("PARITY " + json.dumps(kw))
    try:
        os.makedirs(os.path.dirname(RECORD), exist_ok=True)
        with open(RECORD, "a") as f:
            f.write(json.dumps(kw) + "\n")
    except OSError:
        pass


def _dev(torch, a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


# ------------------------------------------------------------------------------------------------------------------
# EPI_BIAS_HILO: the QKV GEMM's two-plane output
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype_name", ["f16", "bf16"])
@pytest.mark.parametrize("kernel", [1, 945, 445, 245, 122, 0, 2])
def test_gemm_hilo_planes_every_kernel_family(binding, torch_gpu, kernel, dtype_name):
    """epilogue 5 on each GEMM family: hi + lo / 2048 reproduces acc + bias to f32 grade (float64 products of the same operands on
    sampled rows: f32 accumulation noise + 2^-20 relative for fp16 planes, 2^-14 for bf16), hi alone is the plain rounded output
    (what epilogue 0 stores, bit for bit), rows past M_real are stored in neither plane."""
    torch = torch_gpu
    dt = binding.F16 if dtype_name == "f16" else binding.BF16
    tdt = torch.float16 if dtype_name == "f16" else torch.bfloat16
    M, N, K = 33280, 768, 768
    M_real = M - 100
    g = torch.Generator(device="cuda").manual_seed(77 + kernel)
    A = (torch.randn((M, K), device="cuda", generator=g) * 0.7).to(tdt)
    W = (torch.randn((N, K), device="cuda", generator=g) * 0.05).to(tdt)
    bias = torch.randn(N, device="cuda", generator=g) * 0.1
    rng = np.random.default_rng(kernel)
    rows = np.unique(np.concatenate([np.arange(0, 4), np.arange(252, 260), np.arange(M_real - 160, M_real), rng.integers(0, M_real, 400)]))
    ridx = torch.from_numpy(rows).cuda()
    a64 = A[ridx].double().cpu().numpy(); w64 = W.double().cpu().numpy()
    v = a64 @ w64.T + bias.double().cpu().numpy()
    tol_acc = (np.abs(a64) @ np.abs(w64).T) * 2e-6 + 1e-6
    L = binding.lib()
    planes = torch.full((2, M, N), 7.0, dtype=tdt, device="cuda")
    plain = torch.full((M, N), 7.0, dtype=tdt, device="cuda")
    binding.check(L.vitx_op_gemm_ex(dt, 5, kernel, A.data_ptr(), W.data_ptr(), bias.data_ptr(), planes.data_ptr(), None, M, M_real, N, K, 0, None), f"gemm hilo kernel {kernel}")
    binding.check(L.vitx_op_gemm_ex(dt, 0, kernel, A.data_ptr(), W.data_ptr(), bias.data_ptr(), plain.data_ptr(), None, M, M_real, N, K, 0, None), f"gemm kernel {kernel}")
    torch.cuda.synchronize()
    assert torch.equal(planes[0], plain)                                     # the hi plane IS the ordinary rounded output
    got = planes[0][ridx].double().cpu().numpy() + planes[1][ridx].double().cpu().numpy() / 2048.0
    rel = 2.0 ** -20 if dtype_name == "f16" else 2.0 ** -14
    assert (np.abs(got - v) <= tol_acc + np.abs(v) * rel).all(), float((np.abs(got - v) - tol_acc).max())
    for p in range(2):
        tail = planes[p][M_real:].float()
        assert float(tail.min()) == 7.0 and float(tail.max()) == 7.0


# ------------------------------------------------------------------------------------------------------------------
# The precise attention kernel (F16 parity mode)
# ------------------------------------------------------------------------------------------------------------------
def _precise(binding, torch, qkv32, n_img, N, D, H):
    dq = _dev(torch, qkv32)
    out = torch.full((n_img * N, D), float("nan"), dtype=torch.float16, device="cuda")
    binding.check(binding.lib().vitx_op_attention_f32(dq.data_ptr(), out.data_ptr(), n_img, N, D, H, None), "vitx_op_attention_f32")
    torch.cuda.synchronize()
    return out.float().cpu().numpy()


def test_attention_precise_on_non_representable_inputs(binding, oracle, torch_gpu):
    """The r03 verdict's done-criterion: on f32 q, k, v that are NOT fp16-representable the parity mode's attention must sit at the level
    the engine had against an oracle that rounds too (3e-3 / 3e-4 max / mean), not at the 6e-3 / 6e-4 of fp16-rounded operands.  The
    r03 kernel (q, k, v rounded on upload) is measured on the same inputs for the record."""
    torch = torch_gpu
    n_img, N, H = 2, 197, 4; D = H * 64
    rng = np.random.default_rng(42)
    qkv32 = (rng.standard_normal((n_img * N, 3 * D)) * 0.8).astype(np.float32)
    assert (qkv32.astype(np.float16).astype(np.float32) != qkv32).mean() > 0.99
    ref = oracle.attention(qkv32, n_img, N, D, H, oracle.REF)
    got = _precise(binding, torch, qkv32, n_img, N, D, H)
    d = np.abs(got - ref)
    dq16 = _dev(torch, qkv32.astype(np.float16))
    out16 = torch.zeros((n_img * N, D), dtype=torch.float16, device="cuda")
    binding.check(binding.lib().vitx_op_attention(binding.F16, dq16.data_ptr(), out16.data_ptr(), n_img, N, D, H, None))
    torch.cuda.synchronize()
    d16 = np.abs(out16.float().cpu().numpy() - ref)
    _record(test="attention_non_representable", precise_max=float(d.max()), precise_mean=float(d.mean()), rounded_operands_max=float(d16.max()), rounded_operands_mean=float(d16.mean()))
    assert np.isfinite(got).all()
    assert d.max() <= 3e-3 and d.mean() <= 3e-4
    assert d.mean() < 0.75 * d16.mean()                      # and visibly better than rounding the operands


@pytest.mark.parametrize("n_img,N,H", [(2, 197, 3), (1, 577, 2), (3, 17, 2), (1, 1, 1), (2, 64, 1), (2, 65, 2), (1, 128, 1), (1, 129, 3), (1, 257, 2), (1, 785, 1), (5, 33, 1), (1, 1025, 1)])
def test_attention_precise_any_token_count(binding, oracle, torch_gpu, n_img, N, H):
    """Every token count, ragged last chunks and last tiles included, against the reference semantics on generic f32 inputs."""
    torch = torch_gpu
    D = H * 64
    rng = np.random.default_rng(n_img * 1000 + N * 7 + H)
    qkv32 = (rng.standard_normal((n_img * N, 3 * D)) * 0.8).astype(np.float32)
    ref = oracle.attention(qkv32, n_img, N, D, H, oracle.REF)
    got = _precise(binding, torch, qkv32, n_img, N, D, H)
    assert np.isfinite(got).all()
    d = np.abs(got - ref)
    assert d.max() <= 3e-3 and d.mean() <= 3e-4, (float(d.max()), float(d.mean()))


def test_attention_precise_forced_spike_and_scale(binding, oracle, torch_gpu):
    """One key dominates one query (the softmax is one-hot there), large-magnitude rows, and tiny rows whose lo plane would be subnormal
    without the 2048 scale: all against the reference semantics."""
    torch = torch_gpu
    n_img, N, H = 1, 197, 1; D = 64
    rng = np.random.default_rng(5)
    qkv = (rng.standard_normal((N, 3 * D)) * 0.3).astype(np.float32)
    qkv[10, :64] = 4.0003; qkv[150, 64:128] = 3.9997              # q10 . k150 ~ 1024 -> * 0.125 = 128
    qkv[20:30] *= 1e-3                                             # tiny q, k, v rows
    qkv[40:44, 128:] *= 40.0                                       # large v rows
    ref = oracle.attention(qkv, n_img, N, D, H, oracle.REF)
    got = _precise(binding, torch, qkv, n_img, N, D, H)
    assert np.isfinite(got).all()
    assert np.abs(got[10] - qkv[150, 128:]).max() <= 2e-3
    assert (np.abs(got - ref) <= 3e-3 + np.abs(ref) * 2e-3).all()


# ------------------------------------------------------------------------------------------------------------------
# The streaming kernel's fast build (both operand types)
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_img,N,H", [(1, 577, 2), (2, 197, 3), (1, 785, 2), (2, 300, 1), (3, 225, 2), (1, 1025, 1), (2, 129, 2), (9, 65, 1), (2, 64, 3), (1, 1, 1), (4, 33, 2), (1, 31, 1)])
def test_attention_stream_fast_any_token_count(binding, oracle, torch_gpu, n_img, N, H):
    torch = torch_gpu
    D = H * 64
    rng = np.random.default_rng(n_img * 100 + N + H)
    base = (rng.standard_normal((n_img * N, 3 * D)) * 0.8).astype(np.float32)
    for dt, tdt, mode, tmax, tmean in ((binding.F16, torch.float16, oracle.REF, 3e-3, 3e-4), (binding.BF16, torch.bfloat16, oracle.GPU_BF16, 2.5e-2, 2.5e-3)):
        qkv = torch.from_numpy(base).to(tdt)
        ref = oracle.attention(qkv.float().numpy(), n_img, N, D, H, mode)
        dq = qkv.cuda()
        out = torch.full((n_img * N, D), float("nan"), dtype=tdt, device="cuda")
        binding.check(binding.lib().vitx_op_attention_ex(dt, binding.ATTN_STREAM, dq.data_ptr(), out.data_ptr(), n_img, N, D, H, None), "attention stream")
        torch.cuda.synchronize()
        got = out.float().cpu().numpy()
        assert np.isfinite(got).all()
        d = np.abs(got - ref)
        assert d.max() <= tmax and d.mean() <= tmean, (float(d.max()), float(d.mean()))


@pytest.mark.parametrize("N", [577, 70, 197])
def test_attention_stream_never_reads_past_the_tensor(binding, torch_gpu, N):
    """Whole 64-key chunks are streamed: keys past N of the LAST image lie behind the qkv tensor.  The buffer descriptor ends at the tensor,
    so those loads return 0; NaNs planted right behind it must not reach the result (both builds)."""
    torch = torch_gpu
    n_img, H = 3, 2; D = H * 64
    g = torch.Generator(device="cuda").manual_seed(7 * N)
    rows = n_img * N
    big = torch.full((rows + 128, 3 * D), float("nan"), device="cuda", dtype=torch.bfloat16)
    big[:rows] = (torch.randn((rows, 3 * D), device="cuda", generator=g) * 0.8).to(torch.bfloat16)
    clean = big[:rows].clone()
    outs = []
    for src in (big, clean):
        out = torch.zeros((rows, D), dtype=torch.bfloat16, device="cuda")
        binding.check(binding.lib().vitx_op_attention_ex(binding.BF16, binding.ATTN_STREAM, src.data_ptr(), out.data_ptr(), n_img, N, D, H, None))
        torch.cuda.synchronize(); outs.append(out)
    assert torch.isfinite(outs[0].float()).all() and torch.equal(outs[0], outs[1])


# ------------------------------------------------------------------------------------------------------------------
# Whole forwards in the parity mode
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,n", [("vit_tiny_patch16_224", 5), ("vit_base_patch16_224", 3), ("vit_micro_patch8_224", 2)])
def test_forward_parity_mode_and_its_fast_attention_option(pkg, binding, oracle, torch_gpu, name, n):
    """F16 contexts run the precise attention; `f16_fast_attention` restores the r03 kernels.  Both within 1e-3 of the reference on the
    x4 head; the measured deltas go to the record."""
    path = pkg.synth.cached_synthetic(name, head_scale=4.0)
    hp = pkg.synth.hparams_for(name)
    imgs = pkg.synth.normalize_u8(pkg.synth.synthetic_images_u8(n, hp.img_size, seed=123))
    _, ref = oracle.OracleModel(path).forward(imgs, oracle.REF)
    model = binding.Model(path)
    res = {}
    for label, opts in (("precise", {}), ("fast", {"f16_fast_attention": 1})):
        ctx = binding.Context(model, device=0, max_batch=n, dtype=binding.F16, **opts)
        res[label] = ctx.forward(imgs); ctx.close()
    model.close()
    dp, df = float(np.abs(res["precise"] - ref).max()), float(np.abs(res["fast"] - ref).max())
    _record(test="forward_parity_mode", model=name, images=n, precise_max_dprob=dp, fast_attention_max_dprob=df)
    assert dp <= 1e-3 and df <= 1e-3
    assert (res["precise"].argmax(1) == ref.argmax(1)).all()


def test_attention_beyond_one_launch_window_is_chunked_not_rerouted(binding, torch_gpu):
    """More (image, head) items than the 32-bit buffer offsets of ONE persistent launch cover (~4.4 k ViT-B images): the launcher cuts the batch
    into several launches of the SAME kernel (r03 advisor: it used to hand the whole batch to another kernel family, whose f32 sums are grouped
    differently -- an image's result must not depend on the batch it arrives in).  Images on both sides of the cut equal their stand-alone result."""
    torch = torch_gpu
    N, H = 197, 12; D = H * 64
    n_img = 4500                                          # 4500 x 197 x 2304 x 2 B = 4.08 GB > 0xf0000000
    assert n_img * N * 3 * D * 2 > 0xf0000000
    g = torch.Generator(device="cuda").manual_seed(4500)
    qkv = torch.empty((n_img * N, 3 * D), dtype=torch.bfloat16, device="cuda")
    for i0 in range(0, n_img, 500):                       # filled in pieces: no 8 GB f32 temporary
        qkv[i0 * N:(i0 + 500) * N] = (torch.randn((500 * N, 3 * D), device="cuda", generator=g) * 0.8).to(torch.bfloat16)
    out = torch.zeros((n_img * N, D), dtype=torch.bfloat16, device="cuda")
    binding.check(binding.lib().vitx_op_attention(binding.BF16, qkv.data_ptr(), out.data_ptr(), n_img, N, D, H, None), "attention")
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    cut = (0xf0000000 - 1) // (N * 3 * D * 2)             # images of the first launch
    for first, cnt in ((0, 40), (cut - 20, 40), (n_img - 40, 40)):
        sub = qkv[first * N:(first + cnt) * N].contiguous()
        o2 = torch.zeros((cnt * N, D), dtype=torch.bfloat16, device="cuda")
        binding.check(binding.lib().vitx_op_attention(binding.BF16, sub.data_ptr(), o2.data_ptr(), cnt, N, D, H, None), "attention")
        torch.cuda.synchronize()
        assert torch.equal(out[first * N:(first + cnt) * N], o2), first


@pytest.mark.parametrize("big,dtype_name,streams", [(5000, "bf16", 2), (5000, "f16", 2), (7001, "bf16", 2), (3326, "bf16", 1), (2217, "f16", 1)])
def test_forward_of_thousands_of_images_equals_the_batch_256_result(pkg, binding, torch_gpu, big, dtype_name, streams):
    """288 GB of HBM invite batches the 32-bit byte offsets of the buffer instructions do not cover: QKV of a 2500-image sub-batch is 2.3 GB (r04: the
    persistent attention kernel's item offset was a signed int -- wrong results from 2366 images per launch on), the MLP hidden tensor of a 3538-image
    sub-batch 4.3 GB (garbage at batch 10 000).  A pass of the kernels is now bounded by vitx_ctx_create_ex and larger batches run as several passes:
    every image of a batch of thousands must get the bits it gets in a batch of 256.  streams = 1 at 3326 / 2217 images: ONE sub-batch that fills the
    window to the last row block (hidden tensor 4.02 GB, the two F16 QKV planes 4.02 GB)."""
    torch = torch_gpu
    name = "vit_base_patch16_224"
    dt = binding.BF16 if dtype_name == "bf16" else binding.F16
    path = pkg.synth.cached_synthetic(name, head_scale=8.0); hp = pkg.synth.hparams_for(name)
    model = binding.Model(path)
    base = torch.randn((256, hp.img_size, hp.img_size, 3), device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    c0 = binding.Context(model, max_batch=256, dtype=dt)
    p0 = torch.empty((256, hp.num_classes), device="cuda")
    c0.forward_device(base.data_ptr(), 256, p0.data_ptr(), 0, 0); c0.synchronize(); c0.close()
    reps = (big + 255) // 256
    imgs = base.repeat(reps, 1, 1, 1)[:big].contiguous()
    c1 = binding.Context(model, max_batch=big, dtype=dt, streams=streams)
    p1 = torch.empty((big, hp.num_classes), device="cuda")
    c1.forward_device(imgs.data_ptr(), big, p1.data_ptr(), 0, 0); c1.synchronize()
    assert torch.equal(p1, p0.repeat(reps, 1)[:big])
    c1.close(); model.close()


@pytest.mark.parametrize("n", [1, 37])
def test_forward_with_21843_classes(pkg, binding, oracle, torch_gpu, n):
    """timm's *_in21k checkpoints carry a 21 843-class head (the reference's converter writes whatever the checkpoint has): the classifier GEMM runs
    86 column tiles with a ragged last one and the class softmax spans 21 843 columns -- against the oracle in both operand types."""
    name = "vit_micro_c21843_patch16_64"
    path = pkg.synth.cached_synthetic(name, head_scale=8.0)
    imgs = pkg.synth.normalize_u8(pkg.synth.synthetic_images_u8(n, 64, seed=21))
    om = oracle.OracleModel(path)
    _, rp = om.forward(imgs, oracle.REF)
    _, bp = om.forward(imgs, oracle.GPU_BF16)
    model = binding.Model(path)
    assert model.num_classes == 21843 and model.label(21842) is not None
    for dt, ref, tol in ((binding.F16, rp, 1e-3), (binding.BF16, bp, 6e-3)):
        ctx = binding.Context(model, max_batch=n, dtype=dt)
        probs = ctx.forward(imgs); ctx.close()
        assert probs.shape == (n, 21843) and np.isfinite(probs).all()
        assert np.abs(probs.sum(1) - 1).max() < 1e-4
        assert np.abs(probs - ref).max() <= tol
        picked = ref[np.arange(n), probs.argmax(1)]              # the class the engine ranks first is (within the tolerance) the reference's best
        assert (picked >= ref.max(1) - 2 * tol).all()
    model.close()
