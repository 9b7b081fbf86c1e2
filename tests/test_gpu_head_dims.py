"""Head dimensions other than 64 (attention_generic_kernel): /root/reference/vit.cpp:826-866 is generic in n_enc_head_dim and the
reference's converter writes any timm ViT (ViT-H/14: 1280 wide, 16 heads -> 80).  Kernel-level parity against the oracle for head dims
8 .. 128 and token counts around the tile edges, the scale 1 / sqrt(head_dim) where it is not a power of two, and whole forwards of
synthetic models with head dims 32, 96 and 80 (patch 14) against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [(2, 197, 2, 80), (1, 257, 3, 80), (3, 17, 2, 32), (1, 50, 4, 96), (2, 65, 1, 128), (1, 1, 2, 8), (2, 197, 1, 72), (1, 300, 2, 40), (1, 64, 2, 16), (2, 33, 3, 120)]


@pytest.mark.parametrize("n_img,N,H,DH", CASES)
def test_attention_other_head_dims_f16(binding, oracle, torch_gpu, n_img, N, H, DH):
    torch = torch_gpu
    D = H * DH
    rng = np.random.default_rng(n_img * 1000 + N * 7 + DH)
    qkv = (rng.standard_normal((n_img * N, 3 * D)) * 0.8).astype(np.float16)
    ref = oracle.attention(qkv.astype(np.float32), n_img, N, D, H, oracle.REF)
    dq = torch.from_numpy(qkv).cuda()
    out = torch.full((n_img * N, D), float("nan"), dtype=torch.float16, device="cuda")
    binding.check(binding.lib().vitx_op_attention(binding.F16, dq.data_ptr(), out.data_ptr(), n_img, N, D, H, None), "attention")
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    assert np.isfinite(got).all()                     # every output element written, nothing from a padded key or dim
    assert np.abs(got - ref).max() <= 3e-3 and np.abs(got - ref).mean() <= 3e-4


@pytest.mark.parametrize("n_img,N,H,DH", [(2, 197, 2, 80), (1, 50, 4, 96), (3, 17, 2, 32)])
def test_attention_other_head_dims_bf16(binding, oracle, torch_gpu, n_img, N, H, DH):
    torch = torch_gpu
    D = H * DH
    rng = np.random.default_rng(n_img * 1000 + N * 7 + DH)
    qkv = torch.from_numpy((rng.standard_normal((n_img * N, 3 * D)) * 0.8).astype(np.float32)).to(torch.bfloat16)
    ref = oracle.attention(qkv.float().numpy(), n_img, N, D, H, oracle.GPU_BF16)
    dq = qkv.cuda()
    out = torch.zeros((n_img * N, D), dtype=torch.bfloat16, device="cuda")
    binding.check(binding.lib().vitx_op_attention(binding.BF16, dq.data_ptr(), out.data_ptr(), n_img, N, D, H, None), "attention")
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    assert np.abs(got - ref).max() <= 2.5e-2 and np.abs(got - ref).mean() <= 2.5e-3


def test_attention_poisoned_neighbours(binding, torch_gpu):
    """Dims past head_dim of a head's slice belong to the NEXT head, keys past N to the next image (or lie behind the tensor): NaNs
    planted in every other head and image must not reach this head's output."""
    torch = torch_gpu
    n_img, N, H, DH = 2, 37, 3, 80; D = H * DH
    g = torch.Generator(device="cuda").manual_seed(3)
    clean = (torch.randn((n_img * N, 3 * D), device="cuda", generator=g) * 0.8).to(torch.float16)
    outs = []
    for poison in (False, True):
        q = clean.clone()
        if poison:
            v = q.view(n_img, N, 3, H, DH)
            v[1] = float("nan"); v[0, :, :, 0] = float("nan"); v[0, :, :, 2] = float("nan")       # only image 0 / head 1 stays clean
        out = torch.zeros((n_img * N, D), dtype=torch.float16, device="cuda")
        binding.check(binding.lib().vitx_op_attention(binding.F16, q.data_ptr(), out.data_ptr(), n_img, N, D, H, None))
        torch.cuda.synchronize()
        outs.append(out.view(n_img, N, H, DH)[0, :, 1].clone())
    assert torch.isfinite(outs[1]).all() and torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("D", [320, 448, 576, 640, 896, 1152, 1280, 1408, 1536, 1664, 2048])
def test_layernorm_other_widths(binding, oracle, torch_gpu, D):
    """Hidden sizes of timm ViTs beyond tiny / small / base / large (SO400M 1152, ViT-H 1280, ViT-g 1408, ViT-G 1664, ...)."""
    torch = torch_gpu
    M = 37
    rng = np.random.default_rng(D)
    x = (rng.standard_normal((M, D)) * 0.7 + 0.1).astype(np.float32)
    w = (1 + 0.02 * rng.standard_normal(D)).astype(np.float32)
    b = (0.02 * rng.standard_normal(D)).astype(np.float32)
    ref = oracle.layernorm(x, w, b, 1e-6)
    dx, dw, db = (torch.from_numpy(a).cuda() for a in (x, w, b))
    y = torch.empty((M, D), dtype=torch.float16, device="cuda")
    binding.check(binding.lib().vitx_op_layernorm(binding.F16, dx.data_ptr(), dw.data_ptr(), db.data_ptr(), y.data_ptr(), M, D, 1e-6, None))
    torch.cuda.synchronize()
    got = y.float().cpu().numpy()
    ref16 = ref.astype(np.float16).astype(np.float32)
    assert np.abs(got - ref16).max() <= np.abs(ref).max() * 2.0 ** -10 and (got != ref16).mean() < 0.01


@pytest.mark.parametrize("name,n", [("vit_micro_hd32_patch16_64", 5), ("vit_micro_hd96_patch16_96", 3), ("vit_mini_hd80_patch14_112", 3), ("vit_mini_hd72_patch14_112", 3)])
def test_forward_other_head_dims_vs_oracle(pkg, binding, oracle, torch_gpu, name, n):
    torch = torch_gpu
    path = pkg.synth.cached_synthetic(name, head_scale=4.0)
    S = pkg.synth.CONFIGS[name][5]
    imgs = pkg.synth.normalize_u8(pkg.synth.synthetic_images_u8(n, S))
    m = binding.Model(path)
    ref_logits, ref_probs = oracle.OracleModel(path).forward(imgs, oracle.REF)
    for dt, tol in ((binding.F16, 1e-3), (binding.BF16, 2e-2)):
        ctx = binding.Context(m, 0, n, dt)
        probs = ctx.forward(imgs)
        ctx.close()
        assert np.isfinite(probs).all() and np.abs(probs.sum(1) - 1).max() < 1e-4
        assert np.abs(probs - ref_probs).max() <= tol
    m.close()


def test_forward_ragged_width_large_batch_equals_small_batch(pkg, binding, torch_gpu):
    """1152 columns are 4.5 of the 256-column tiles: at a batch that takes the wide persistent GEMMs the last column tile is ragged (the
    per-element epilogue path) and the LayerNorms are not fused (4.5 tiles); an image's probabilities must not depend on the batch."""
    name = "vit_mini_hd72_patch14_112"
    path = pkg.synth.cached_synthetic(name, head_scale=4.0)
    imgs = pkg.synth.normalize_u8(pkg.synth.synthetic_images_u8(150, 112, seed=5))
    m = binding.Model(path)
    for dt in (binding.F16, binding.BF16):
        big = binding.Context(m, 0, 150, dt); pb = big.forward(imgs); big.close()
        small = binding.Context(m, 0, 3, dt); ps = small.forward(imgs[:3]); small.close()
        assert np.isfinite(pb).all() and np.array_equal(pb[:3], ps)
    m.close()
