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


@pytest.mark.parametrize("name,n", [("vit_micro_hd32_patch16_64", 5), ("vit_micro_hd96_patch16_96", 3), ("vit_mini_hd80_patch14_112", 3)])
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
