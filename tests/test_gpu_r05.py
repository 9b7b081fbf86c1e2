"""Round-5 tests.

  * the F16 parity mode's attention at 193..224 tokens is now a PERSISTENT kernel (attention_precise_kernel: one workgroup per CU walks the
    (image, head) items, a ring of 64-row chunks runs across items): many more items than CUs, ragged batches, against the oracle and
    against the same image computed alone (bits);
  * r04 advisor: the co-residency hazard (a foreign wave's DPP sums going wrong beside a back-to-back MFMA stream) -- LayerNorm bit for bit
    beside EVERY MFMA kernel family, not only the one that showed it;
  * r04 advisor: the fused LayerNorm's fall-back budget re-arms after a cool-down;
  * the bracket calibration bench.py subtracts, and the argument checks of the test-only attention entry points.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(torch, a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


def _precise(binding, torch, qkv32, n_img, N, D, H):
    dq = _dev(torch, qkv32)
    out = torch.full((n_img * N, D), float("nan"), dtype=torch.float16, device="cuda")
    binding.check(binding.lib().vitx_op_attention_f32(dq.data_ptr(), out.data_ptr(), n_img, N, D, H, None), "vitx_op_attention_f32")
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("n_img,N,H", [(45, 197, 12), (23, 208, 12), (9, 224, 16), (301, 197, 1), (2, 193, 3), (30, 209, 12)])
def test_precise_attention_persistent_many_items(binding, oracle, torch_gpu, n_img, N, H):
    """More items than CUs (540 items on 256 workgroups: three items a workgroup and a ragged tail), both register builds (13 / 14 score tiles),
    fewer items than CUs, one head: a sample of images against the reference semantics, and EVERY image against itself computed alone --
    the ring that runs across items must not leak one item's rows into the next."""
    torch = torch_gpu
    D = H * 64
    rng = np.random.default_rng(n_img * 1000 + N * 7 + H)
    qkv32 = (rng.standard_normal((n_img * N, 3 * D)) * 0.8).astype(np.float32)
    got = _precise(binding, torch, qkv32, n_img, N, D, H)
    assert torch.isfinite(got.float()).all()
    pick = sorted({0, n_img - 1, n_img // 2, min(n_img - 1, 21)})
    for b in pick:
        one = qkv32[b * N:(b + 1) * N]
        ref = oracle.attention(one, 1, N, D, H, oracle.REF)
        d = np.abs(got[b * N:(b + 1) * N].float().cpu().numpy() - ref)
        assert d.max() <= 3e-3 and d.mean() <= 3e-4, (b, float(d.max()), float(d.mean()))
    for b in sorted(set(pick) | {1, n_img - 2} if n_img > 2 else set(pick)):
        alone = _precise(binding, torch, qkv32[b * N:(b + 1) * N], 1, N, D, H)
        assert torch.equal(alone, got[b * N:(b + 1) * N]), b


def test_precise_attention_persistent_is_repeatable_and_ignores_what_lies_behind_the_planes(binding, torch_gpu):
    """Tail chunks are fetched through a descriptor that ends at the last possible token row of the item: NaNs behind the last image of either
    plane (and between the planes) must not reach the result; 5 runs give the same bits."""
    torch = torch_gpu
    n_img, N, H = 35, 197, 12; D = H * 64
    rows = n_img * N
    g = torch.Generator(device="cuda").manual_seed(77)
    x = (torch.randn((rows, 3 * D), device="cuda", generator=g) * 0.8)
    hi = x.to(torch.float16)
    lo = ((x - hi.float()) * 2048.0).to(torch.float16)
    pad = 64                                              # rows of NaN behind each plane
    buf = torch.full((2 * (rows + pad), 3 * D), float("nan"), dtype=torch.float16, device="cuda")
    buf[:rows] = hi; buf[rows + pad:2 * rows + pad] = lo
    lo_off = (rows + pad) * 3 * D
    outs = []
    for _ in range(5):
        out = torch.full((rows, D), float("nan"), dtype=torch.float16, device="cuda")
        binding.check(binding.lib().vitx_op_attention_planes(buf.data_ptr(), lo_off, out.data_ptr(), n_img, N, D, H, None), "vitx_op_attention_planes")
        torch.cuda.synchronize(); outs.append(out)
    assert torch.isfinite(outs[0].float()).all()
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    clean = torch.zeros_like(buf); clean[:rows] = hi; clean[rows + pad:2 * rows + pad] = lo
    out2 = torch.empty_like(outs[0])
    binding.check(binding.lib().vitx_op_attention_planes(clean.data_ptr(), lo_off, out2.data_ptr(), n_img, N, D, H, None))
    torch.cuda.synchronize()
    assert torch.equal(out2, outs[0])


def test_attention_planes_argument_checks(binding, torch_gpu):
    torch = torch_gpu
    n_img, N, H = 1, 197, 1; D = 64
    buf = torch.zeros((2 * N, 3 * D), dtype=torch.float16, device="cuda"); out = torch.zeros((N, D), dtype=torch.float16, device="cuda")
    L = binding.lib()
    n = N * 3 * D
    assert L.vitx_op_attention_planes(buf.data_ptr(), n, out.data_ptr(), n_img, N, D, H, None) == 0
    for bad in (n - 4, n + 2, 0, -n):                     # overlapping the hi plane, not a multiple of 4 elements, non-positive
        assert L.vitx_op_attention_planes(buf.data_ptr(), bad, out.data_ptr(), n_img, N, D, H, None) != 0
        assert b"lo_off" in L.vitx_last_error() or b"invalid" in L.vitx_last_error()
    torch.cuda.synchronize()


FAMILIES = ["gemm_pp", "gemm_945", "gemm_445", "gemm_245", "gemm_122", "attn_single", "attn_flow", "attn_persist", "attn_stream", "attn_generic", "attn_precise"]


@pytest.mark.parametrize("family", FAMILIES)
def test_layernorm_bits_beside_every_mfma_kernel_family(binding, torch_gpu, family):
    """r04: waves of ANOTHER kernel (LayerNorm: DPP / permute reductions) computed wrong sums on SIMDs that also ran the streaming attention
    kernel's back-to-back 16x16x32 MFMAs (profiles/r04/coresidency_layernorm.txt).  That kernel claims its SIMDs' whole register file since;
    the advisor asked for the same check beside every other MFMA family: LayerNorm of a fixed input on one stream while the family loops
    on another, 10 runs, every output bit for bit the LayerNorm computed alone."""
    torch = torch_gpu
    L = binding.lib()
    g = torch.Generator(device="cuda").manual_seed(5)
    rows, Dl = 28 * 577, 1024
    x = torch.randn((rows, Dl), device="cuda", generator=g); w = torch.randn((Dl,), device="cuda", generator=g); b = torch.randn((Dl,), device="cuda", generator=g)
    y0 = torch.empty((rows, Dl), dtype=torch.float16, device="cuda")
    binding.check(L.vitx_op_layernorm(binding.F16, x.data_ptr(), w.data_ptr(), b.data_ptr(), y0.data_ptr(), rows, Dl, 1e-6, None))
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    if family.startswith("gemm"):
        M, N, K = 12800, 2304, 768
        A = (torch.randn((M, K), device="cuda", generator=g) * 0.5).to(torch.bfloat16); W = (torch.randn((N, K), device="cuda", generator=g) * 0.05).to(torch.bfloat16)
        bias = torch.randn((N,), device="cuda", generator=g); out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        kern = 1 if family == "gemm_pp" else int(family.split("_")[1])
        def co():
            binding.check(L.vitx_op_gemm_ex(binding.BF16, 0, kern, A.data_ptr(), W.data_ptr(), bias.data_ptr(), out.data_ptr(), None, M, M, N, K, 0, sa.cuda_stream), family)
    else:
        which = family.split("_")[1]
        n_img, N, H, hd = {"single": (64, 197, 12, 64), "flow": (42, 577, 16, 64), "persist": (64, 197, 12, 64), "stream": (42, 577, 16, 64),
                           "generic": (48, 197, 12, 32), "precise": (64, 197, 12, 64)}[which]
        D = H * hd
        if which == "precise":
            planes = (torch.randn((2 * n_img * N, 3 * D), device="cuda", generator=g) * 0.5).to(torch.float16)
            out = torch.empty((n_img * N, D), dtype=torch.float16, device="cuda")
            def co():
                binding.check(L.vitx_op_attention_planes(planes.data_ptr(), n_img * N * 3 * D, out.data_ptr(), n_img, N, D, H, sa.cuda_stream), family)
        else:
            qkv = (torch.randn((n_img * N, 3 * D), device="cuda", generator=g) * 0.5).to(torch.bfloat16)
            out = torch.empty((n_img * N, D), dtype=torch.bfloat16, device="cuda")
            kid = {"single": binding.ATTN_SINGLE, "flow": binding.ATTN_FLOW, "persist": binding.ATTN_PERSIST, "stream": binding.ATTN_STREAM, "generic": binding.ATTN_AUTO}[which]
            def co():
                binding.check(L.vitx_op_attention_ex(binding.BF16, kid, qkv.data_ptr(), out.data_ptr(), n_img, N, D, H, sa.cuda_stream), family)
    bad_runs = 0
    for run in range(10):
        ys = [torch.empty_like(y0) for _ in range(4)]
        for _ in range(6): co()
        for y in ys:
            binding.check(L.vitx_op_layernorm(binding.F16, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), rows, Dl, 1e-6, sb.cuda_stream))
            co()
        torch.cuda.synchronize()
        if not all(torch.equal(y, y0) for y in ys):
            bad_runs += 1
    assert bad_runs == 0, f"LayerNorm beside {family}: {bad_runs} of 10 runs differ from the LayerNorm computed alone"


def test_ln_fallback_budget_rearms_after_its_cooldown(pkg, binding, torch_gpu):
    """r04 advisor: one window of contention must not cost the fused path for the rest of the context's life.  Forced, counting fall-backs
    (ln_test 1 | 4) trip the budget within two windows; 256 forwards later the context fuses again (and, the faults still being injected, trips
    again, now with a cool-down of 512).  The probabilities never change."""
    torch = torch_gpu
    name, n = "vit_base_patch16_224", 256
    path = pkg.synth.cached_synthetic(name, head_scale=8.0)
    hp = pkg.synth.hparams_for(name)
    imgs = torch.randn((n, hp.img_size, hp.img_size, 3), device="cuda", generator=torch.Generator(device="cuda").manual_seed(12))
    model = binding.Model(path)
    ctx = binding.Context(model, max_batch=n, dtype=binding.BF16, ln_test=binding.LN_TEST_KEY | 5)
    ref = torch.empty((n, hp.num_classes), device="cuda"); probs = torch.empty_like(ref)
    ctx.forward_device(imgs.data_ptr(), n, ref.data_ptr(), 0, 0); ctx.synchronize()
    states = [ctx.ln_fusion_active()]
    for rep in range(340):
        ctx.forward_device(imgs.data_ptr(), n, probs.data_ptr(), 0, 0)
        states.append(ctx.ln_fusion_active())
    ctx.synchronize()
    assert torch.equal(probs, ref)
    first_off = states.index(-1)
    assert first_off <= 36, first_off
    back_on = next((i for i in range(first_off, len(states)) if states[i] == 1), None)
    assert back_on is not None and 250 <= back_on - first_off <= 262, (first_off, back_on)
    assert -1 in states[back_on:], "the faults are still injected: the budget must trip again"
    ctx.close(); model.close()


def test_profile_bracket_cost_is_measured(pkg, binding, torch_gpu):
    """vitx_profile_bracket_us: what a HIP-event bracket adds to one launch (bench.py subtracts it from the profiled step's intervals)."""
    path = pkg.synth.cached_synthetic("vit_tiny_patch16_224", head_scale=4.0)
    model = binding.Model(path); ctx = binding.Context(model, max_batch=4, dtype=binding.F16)
    us = [ctx.profile_bracket_us() for _ in range(3)]
    assert all(0.2 < u < 25.0 for u in us), us
    ctx.close(); model.close()
