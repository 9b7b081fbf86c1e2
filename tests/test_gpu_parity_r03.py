"""Round-3 parity tests (VERDICT r02 "Next round" item 1): the benchmarked configurations that had no oracle comparison.

  * BASELINE.json config 5's actual model: ViT-B/16 from a q4_0 file at batch 256, exactly as `bench.py --ftype q4_0` runs it, against
    the oracle on the same dequantised weights (the engine's semantics) AND against the reference's full semantics -- q4_0 x q8_0 block
    dots with q8_0-quantised activations (/root/reference/vit.cpp:384-414, quantize.cpp:271-303); the measured deviation is printed;
  * the peaked fixtures SURVEY.md 8(d) asks for (head x8 and x10): the measured |dp| is PRINTED next to the oracle's own
    summation-order noise instead of hiding behind a max();
  * vitx_ctx_options: every scheduling option leaves the results bit-identical (the library reads no environment variable).
"""
import dataclasses
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from conftest import boundary_rows      # rows on both sides of the context's actual sub-batch boundary
RECORD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_r03.jsonl")


def _record(**kw):
    """Measured deviations go to stdout (pytest -s / the captured log) and to gpurun_out/parity_r03.jsonl for profiles/."""
    print("PARITY " + json.dumps(kw))
    try:
        os.makedirs(os.path.dirname(RECORD), exist_ok=True)
        with open(RECORD, "a") as f:
            f.write(json.dumps(kw) + "\n")
    except OSError:
        pass


def _bench_like(binding, torch, path, imgs, dtype, **options):
    """bench.py's call: device-resident images, vitx_forward_device on a non-default torch stream."""
    n = imgs.shape[0]
    model = binding.Model(path)
    ctx = binding.Context(model, device=0, max_batch=n, dtype=dtype, **options)
    d_imgs = torch.from_numpy(imgs).cuda()
    d_probs = torch.empty((n, model.num_classes), device="cuda"); d_logits = torch.empty_like(d_probs)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ctx.forward_device(d_imgs.data_ptr(), n, d_probs.data_ptr(), d_logits.data_ptr(), st.cuda_stream)
    st.synchronize()
    probs, logits = d_probs.cpu().numpy(), d_logits.cpu().numpy()
    ctx.close(); model.close()
    return probs, logits


def test_forward_base_bs256_q4_0_vs_oracle(pkg, binding, oracle, torch_gpu):
    """BASELINE config 5 on its own model.  The file is what `bench.py --ftype q4_0` loads (synth writes the reference's q4_0 blocks)."""
    name = "vit_base_patch16_224"
    path = pkg.synth.cached_synthetic(name, ftype=2, head_scale=4.0)
    imgs = pkg.synth.normalize_u8(pkg.synth.synthetic_images_u8(256, 224, seed=2025))
    om = oracle.OracleModel(path)
    CHECK_IDS = boundary_rows(binding, path, 256, binding.F16)
    assert CHECK_IDS == boundary_rows(binding, path, 256, binding.BF16)
    sub = imgs[CHECK_IDS]
    _, p_deq = om.forward(sub, dataclasses.replace(oracle.REF, quant_act=0))      # dequantised weights x fp16 activations: the engine's semantics
    _, p_ggml = om.forward(sub, oracle.REF)                                       # q4_0 x q8_0 integer block dots: the reference's semantics
    _, p_bf = om.forward(sub, oracle.GPU_BF16)
    model_gap = float(np.abs(p_deq - p_ggml).max())                               # what ggml's activation quantisation itself moves
    self_noise = float(np.abs(om.forward(sub, dataclasses.replace(oracle.REF, dot_exact=1))[1] - p_ggml).max())      # ... and what it moves against ITSELF (summation order only)
    for dt, dname in ((binding.F16, "f16"), (binding.BF16, "bf16")):
        probs, _ = _bench_like(binding, torch_gpu, path, imgs, dt)
        assert np.isfinite(probs).all() and np.abs(probs.sum(1) - 1).max() < 1e-4
        got = probs[CHECK_IDS]
        d_deq, d_ggml, d_bf = float(np.abs(got - p_deq).max()), float(np.abs(got - p_ggml).max()), float(np.abs(got - p_bf).max())
        _record(test="base_bs256_q4_0", dtype=dname, max_dprob_vs_dequantised_oracle=d_deq, max_dprob_vs_ggml_q8_0_activations=d_ggml,
                max_dprob_vs_bf16_oracle=d_bf, oracle_dequantised_vs_ggml=model_gap, ggml_self_noise=self_noise, top1=[float(x) for x in p_ggml.max(1)])
        assert (got.argmax(1) == p_ggml.argmax(1)).all()
        assert d_ggml < 4.0 * self_noise, (d_ggml, self_noise)      # bench.py's vs_reference_semantics.ratio (r05 verdict item 6), by the engine itself
        if dt == binding.F16:
            assert d_deq <= 1e-3                      # north_star's tolerance against the same dequantised weights
            assert d_ggml <= 2e-2                     # the stated bound for the un-modelled q8_0 activation rounding
        else:
            assert d_bf <= 6e-3 and d_ggml <= 3e-2


@pytest.mark.parametrize("head_scale", [8.0, 10.0])
def test_peaked_head_measured_deltas(pkg, binding, oracle, torch_gpu, head_scale):
    """head.weight x8 ... x10 so that the softmax is peaked.  On such a head the reference's own arithmetic is not
    reproducible to 1e-3 across f32 summation orders; the test prints the engine's deviation NEXT TO that noise floor
    (oracle with double-accumulated dot products vs the oracle itself) and bounds the first by a small multiple of the second."""
    name = "vit_base_patch16_224"
    path = pkg.synth.cached_synthetic(name, head_scale=head_scale)
    imgs = pkg.synth.normalize_u8(pkg.synth.synthetic_images_u8(8, 224, seed=77))
    om = oracle.OracleModel(path)
    _, rp = om.forward(imgs, oracle.REF)
    _, ap = om.forward(imgs, dataclasses.replace(oracle.REF, dot_exact=1))
    _, bp = om.forward(imgs, oracle.GPU_BF16)
    noise = float(np.abs(ap - rp).max())
    model = binding.Model(path)
    out = {}
    for dt, dname in ((binding.F16, "f16"), (binding.BF16, "bf16")):
        ctx = binding.Context(model, device=0, max_batch=8, dtype=dt)
        out[dname] = ctx.forward(imgs); ctx.close()
    model.close()
    d16, dbf, dbf_own = float(np.abs(out["f16"] - rp).max()), float(np.abs(out["bf16"] - rp).max()), float(np.abs(out["bf16"] - bp).max())
    _record(test="peaked_head", head_scale=head_scale, top1_prob=[round(float(x), 3) for x in rp.max(1)], oracle_summation_order_noise=noise,
            f16_max_dprob_vs_ref=d16, bf16_max_dprob_vs_ref=dbf, bf16_max_dprob_vs_bf16_oracle=dbf_own,
            bf16_top1_agree=int((out["bf16"].argmax(1) == rp.argmax(1)).sum()), images=int(rp.shape[0]))
    assert (out["f16"].argmax(1) == rp.argmax(1)).all()
    assert d16 <= max(3 * noise, 1e-3), (d16, noise)
    assert dbf <= 5e-2 and dbf_own <= 2.5e-2
    # bf16 (8-bit significand) may swap two near-tied classes on a peaked head: where its top-1 differs, the reference's own margin between
    # those two classes must be inside bf16's measured deviation
    for b in np.nonzero(out["bf16"].argmax(1) != rp.argmax(1))[0]:
        assert rp[b].max() - rp[b, out["bf16"][b].argmax()] <= 2 * dbf, (b, rp[b].max(), rp[b, out["bf16"][b].argmax()], dbf)


def test_context_options_do_not_change_results(pkg, binding, torch_gpu):
    """vitx_ctx_options only re-schedules: one stream / two / three, a forced split point and per-kernel LayerNorm all give the bits of
    the default context (images are independent in every kernel; the fused and the stand-alone LayerNorm compute the same statistics
    in the same order).  Unknown options and out-of-range values are refused."""
    name = "vit_base_patch16_224"
    path = pkg.synth.cached_synthetic(name, head_scale=8.0)
    imgs = pkg.synth.normalize_u8(pkg.synth.synthetic_images_u8(40, 224, seed=5))
    base_p, base_l = _bench_like(binding, torch_gpu, path, imgs, binding.BF16)
    for opts in ({"streams": 1}, {"streams": 3}, {"split_first": 13}, {"no_ln_fusion": 1}, {"streams": 1, "no_ln_fusion": 1}):
        p, l = _bench_like(binding, torch_gpu, path, imgs, binding.BF16, **opts)
        assert np.array_equal(l, base_l), opts
    model = binding.Model(path)
    with pytest.raises(TypeError):
        binding.Context(model, max_batch=4, bogus=1)
    with pytest.raises(binding.VitxError):
        binding.Context(model, max_batch=4, streams=9)
    model.close()


# ------------------------------------------------------------------------------------------------------------------
# LayerNorm fused into the residual GEMMs (GemmLn: statistics exchanged between the column tiles of a row block inside the launch)
# ------------------------------------------------------------------------------------------------------------------
def _gemm_ln_case(binding, torch, dt, tdt, M, N, K, test, timeout_us=200, seed=0):
    g = torch.Generator(device="cuda").manual_seed(1000 + seed + N + K)
    A = (torch.randn((M, K), device="cuda", generator=g) * 0.7).to(tdt)
    W = (torch.randn((N, K), device="cuda", generator=g) * 0.05).to(tdt)
    bias = torch.randn(N, device="cuda", generator=g) * 0.1
    # residual rows with a per-row offset and scale, so that mean and variance differ from row to row and from tile to tile
    x0 = torch.randn((M, N), device="cuda", generator=g) * (0.5 + torch.rand((M, 1), device="cuda", generator=g)) + torch.randn((M, 1), device="cuda", generator=g)
    x0[:, 300:320] += 6.0                                   # a block of outlier channels inside ONE column tile (what real ViT residual streams have)
    lw = 1 + 0.1 * torch.randn(N, device="cuda", generator=g); lb = 0.1 * torch.randn(N, device="cuda", generator=g)
    x = x0.clone(); y = torch.full((M, N), 9.0, dtype=tdt, device="cuda")
    fb = __import__("ctypes").c_int(-1)
    binding.check(binding.lib().vitx_op_gemm_ln(dt, A.data_ptr(), W.data_ptr(), bias.data_ptr(), x.data_ptr(), lw.data_ptr(), lb.data_ptr(), y.data_ptr(),
                                                M, N, K, 1e-6, test, timeout_us, fb, None), "vitx_op_gemm_ln")
    torch.cuda.synchronize()
    return A, W, bias, x0, lw, lb, x, y, fb.value


@pytest.mark.parametrize("dtype_name", ["f16", "bf16"])
@pytest.mark.parametrize("M,N,K", [(256 * 48, 768, 768), (256 * 44, 768, 3072), (256 * 33, 1024, 1024), (256 * 130, 256, 256), (256 * 67, 512, 256)])
def test_gemm_ln_fused_vs_reference_and_standalone(binding, torch_gpu, M, N, K, dtype_name):
    """The fused residual GEMM + LayerNorm: X against float64 products (as every GEMM test), Y against a float64 LayerNorm of the X the
    kernel itself produced (within one output ulp), and BIT-IDENTICAL to the stand-alone kernel applied to that X (the same tiled
    statistics: device_common.h).  1, 2, 3 and 4 column tiles; row-block counts that are not multiples of 8 (uneven XCD shares) and
    leave partial rounds; no tile may fall back on an otherwise idle GPU."""
    torch = torch_gpu
    dt = binding.F16 if dtype_name == "f16" else binding.BF16
    tdt = torch.float16 if dtype_name == "f16" else torch.bfloat16
    ulp = 2.0 ** -10 if dtype_name == "f16" else 2.0 ** -7
    A, W, bias, x0, lw, lb, x, y, fb = _gemm_ln_case(binding, torch, dt, tdt, M, N, K, 0)
    assert fb == 0
    rows = torch.from_numpy(np.unique(np.concatenate([np.arange(0, 8), np.arange(250, 262), np.arange(M - 260, M), np.random.default_rng(1).integers(0, M, 300)]))).cuda()
    a64, w64 = A[rows].double(), W.double()
    want_x = a64 @ w64.T + bias.double() + x0[rows].double()
    tol = (a64.abs() @ w64.abs().T) * 2e-6 + want_x.abs() * 2e-7 + 1e-6
    assert bool(((x[rows].double() - want_x).abs() <= tol).all())
    xs = x[rows].double()
    mean = xs.mean(1, keepdim=True); var = ((xs - mean) ** 2).mean(1, keepdim=True)
    want_y = (xs - mean) / torch.sqrt(var + 1e-6) * lw.double() + lb.double()
    err = (y[rows].double() - want_y).abs()
    assert bool((err <= want_y.abs() * ulp + 1e-3 * ulp + 2e-6).all()), float((err / (want_y.abs() * ulp + 1e-6)).max())
    y2 = torch.empty_like(y)
    binding.check(binding.lib().vitx_op_layernorm(dt, x.data_ptr(), lw.data_ptr(), lb.data_ptr(), y2.data_ptr(), M, N, 1e-6, None))
    torch.cuda.synchronize()
    assert torch.equal(y, y2)


@pytest.mark.parametrize("test_mode", [1, 3])
def test_gemm_ln_fallback_path_gives_the_same_bits(binding, torch_gpu, test_mode):
    """Every fifth tile pretends a peer timed out (1) or really withholds its statistics so that its peers time out (3; 50 us): those row
    blocks go through launch_layernorm_fixup and the result must equal the all-fused run bit for bit -- X and Y."""
    torch = torch_gpu
    M, N, K = 256 * 50, 768, 768
    ref = _gemm_ln_case(binding, torch, binding.BF16, torch.bfloat16, M, N, K, 0)
    got = _gemm_ln_case(binding, torch, binding.BF16, torch.bfloat16, M, N, K, test_mode, timeout_us=50)
    assert ref[8] == 0 and got[8] >= 50 * 3 // 5
    assert torch.equal(ref[6], got[6]) and torch.equal(ref[7], got[7])


@pytest.mark.parametrize("dtype_name", ["f16", "bf16"])
@pytest.mark.parametrize("name,n", [("vit_base_patch16_224", 256), ("vit_base_patch16_224", 97), ("vit_large_patch16_384", 70)])
def test_forward_ln_fusion_on_off_identical(pkg, binding, torch_gpu, name, n, dtype_name):
    """Whole forwards with the LayerNorms fused into proj / fc2 (default) and as their own launches (no_ln_fusion): identical logits, with
    ragged last row blocks (97 x 197 and 35 x 577 rows are not multiples of 256: the fused GEMMs store the padded rows as well), two
    sub-batch streams with two fused GEMMs in flight at once (256 images), both operand types; repeated forwards on one context stay
    identical, and the number of tiles that took the fix-up path is reported (it is 0 unless the two streams' GEMMs blocked each other)."""
    torch = torch_gpu
    dt = binding.F16 if dtype_name == "f16" else binding.BF16
    path = pkg.synth.cached_synthetic(name, head_scale=8.0)
    hp = pkg.synth.hparams_for(name)
    g = torch.Generator(device="cuda").manual_seed(n)
    imgs = torch.randn((n, hp.img_size, hp.img_size, 3), device="cuda", generator=g)
    outs = {}
    for off in (1, 0):
        model = binding.Model(path)
        ctx = binding.Context(model, device=0, max_batch=n, dtype=dt, no_ln_fusion=off)
        res = []
        for rep in range(3):
            probs = torch.empty((n, hp.num_classes), device="cuda"); logits = torch.empty_like(probs)
            ctx.forward_device(imgs.data_ptr(), n, probs.data_ptr(), logits.data_ptr(), 0)
            ctx.synchronize()
            res.append(logits.clone())
        assert torch.equal(res[0], res[1]) and torch.equal(res[0], res[2])
        outs[off] = res[0]
        fb = ctx.ln_fallbacks()
        if not off:
            _record(test="ln_fusion_forward", model=name, n=n, dtype=dtype_name, fixup_tiles_in_3_forwards=fb)
        else:
            assert fb == 0
        ctx.close(); model.close()
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1])


# ------------------------------------------------------------------------------------------------------------------
# One process, several GPUs (vitx_group_*): device-resident shards, persistent workers, top-k payload
# ------------------------------------------------------------------------------------------------------------------
def test_group_device_resident_shards_and_topk(pkg, binding, torch_gpu):
    """vitx_group_forward_device on a one-GPU box (the RCCL path with one rank): the shard is already in HBM, the gathered result is read
    from the device -- full probability rows, then the device-side top-5 pairs, which must equal vitx_topk of the same rows (descending,
    ties by the lower class).  Repeated calls run on the SAME persistent worker set (no thread is created per call)."""
    torch = torch_gpu
    import ctypes
    path = pkg.synth.cached_synthetic("vit_tiny_patch16_224", head_scale=4.0)
    imgs = pkg.synth.normalize_u8(pkg.synth.synthetic_images_u8(7, 224, seed=77))
    model = binding.Model(path)
    ctx = binding.Context(model, device=0, max_batch=8, dtype=binding.F16)
    want = ctx.forward(imgs); ctx.close()
    grp = binding.Group(model, [0], 8, binding.F16)
    d_imgs = torch.from_numpy(imgs).cuda()
    for n in (7, 3, 7):
        ptrs, n_max = grp.forward_device([d_imgs.data_ptr()], [n], topk=0)
        assert n_max == n
        got = np.empty((n, 1000), np.float32)
        torch.cuda.synchronize()
        assert ctypes.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(got.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptrs[0]), got.nbytes, 2) == 0
        assert np.array_equal(got, want[:n])
    ptrs, n_max = grp.forward_device([d_imgs.data_ptr()], [7], topk=5)
    pairs = np.empty((7, 5, 2), np.float32)
    assert ctypes.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(pairs.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptrs[0]), pairs.nbytes, 2) == 0
    for b in range(7):
        idx, val = binding.topk(want[b], 5)
        assert list(pairs[b, :, 1].view(np.int32)) == idx
        assert np.array_equal(pairs[b, :, 0], np.array(val, np.float32))
    with pytest.raises(binding.VitxError):
        grp.forward_device([d_imgs.data_ptr()], [9])              # 9 > 8 per device
    with pytest.raises(binding.VitxError):
        grp.forward_device([0], [3])                              # NULL pointer for a non-empty shard
    with pytest.raises(ValueError):
        grp.forward(imgs[:, :, :, 0])                             # wrong image shape (r02 advisor: unchecked buffer size)
    grp.close(); model.close()


def test_group_vitstr_output_size(pkg, binding, torch_gpu):
    """r02 advisor (medium): for a ViTSTR file the group writes 25 x num_classes floats per image; the binding allocated num_classes."""
    path = pkg.synth.cached_synthetic("vitstr_tiny_patch16_224", head_scale=4.0)
    rng = np.random.default_rng(3)
    imgs = rng.uniform(-1, 1, (5, 224, 224)).astype(np.float32)
    model = binding.Model(path)
    assert model.seq_len == 25
    ctx = binding.Context(model, device=0, max_batch=8, dtype=binding.F16)
    want = ctx.forward(imgs); ctx.close()
    grp = binding.Group(model, [0], 8, binding.F16)
    got = grp.forward(imgs)
    assert got.shape == (5, 25, model.num_classes) and np.array_equal(got, want)
    with pytest.raises(ValueError):
        grp.forward(np.zeros((2, 224, 224, 3), np.float32))       # a 3-channel batch for a one-channel model
    grp.close(); model.close()


def test_ln_test_switch_needs_its_key_and_fusion_is_reported(pkg, binding, torch_gpu):
    """The fault-injection switch is refused without VITX_LN_TEST_KEY (r03 advisor: it ships in the public options struct); a default context on
    an MI355X reports fused LayerNorms, a no_ln_fusion context does not."""
    path = pkg.synth.cached_synthetic("vit_base_patch16_224", head_scale=4.0)
    model = binding.Model(path)
    with pytest.raises(binding.VitxError):
        binding.Context(model, max_batch=64, dtype=binding.BF16, ln_test=1)
    a = binding.Context(model, max_batch=64, dtype=binding.BF16)
    b = binding.Context(model, max_batch=64, dtype=binding.BF16, no_ln_fusion=1)
    assert a.ln_fusion_active() == 1 and b.ln_fusion_active() == 0
    a.close(); b.close(); model.close()


def test_ln_fallback_budget_switches_the_fusion_off(pkg, binding, torch_gpu):
    """A context whose column-tile peers keep missing each other (a partitioned or shared device: here every fifth tile forced, ln_test bit 4 =
    the forced fall-backs count) must not spin for the time-out in every launch forever: after a window of 16 forwards with more than 8 fix-up
    tiles each the context runs stand-alone LayerNorms for good (r03 advisor).  Same bits before and after, and as a context that never fused."""
    torch = torch_gpu
    name, n = "vit_base_patch16_224", 256            # sub-batches wide enough for the fusing kernel
    path = pkg.synth.cached_synthetic(name, head_scale=8.0)
    hp = pkg.synth.hparams_for(name)
    imgs = torch.randn((n, hp.img_size, hp.img_size, 3), device="cuda", generator=torch.Generator(device="cuda").manual_seed(11))
    model = binding.Model(path)
    plain = binding.Context(model, max_batch=n, dtype=binding.BF16, no_ln_fusion=1)
    ctx = binding.Context(model, max_batch=n, dtype=binding.BF16, ln_test=binding.LN_TEST_KEY | 5)
    ref = torch.empty((n, hp.num_classes), device="cuda"); probs = torch.empty_like(ref)
    plain.forward_device(imgs.data_ptr(), n, ref.data_ptr(), 0, 0); plain.synchronize()
    assert ctx.ln_fusion_active() == 1
    states = []
    for rep in range(40):
        ctx.forward_device(imgs.data_ptr(), n, probs.data_ptr(), 0, 0); ctx.synchronize()
        assert torch.equal(probs, ref), rep
        states.append(ctx.ln_fusion_active())
    assert states[0] == 1 and states[-1] == -1 and states.index(-1) <= 34, states          # off at the first or second window boundary
    before = ctx.ln_fallbacks()
    assert before > 16 * 8
    ctx.forward_device(imgs.data_ptr(), n, probs.data_ptr(), 0, 0); ctx.synchronize()
    assert ctx.ln_fallbacks() == before and torch.equal(probs, ref)                          # no fusing launch any more
    ctx.close(); plain.close(); model.close()


@pytest.mark.parametrize("ln_test", [1, 3])
@pytest.mark.parametrize("name,n", [("vit_base_patch16_224", 256), ("vit_large_patch16_384", 70)])
def test_forward_ln_fallback_fixed_by_the_consumer_gemm(pkg, binding, torch_gpu, name, n, ln_test):
    """Whole forwards in which every fifth tile of every LayerNorm-fusing GEMM falls back (vitx_ctx_options::ln_test; 3 = with real 50 us
    time-outs of its peers): the row blocks left behind are recomputed from X in the PROLOGUE of the GEMM that consumes them (qkv after
    fc2, fc1 after proj: GemmArgs::fix) -- no launch in between -- and the logits equal the un-fused forward bit for bit."""
    torch = torch_gpu
    path = pkg.synth.cached_synthetic(name, head_scale=8.0)
    hp = pkg.synth.hparams_for(name)
    g = torch.Generator(device="cuda").manual_seed(7 + n)
    imgs = torch.randn((n, hp.img_size, hp.img_size, 3), device="cuda", generator=g)
    outs = {}
    for key, opts in (("plain", {"no_ln_fusion": 1}), ("forced", {"ln_test": binding.LN_TEST_KEY | ln_test})):
        model = binding.Model(path)
        ctx = binding.Context(model, device=0, max_batch=n, dtype=binding.BF16, **opts)
        probs = torch.empty((n, hp.num_classes), device="cuda"); logits = torch.empty_like(probs)
        for rep in range(2):
            ctx.forward_device(imgs.data_ptr(), n, probs.data_ptr(), logits.data_ptr(), 0); ctx.synchronize()
        outs[key] = logits.clone()
        fb = ctx.ln_fallbacks()
        assert (fb > 0) == (key == "forced"), (key, fb)
        ctx.close(); model.close()
    assert torch.isfinite(outs["forced"]).all() and torch.equal(outs["plain"], outs["forced"])


def test_two_forwards_in_flight_equal_the_split_schedule(pkg, binding, torch_gpu):
    """two contexts without the sub-batch split, fed alternately from two caller streams, keep two whole forwards in
    flight.  Each must return exactly what one context with the default two-sub-batch schedule returns for the same images, however the two
    interleave on the GPU (the LayerNorm-fusing GEMMs of both contexts poll their own statistics buffers)."""
    torch = torch_gpu
    name, n = "vit_base_patch16_224", 256
    path = pkg.synth.cached_synthetic(name, head_scale=8.0)
    m = binding.Model(path)
    g = torch.Generator(device="cuda").manual_seed(9)
    imgs = [torch.randn((n, 224, 224, 3), device="cuda", generator=g) for _ in range(2)]
    ref_ctx = binding.Context(m, 0, n, binding.BF16)
    refs = []
    for im in imgs:
        p = torch.empty((n, 1000), device="cuda")
        ref_ctx.forward_device(im.data_ptr(), n, p.data_ptr(), 0, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
        refs.append(p)
    ref_ctx.close()
    pair = [binding.Context(m, 0, n, binding.BF16, streams=1) for _ in range(2)]
    sts = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [torch.empty((n, 1000), device="cuda") for _ in range(2)]
    for it in range(6):
        for k in range(2):
            pair[k].forward_device(imgs[k].data_ptr(), n, outs[k].data_ptr(), 0, sts[k].cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(outs[0], refs[0]) and torch.equal(outs[1], refs[1]), f"iteration {it}"
        outs[0].zero_(); outs[1].zero_()
    assert pair[0].ln_fallbacks() >= 0
    for c in pair: c.close()
    m.close()


@pytest.mark.parametrize("n_img,N,H", [(1, 4097, 2), (2, 2305, 1)])
def test_attention_thousands_of_tokens(binding, torch_gpu, n_img, N, H):
    """Any token count means any: 4097 tokens = a 1024 x 1024 input at patch 16 (the pipelined two-pass kernel streams the keys; nothing is sized
    by N).  Against a float32 torch reference on the same fp16 / bf16 inputs."""
    torch = torch_gpu
    D = H * 64
    g = torch.Generator(device="cuda").manual_seed(N)
    for dt, tdt, tol in ((binding.F16, torch.float16, 5e-4), (binding.BF16, torch.bfloat16, 4e-3)):
        qkv = (torch.randn((n_img * N, 3 * D), device="cuda", generator=g) * 0.8).to(tdt)
        out = torch.full((n_img * N, D), float("nan"), device="cuda", dtype=tdt)
        binding.check(binding.lib().vitx_op_attention(dt, qkv.data_ptr(), out.data_ptr(), n_img, N, D, H, None))
        torch.cuda.synchronize()
        q, k, v = qkv.float().view(n_img, N, 3, H, 64).permute(2, 0, 3, 1, 4)
        ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).permute(0, 2, 1, 3).reshape(n_img * N, D)
        assert torch.isfinite(out).all() and float((out.float() - ref).abs().max()) <= tol


def test_contexts_of_one_model_share_the_device_weights(pkg, binding, torch_gpu):
    """vitx_ctx_shares_weights: the second context of a loaded model (same device, operand type, block mode) attaches to the first one's
    device copy; the copy lives as long as any of them; other operand types / another load of the file get their own."""
    torch = torch_gpu
    name, n = "vit_small_patch16_224", 8
    path = pkg.synth.cached_synthetic(name, head_scale=4.0)
    imgs = pkg.synth.normalize_u8(pkg.synth.synthetic_images_u8(n, 224, seed=3))
    m = binding.Model(path)
    torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
    a = binding.Context(m, 0, n, binding.BF16)
    free1 = torch.cuda.mem_get_info()[0]
    b = binding.Context(m, 0, n, binding.BF16, streams=1)
    free2 = torch.cuda.mem_get_info()[0]
    assert not a.shares_weights() and b.shares_weights() and a.weight_bytes() == b.weight_bytes() > 0
    assert (free1 - free2) < (free0 - free1) - a.weight_bytes() // 2          # the second context paid for scratch only
    c16 = binding.Context(m, 0, n, binding.F16)
    assert not c16.shares_weights()                                            # another operand type: its own copy
    pa = a.forward(imgs)
    a.close()                                                                  # the copy stays: b still holds it
    pb = b.forward(imgs)
    assert np.array_equal(pa, pb)
    d = binding.Context(m, 0, n, binding.BF16)
    assert d.shares_weights()
    assert np.array_equal(d.forward(imgs), pa)
    b.close(); d.close(); c16.close()
    e = binding.Context(m, 0, n, binding.BF16)                                 # every holder is gone: uploaded again
    assert not e.shares_weights() and np.array_equal(e.forward(imgs), pa)
    m2 = binding.Model(path)                                                   # another load of the same file is another model
    f = binding.Context(m2, 0, n, binding.BF16)
    assert not f.shares_weights()
    e.close(); f.close(); m.close(); m2.close()
