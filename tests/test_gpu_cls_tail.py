"""The last encoder layer of a classifier carries only the class-token row past its qkv projection (vitx_ctx::cls_tail, engine.cpp).

/root/reference/vit.cpp:910-911 reads row 0 of the last layer's output and nothing else; inside a layer, rows meet only through k and v
(vit.cpp:848-858).  So the engine computes, for the last layer, qkv of every token, then the class token's attention (attention_cls_kernel),
output projection, norm2 and MLP on one row per image.  `last_layer_all_rows=1` computes every row as the reference graph does.

  * the class-token attention kernel against row 0 of the full attention op, against the oracle, both operand types, the parity mode's planes,
    every supported head dim, ragged token counts;
  * whole forwards: default (class rows only) vs `last_layer_all_rows` vs the oracle, per model / operand type / quantised file / batch split;
  * an image's probabilities do not depend on the batch it arrives in (bits), with the tail on;
  * ViTSTR files and traced contexts compute every row.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RECORD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "cls_tail_record.jsonl")


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


@pytest.mark.parametrize("dtype_name", ["f16", "bf16"])
@pytest.mark.parametrize("n_img,N,D,H", [(7, 197, 768, 12), (3, 577, 1024, 16), (5, 50, 192, 3), (4, 65, 256, 8), (2, 197, 256, 2), (3, 17, 64, 4), (9, 1, 128, 2), (2, 785, 64, 8)])
def test_attention_cls_is_row_0_of_the_attention(binding, oracle, torch_gpu, dtype_name, n_img, N, D, H):
    """Head dims 64, 32, 128, 16, 8; 1 .. 785 tokens.  Against row 0 of the full attention op on the same operands (the two differ by the f32
    summation order and by the full kernels' rounding of P to the operand type before P.v) and against the oracle's rule for the operand type."""
    torch = torch_gpu
    dt, tdt = (binding.F16, torch.float16) if dtype_name == "f16" else (binding.BF16, torch.bfloat16)
    rng = np.random.default_rng(n_img * 131 + N * 7 + D + H)
    qkv32 = (rng.standard_normal((n_img * N, 3 * D)) * 0.8).astype(np.float32)
    dq = _dev(torch, qkv32, tdt)
    full = torch.empty((n_img * N, D), dtype=tdt, device="cuda")
    binding.check(binding.lib().vitx_op_attention(dt, dq.data_ptr(), full.data_ptr(), n_img, N, D, H, None), "vitx_op_attention")
    got = torch.full((n_img, D), float("nan"), dtype=tdt, device="cuda")
    binding.check(binding.lib().vitx_op_attention_cls(dt, dq.data_ptr(), 0, got.data_ptr(), n_img, N, D, H, None), "vitx_op_attention_cls")
    torch.cuda.synchronize()
    g = got.float().cpu().numpy()
    assert np.isfinite(g).all()
    f0 = full.float().cpu().numpy()[::N]
    ulp = 2.0 ** -10 if dtype_name == "f16" else 2.0 ** -7          # one operand ulp at magnitude ~1 .. 2
    assert np.abs(g - f0).max() <= 2 * ulp, float(np.abs(g - f0).max())
    ref = oracle.attention(dq.float().cpu().numpy(), n_img, N, D, H, oracle.GPU_F16 if dtype_name == "f16" else oracle.GPU_BF16)[::N]
    d = np.abs(g - ref)
    assert d.max() <= 2 * ulp and d.mean() <= ulp / 4, (float(d.max()), float(d.mean()))


@pytest.mark.parametrize("n_img,N,H", [(6, 197, 12), (3, 577, 16), (4, 33, 3)])
def test_attention_cls_on_the_parity_modes_planes(binding, oracle, torch_gpu, n_img, N, H):
    """hi + lo / 2048 operands (what EPI_BIAS_HILO emits), f32 products: against the reference semantics on the f32 values, and against row 0 of
    the precise streaming kernel on the same planes."""
    torch = torch_gpu
    D = H * 64
    rows = n_img * N
    rng = np.random.default_rng(N * 3 + H)
    x = _dev(torch, (rng.standard_normal((rows, 3 * D)) * 0.8).astype(np.float32))
    hi = x.to(torch.float16)
    lo = ((x - hi.float()) * 2048.0).to(torch.float16)
    pad = 8
    buf = torch.full((2 * (rows + pad), 3 * D), float("nan"), dtype=torch.float16, device="cuda")      # NaN rows behind each plane: never read
    buf[:rows] = hi; buf[rows + pad:2 * rows + pad] = lo
    lo_off = (rows + pad) * 3 * D
    got = torch.full((n_img, D), float("nan"), dtype=torch.float16, device="cuda")
    binding.check(binding.lib().vitx_op_attention_cls(binding.F16, buf.data_ptr(), lo_off, got.data_ptr(), n_img, N, D, H, None), "vitx_op_attention_cls")
    full = torch.empty((rows, D), dtype=torch.float16, device="cuda")
    binding.check(binding.lib().vitx_op_attention_planes(buf.data_ptr(), lo_off, full.data_ptr(), n_img, N, D, H, None), "vitx_op_attention_planes")
    torch.cuda.synchronize()
    g = got.float().cpu().numpy()
    assert np.isfinite(g).all()
    ref = oracle.attention(x.cpu().numpy(), n_img, N, D, H, oracle.REF)[::N]
    d = np.abs(g - ref)
    assert d.max() <= 3e-3 and d.mean() <= 3e-4, (float(d.max()), float(d.mean()))
    df = np.abs(g - full.float().cpu().numpy()[::N])
    assert df.max() <= 2.0 ** -9, float(df.max())


def test_attention_cls_argument_checks(binding, torch_gpu):
    torch = torch_gpu
    L = binding.lib()
    q = torch.zeros((4 * 10, 3 * 96), dtype=torch.float16, device="cuda"); o = torch.zeros((4, 96), dtype=torch.float16, device="cuda")
    assert L.vitx_op_attention_cls(binding.F16, q.data_ptr(), 0, o.data_ptr(), 4, 10, 96, 4, None) != 0          # head dim 24
    assert b"head_dim" in L.vitx_last_error()
    assert L.vitx_op_attention_cls(binding.BF16, q.data_ptr(), 4 * 10 * 3 * 96, o.data_ptr(), 4, 10, 96, 12, None) != 0      # planes are F16 only
    assert L.vitx_op_attention_cls(binding.F16, q.data_ptr(), 100, o.data_ptr(), 4, 10, 96, 12, None) != 0       # lo plane inside the hi plane
    assert L.vitx_op_attention_cls(binding.F16, None, 0, o.data_ptr(), 4, 10, 96, 12, None) != 0
    assert L.vitx_op_attention_cls(7, q.data_ptr(), 0, o.data_ptr(), 4, 10, 96, 12, None) != 0


CASES = [("vit_tiny_patch16_224", 5, "f16", {}), ("vit_tiny_patch16_224", 37, "bf16", {}), ("vit_base_patch16_224", 3, "f16", {}), ("vit_base_patch16_224", 24, "bf16", {}),
         ("vit_base_patch16_224", 19, "f16", {"f16_fast_attention": 1}), ("vit_micro_patch8_224", 2, "bf16", {}), ("vit_base_patch16_224", 40, "bf16", {"no_ln_fusion": 1}),
         ("vit_tiny_patch16_224", 1, "bf16", {"graph": 1}), ("vit_micro_hd32_patch16_64", 5, "f16", {}), ("vit_micro_hd32_patch16_64", 5, "bf16", {}),
         ("vit_micro_hd96_patch16_96", 3, "bf16", {})]           # head dim 96: no class-row kernel, the context computes every row either way


@pytest.mark.parametrize("name,n,dtype_name,opts", CASES)
def test_forward_class_rows_only_vs_every_row_vs_oracle(pkg, binding, oracle, torch_gpu, name, n, dtype_name, opts):
    """Both forms against the oracle with the test suite's usual bounds (F16 1e-3 vs the reference, bf16 6e-3 vs the bf16 oracle, x4 head), and
    against each other: they differ by the class row's attention arithmetic in ONE layer only."""
    path = pkg.synth.cached_synthetic(name, head_scale=4.0)
    hp = pkg.synth.hparams_for(name)
    imgs = pkg.synth.normalize_u8(pkg.synth.synthetic_images_u8(n, hp.img_size, seed=77))
    dt = binding.F16 if dtype_name == "f16" else binding.BF16
    _, ref = oracle.OracleModel(path).forward(imgs, oracle.REF if dtype_name == "f16" else oracle.GPU_BF16)
    model = binding.Model(path)
    res = {}
    for label, extra in (("cls", {}), ("all", {"last_layer_all_rows": 1})):
        ctx = binding.Context(model, max_batch=n, dtype=dt, **opts, **extra)
        res[label] = ctx.forward(imgs)
        if opts.get("graph"):
            for _ in range(3):
                again = ctx.forward(imgs)              # captured the second time, replayed after
                assert np.array_equal(again, res[label])
        ctx.close()
    model.close()
    tol = 1e-3 if dtype_name == "f16" else 6e-3
    dc, da, dd = (float(np.abs(res["cls"] - ref).max()), float(np.abs(res["all"] - ref).max()), float(np.abs(res["cls"] - res["all"]).max()))
    _record(test="forward_cls_tail", model=name, images=n, dtype=dtype_name, opts=opts, cls_rows_vs_oracle=dc, all_rows_vs_oracle=da, cls_vs_all=dd)
    assert np.isfinite(res["cls"]).all() and np.abs(res["cls"].sum(1) - 1).max() < 1e-4
    assert dc <= tol and da <= tol and dd <= tol
    assert (res["cls"].argmax(1) == res["all"].argmax(1)).all()


@pytest.mark.parametrize("ftype,ftype_id", [("q4_0", 2), ("q8_0", 8), ("q5_1", 7)])
@pytest.mark.parametrize("dtype_name", ["f16", "bf16"])
def test_forward_class_rows_only_on_quantised_files(pkg, binding, oracle, torch_gpu, tmp_path, ftype, ftype_id, dtype_name):
    """The tail GEMMs take the just-in-time expansion of the last layer's blocks like every other GEMM of the layer."""
    name, n = "vit_tiny_patch16_224", 21
    src = pkg.synth.cached_synthetic(name, head_scale=4.0)
    path = str(tmp_path / f"t_{ftype}.gguf")
    binding.quantize_file(src, path, ftype_id)
    hp = pkg.synth.hparams_for(name)
    imgs = pkg.synth.normalize_u8(pkg.synth.synthetic_images_u8(n, hp.img_size, seed=5))
    dt = binding.F16 if dtype_name == "f16" else binding.BF16
    model = binding.Model(path)
    res = {}
    for label, extra in (("cls", {}), ("all", {"last_layer_all_rows": 1}), ("cls_host", {"quant_on_host": 1})):
        ctx = binding.Context(model, max_batch=n, dtype=dt, **extra)
        res[label] = ctx.forward(imgs); ctx.close()
    model.close()
    tol = 1e-3 if dtype_name == "f16" else 6e-3
    assert np.abs(res["cls"] - res["all"]).max() <= tol
    assert np.array_equal(res["cls"], res["cls_host"])             # same expanded bits wherever the expansion happens
    assert (res["cls"].argmax(1) == res["all"].argmax(1)).all()


@pytest.mark.parametrize("dtype_name", ["f16", "bf16"])
def test_an_images_result_does_not_depend_on_its_batch_with_the_tail_on(pkg, binding, torch_gpu, dtype_name):
    """The tail GEMMs run on round_up(images, 128) rows: 1, 7, 130 and 300 images take different kernel families (skinny ring / wide tile) and
    different sub-batch cuts.  Same bits for the same image everywhere."""
    torch = torch_gpu
    name = "vit_base_patch16_224"
    dt = binding.BF16 if dtype_name == "bf16" else binding.F16
    path = pkg.synth.cached_synthetic(name, head_scale=8.0); hp = pkg.synth.hparams_for(name)
    model = binding.Model(path)
    imgs = torch.randn((300, hp.img_size, hp.img_size, 3), device="cuda", generator=torch.Generator(device="cuda").manual_seed(11))
    c = binding.Context(model, max_batch=300, dtype=dt)
    p300 = torch.empty((300, hp.num_classes), device="cuda")
    c.forward_device(imgs.data_ptr(), 300, p300.data_ptr(), 0, 0); c.synchronize()
    for first, cnt in ((0, 1), (5, 7), (100, 130), (299, 1), (40, 256)):
        p = torch.empty((cnt, hp.num_classes), device="cuda")
        c.forward_device(imgs[first:first + cnt].contiguous().data_ptr(), cnt, p.data_ptr(), 0, 0); c.synchronize()
        assert torch.equal(p, p300[first:first + cnt]), (first, cnt)
    c.close(); model.close()


def test_traced_contexts_compute_every_row_of_the_last_layer(pkg, binding, oracle, torch_gpu):
    """vitx_trace_enable shows the residual stream of EVERY token after every layer: a traced forward runs the full last layer (its last stage
    equals the all-rows context's), and switching the trace off returns to the tail (bits of an untraced context)."""
    name, n = "vit_tiny_patch16_224", 6
    path = pkg.synth.cached_synthetic(name, head_scale=4.0)
    hp = pkg.synth.hparams_for(name)
    imgs = pkg.synth.normalize_u8(pkg.synth.synthetic_images_u8(n, hp.img_size, seed=9))
    model = binding.Model(path)
    c = binding.Context(model, max_batch=n, dtype=binding.F16)
    p_tail = c.forward(imgs)
    c.trace_enable([0, n - 1])
    p_traced = c.forward(imgs)
    tr = c.trace_read()
    c.trace_enable([])
    p_again = c.forward(imgs)
    c.close()
    ca = binding.Context(model, max_batch=n, dtype=binding.F16, last_layer_all_rows=1)
    p_all = ca.forward(imgs)
    ca.trace_enable([0, n - 1]); ca.forward(imgs); tr_all = ca.trace_read(); ca.close()
    model.close()
    assert np.array_equal(p_traced, p_all) and np.array_equal(p_again, p_tail)
    assert np.array_equal(np.asarray(tr), np.asarray(tr_all))
    last = np.asarray(tr)[-1]
    assert np.isfinite(last).all() and np.abs(last[:, 1:, :]).max() > 0           # token rows of the last stage are really computed
