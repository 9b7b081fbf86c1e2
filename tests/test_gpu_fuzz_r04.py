"""Randomly drawn architectures (seeded: the same 48 every run) through the whole forward, against the oracle.

The reference is generic in every hparam (vit.h:20-37: hidden size, depth, heads, classes, patch and image size) and its converter writes whatever a timm
checkpoint has; the fixed fixtures of the other modules sit on the headline shapes.  This module walks the space between them: widths of every
LayerNorm instantiation, head dims 64 (tuned kernels, f32-grade attention in F16 mode) and others (generic kernel), token counts on both sides
of every kernel boundary (<= 192, 193..224, 225..288, > 288), ragged class counts, batches that do and do not split into sub-batches."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WIDTHS = [64, 128, 192, 256, 320, 384, 448, 512, 576, 640, 768, 896, 1024, 1152, 1280]


def draw(seed):
    r = np.random.RandomState(1000 + seed)
    D = int(r.choice(WIDTHS))
    heads = [h for h in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 18, 20) if D % h == 0 and (D // h) % 8 == 0 and 8 <= D // h <= 128]
    if r.rand() < 0.6 and D % 64 == 0:
        H = D // 64
    else:
        H = int(r.choice(heads))
    P = int(r.choice([8, 14, 16, 32]))
    g = int(r.choice([1, 2, 3, 5, 7, 13, 14, 15, 16, 17])) if P >= 14 else int(r.choice([2, 4, 9, 14, 15]))
    L = int(r.choice([1, 2, 3]))
    C = int(r.choice([2, 10, 37, 100, 1000, 1001]))
    n = int(r.choice([1, 2, 3, 5, 17, 33]))
    ftype = 1
    if seed >= 24:                                            # the second half: longer sequences (the pipelined kernels), other file types
        g = int(r.choice([3, 14, 17, 18, 19, 20, 24])) if P >= 14 else int(r.choice([14, 18, 24]))
        ftype = int(r.choice([1, 1, 0, 2, 3, 6, 7, 8]))      # f16, f32, q4_0, q4_1, q5_0, q5_1, q8_0
        if D >= 768: L = 1
    if g * g + 1 > 300 or D >= 1024: n = min(n, 3)          # keep the oracle in seconds
    return (D, L, H, C, P, g * P), n, ftype


@pytest.mark.parametrize("seed", range(48))
def test_random_architecture_vs_oracle(pkg, binding, oracle, torch_gpu, seed):
    import dataclasses
    cfg, n, ftype = draw(seed)
    name = "fuzz_%d" % seed
    pkg.synth.CONFIGS[name] = cfg
    D, L, H, C, P, S = cfg
    path = pkg.synth.cached_synthetic(name, ftype=ftype, head_scale=4.0)
    imgs = pkg.synth.normalize_u8(pkg.synth.synthetic_images_u8(n, S, seed=seed))
    om = oracle.OracleModel(path)
    REF = dataclasses.replace(oracle.REF, quant_act=0)       # quantised files: the dequantised-weight semantics the engine implements
    _, rp = om.forward(imgs, REF)
    _, bp = om.forward(imgs, oracle.GPU_BF16)
    _, xp = om.forward(imgs, dataclasses.replace(REF, dot_exact=1))
    noise = float(np.abs(xp - rp).max())
    model = binding.Model(path)
    for dt, ref, tol in ((binding.F16, rp, max(1e-3, 3 * noise)), (binding.BF16, bp, max(8e-3, 10 * noise))):
        ctx = binding.Context(model, max_batch=max(n, 16), dtype=dt)
        probs = ctx.forward(imgs)
        again = ctx.forward(imgs)
        ctx.close()
        assert np.isfinite(probs).all() and np.abs(probs.sum(1) - 1).max() < 1e-4, cfg
        assert np.array_equal(probs, again), cfg
        d = float(np.abs(probs - ref).max())
        assert d <= tol, (cfg, n, ftype, "f16" if dt == binding.F16 else "bf16", d, tol, noise)
    model.close()


@pytest.mark.parametrize("name,nmax", [("vit_base_patch16_224", 300), ("vit_large_patch16_384", 70)])
@pytest.mark.parametrize("dtype_name", ["bf16", "f16"])
def test_every_batch_size_gives_every_image_its_batch_1_bits(pkg, binding, torch_gpu, dtype_name, name, nmax):
    """An image's result must not depend on the batch it arrives in.  The engine changes kernel family with the row count (64 x 128 and 128 x 256 ring
    tiles, the 256 x 256 persistent kernel, the LayerNorm-fusing build from 128 tiles on, one or two sub-batches from 16 images on, the persistent
    attention at every size): every batch size 1..24 and forty more up to 300 must reproduce, bit for bit, what each image gets alone."""
    torch = torch_gpu
    dt = binding.BF16 if dtype_name == "bf16" else binding.F16
    path = pkg.synth.cached_synthetic(name, head_scale=8.0); hp = pkg.synth.hparams_for(name)
    model = binding.Model(path)
    imgs = torch.randn((nmax, hp.img_size, hp.img_size, 3), device="cuda", generator=torch.Generator(device="cuda").manual_seed(77))
    one = binding.Context(model, max_batch=1, dtype=dt)
    ref = torch.empty((nmax, hp.num_classes), device="cuda")
    per = hp.img_size * hp.img_size * 3 * 4
    for i in range(nmax):
        one.forward_device(imgs.data_ptr() + i * per, 1, ref.data_ptr() + i * hp.num_classes * 4, 0, 0)
    one.synchronize(); one.close()
    ctx = binding.Context(model, max_batch=nmax, dtype=dt)
    sizes = list(range(1, 25)) + sorted(set(int(x) for x in np.random.RandomState(5).randint(25, nmax + 1, size=40 if nmax > 100 else 12))) + [nmax]
    out = torch.empty((nmax, hp.num_classes), device="cuda")
    for n in sizes:
        out.fill_(-1.0)
        ctx.forward_device(imgs.data_ptr(), n, out.data_ptr(), 0, 0); ctx.synchronize()
        assert torch.equal(out[:n], ref[:n]), (n, ctx.split(n))
        assert float(out[n:].max()) == -1.0 if n < nmax else True          # nothing written past the batch
    ctx.close(); model.close()


def test_five_models_in_flight_stay_repeatable(pkg, binding, torch_gpu):
    """Five contexts of four models and both operand types, each on its own caller stream, all in flight at once, 80 rounds: every forward must
    return the bits the context produced alone.  (r04 found a co-residency hazard between an MFMA-streaming kernel with spare registers and another
    stream's DPP reductions -- profiles/r04/coresidency_layernorm.txt; with several models in flight far more kernel pairs meet on a SIMD than
    the two sub-batches of one forward produce.)"""
    torch = torch_gpu
    specs = [("vit_base_patch16_224", 64, binding.BF16), ("vit_large_patch16_384", 16, binding.BF16), ("vit_tiny_patch16_224", 64, binding.F16),
             ("vit_base_patch16_224", 48, binding.F16), ("vit_small_patch16_224", 200, binding.BF16)]
    live = []
    for name, n, dt in specs:
        path = pkg.synth.cached_synthetic(name, head_scale=8.0); hp = pkg.synth.hparams_for(name)
        m = binding.Model(path); c = binding.Context(m, max_batch=n, dtype=dt)
        imgs = torch.randn((n, hp.img_size, hp.img_size, 3), device="cuda", generator=torch.Generator(device="cuda").manual_seed(n))
        ref = torch.empty((n, hp.num_classes), device="cuda"); out = torch.empty_like(ref)
        c.forward_device(imgs.data_ptr(), n, ref.data_ptr(), 0, 0); c.synchronize()
        live.append(dict(m=m, c=c, imgs=imgs, ref=ref, out=out, n=n, st=torch.cuda.Stream(), name=name))
    torch.cuda.synchronize()
    for it in range(80):
        for x in live:
            with torch.cuda.stream(x["st"]):
                x["out"].zero_()
            x["c"].forward_device(x["imgs"].data_ptr(), x["n"], x["out"].data_ptr(), 0, x["st"].cuda_stream)
        torch.cuda.synchronize()
        for x in live:
            assert torch.equal(x["out"], x["ref"]), (x["name"], x["n"], it)
    for x in live:
        x["c"].close(); x["m"].close()
