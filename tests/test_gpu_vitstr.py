"""The ViTSTR scene-text extension (/root/reference/extensions/vitstr.cpp) on the GPU: the same encoder on one grey input plane
(vitstr.cpp:713-731), the head on the first 25 tokens of every image (:864-904), greedy decode (:1025-1051) -- against the oracle,
through the C ABI and through the C++ mirror (examples/vitstr_main.cpp)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSET_DIR = os.path.join(ROOT, "tests", "golden", "assets")


def _text(pkg, binding, probs):
    ids, score = binding.vitstr_decode(probs)
    return "".join(pkg.synth.VITSTR_LABELS[i] for i in ids), score


@pytest.mark.parametrize("n", [3, 20])          # 20 images: two sub-batch streams
def test_forward_matches_oracle_f16(pkg, binding, oracle, torch_gpu, n):
    path = pkg.synth.cached_synthetic("vitstr_tiny_patch16_224", head_scale=4.0)
    rng = np.random.default_rng(n)
    imgs = np.clip(rng.standard_normal((n, 224, 224)) * 0.5, -1, 1).astype(np.float32)
    model = binding.Model(path)
    ctx = binding.Context(model, device=0, max_batch=n, dtype=binding.F16)
    probs, logits = ctx.forward(imgs, want_logits=True)
    assert probs.shape == (n, 25, 96)
    rl, rp = oracle.OracleModel(path).forward(imgs, oracle.REF)
    assert np.abs(probs - rp).max() <= 1e-3                      # north_star tolerance, every one of the 25 positions
    assert np.abs(logits - rl).max() <= 2.5e-2
    # the decoded characters agree wherever the reference separates its two best classes by more than the tolerance (a random-init
    # model has near-ties; a different f32 summation order -- e.g. the patch kernel's k order since r03 -- may flip those)
    srt = np.sort(rp, -1)
    decided = (srt[..., -1] - srt[..., -2]) > 2e-3
    assert decided.mean() > 0.9 and ((probs.argmax(-1) == rp.argmax(-1)) | ~decided).all()
    for b in range(n):
        if decided[b].all():
            assert _text(pkg, binding, probs[b])[0] == _text(pkg, binding, rp[b])[0]
    # bf16 engine: same bound as the classifier's bf16 mode (test_gpu_e2e.test_forward_bf16_mode_tracks_its_own_oracle)
    ctxb = binding.Context(model, device=0, max_batch=n, dtype=binding.BF16)
    pb = ctxb.forward(imgs)
    _, rb = oracle.OracleModel(path).forward(imgs, oracle.GPU_BF16)
    assert np.abs(pb - rb).max() <= 5e-3
    ctx.close(); ctxb.close(); model.close()


def test_bundled_image_end_to_end_and_cpp_mirror(pkg, binding, oracle, torch_gpu, tmp_path):
    """JPEG -> load_image_from_file -> grey preprocess -> forward -> decode: the Python binding and examples/vitstr_main.cpp (the
    extension's main.cpp flow on the drop-in header) print the text the oracle decodes from the same bytes."""
    import subprocess
    path = pkg.synth.cached_synthetic("vitstr_tiny_patch16_224", head_scale=4.0)
    jpg = os.path.join(ASSET_DIR, "tench.jpg")
    u8 = binding.load_image(jpg)
    x = binding.preprocess_vitstr(u8, 224)
    assert np.array_equal(x, oracle.preprocess_vitstr(u8, 224))
    model = binding.Model(path)
    ctx = binding.Context(model, device=0, max_batch=1, dtype=binding.F16)
    probs = ctx.forward(x[None])
    _, rp = oracle.OracleModel(path).forward(x[None], oracle.REF)
    assert np.abs(probs - rp).max() <= 1e-3
    text, score = _text(pkg, binding, probs[0])
    want_text, want_score = _text(pkg, binding, rp[0])
    assert text == want_text and abs(score - want_score) <= 0.05 * max(want_score, 1e-30) + 1e-12
    ctx.close(); model.close()

    pkgdir = os.path.join(ROOT, "vit.cpp_amd")
    exe = str(tmp_path / "vitstr_main")
    r = subprocess.run(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "examples", "vitstr_main.cpp"), "-I" + pkgdir, "-L" + pkgdir, "-lvitx", "-L/opt/rocm/lib",
                        "-Wl,-rpath," + pkgdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe, "-m", path, "-i", jpg], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout.splitlines()
    i = out.index("------------------ ")
    assert out[i + 1] == want_text and out[i + 2].startswith("score : ") and out[i + 3] == "------------------ "
    assert "processed, out dims : (224 x 224)" in r.stderr


def test_classifier_entry_points_refuse_a_vitstr_model_and_vice_versa(pkg, binding, torch_gpu):
    """One library, two model kinds: the context reports how many rows it writes; a 17-token model cannot feed a 25-token head."""
    path = pkg.synth.cached_synthetic("vitstr_tiny_patch16_224", head_scale=4.0)
    m = binding.Model(path)
    ctx = binding.Context(m, device=0, max_batch=2, dtype=binding.F16)
    assert binding.lib().vitx_ctx_out_rows(ctx._h) == 25
    ctx.close(); m.close()
    small = pkg.synth.cached_synthetic("vitstr_micro_patch16_64", head_scale=4.0)       # 17 tokens < 25
    m = binding.Model(small)
    with pytest.raises(binding.VitxError, match="25 tokens"):
        binding.Context(m, device=0, max_batch=1, dtype=binding.F16)
    m.close()
    cls = pkg.synth.cached_synthetic("vit_tiny_patch16_224", head_scale=4.0)
    m = binding.Model(cls)
    ctx = binding.Context(m, device=0, max_batch=1, dtype=binding.F16)
    assert binding.lib().vitx_ctx_out_rows(ctx._h) == 1
    ctx.close(); m.close()
