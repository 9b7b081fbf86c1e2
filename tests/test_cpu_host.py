"""CPU tests of the host side of libvitx.so: C-ABI exports, the model-file loader
(acceptance rules of vit_model_load, vit.cpp:308-712), preprocess (vit.cpp:130-305),
top-k, and loud failure without a GPU.  No kernel is launched here."""
import ctypes as C
import os
import re
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_library_exports_every_declared_symbol(binding):
    """Every function include/vitx.h declares is exported by libvitx.so (and vice versa for the binding list)."""
    hdr = open(os.path.join(ROOT, "include", "vitx.h")).read()
    declared = set(re.findall(r"\b(vitx_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"vitx_status", "vitx_dtype", "vitx_interp"}
    L = binding.lib()
    for sym in sorted(declared):
        assert hasattr(L, sym), f"libvitx.so does not export {sym}"
    assert declared == set(binding.EXPORTS)


def test_cpp_api_symbols_present(binding):
    """The C++ drop-in entry points of vit.h (vit_model_load / vit_image_preprocess / vit_predict) are in the library."""
    import subprocess
    out = subprocess.check_output(["nm", "-DC", binding.LIB_PATH]).decode()
    for sig in ("vit_model_load(", "vit_image_preprocess(", "vit_predict(", "vit_predict_batch(", "vit_params_parse(", "print_usage("):
        assert sig in out, sig


def test_loader_round_trip(pkg, binding, tmp_path):
    name = "vit_micro_patch16_64"
    hp = pkg.synth.hparams_for(name)
    w = pkg.synth.make_weights(hp)
    p = str(tmp_path / "m.gguf")
    pkg.ggml_file.write_model(p, hp, w, id2label={i: f"class {i}" for i in range(hp.num_classes)})
    m = binding.Model(p)
    h = m.hparams
    assert (h.hidden_size, h.num_hidden_layers, h.num_attention_heads, h.num_classes, h.patch_size, h.img_size, h.ftype) == (128, 2, 2, 10, 16, 64, 1)
    assert abs(h.eps - 1e-6) < 1e-12
    assert m.label(3) == "class 3" and m.label(99) is None
    tens = m.tensors()
    assert len(tens) == 8 + 12 * hp.num_hidden_layers == 32
    byname = {t[0]: (i, t) for i, t in enumerate(tens)}
    # ggml dims are the reversed torch shape; 2-D weights f16, vectors / pos / cls f32, patch bias 4-D
    i, t = byname["blocks.1.attn.qkv.weight"]; assert t[1] == 1 and t[2][:2] == (128, 384)
    assert np.array_equal(m.tensor_f32(i).reshape(384, 128), w["blocks.1.attn.qkv.weight"].astype(np.float16).astype(np.float32))
    i, t = byname["pos_embed"]; assert t[1] == 0 and np.array_equal(m.tensor_f32(i).reshape(w["pos_embed"].shape), w["pos_embed"])
    i, t = byname["patch_embed.proj.bias"]; assert t[2] == (1, 1, 128, 1)
    i, t = byname["patch_embed.proj.weight"]; assert t[1] == 1 and t[2] == (16, 16, 3, 128)
    # the Python reader sees the same file
    mf = pkg.ggml_file.read_model(p)
    assert [r.name for r in mf.tensors] == [t[0] for t in tens]


@pytest.mark.parametrize("ftype", [2, 3, 6, 7, 8])
def test_loader_dequantises_like_the_python_reference(pkg, binding, tmp_path, ftype):
    name = "vit_micro_patch16_64"
    p = str(tmp_path / "q.gguf")
    pkg.synth.write_synthetic(p, name, ftype=ftype)
    m = binding.Model(p)
    assert m.hparams.ftype == ftype
    mf = pkg.ggml_file.read_model(p)
    for i, (nm, tt, ne, nb) in enumerate(m.tensors()):
        rec = mf.tensors[i]
        assert rec.ttype == tt and len(rec.raw) == nb
        if nm.endswith("fc1.weight") or nm == "head.weight":
            assert tt == ftype
            got = m.tensor_f32(i).reshape(-1)
            ref = pkg.ggml_file.dequantize(tt, rec.raw, got.size)
            assert np.array_equal(got, ref)
        if nm in ("patch_embed.proj.weight",):
            assert tt == 1          # 4-D tensors are never quantised (quantize.cpp:207-223)


def test_q4_0_block_encoding_known_answer(pkg):
    """quantize_row_q4_0_reference on a hand-computed block: max = -8 -> d = 1, q = x + 8 clipped to 15."""
    x = np.array([-8, -7, -6, -5, -4, -3, -2, -1, 0, 1, 2, 3, 4, 5, 6, 7] * 2, np.float32)
    raw = pkg.ggml_file.quantize_q4_0(x)
    assert len(raw) == 18
    assert np.frombuffer(raw[:2], np.float16)[0] == 1.0
    back = pkg.ggml_file.dequantize(2, raw, 32)
    assert np.array_equal(back, x)
    raw8 = pkg.ggml_file.quantize_q8_0(np.arange(-16, 16, dtype=np.float32))
    assert len(raw8) == 34 and np.abs(pkg.ggml_file.dequantize(8, raw8, 32) - np.arange(-16, 16)).max() <= 16 / 127 / 2 + 1e-3


@pytest.mark.parametrize("ftype", [2, 3, 6, 7, 8])
def test_native_quantize_tool_is_byte_identical_to_the_python_restatement(pkg, binding, tmp_path, ftype):
    """vitx_quantize_file (C++, quantize.cpp:34-353 rules) on an f16 file == the file the numpy block encoders write
    from the same weights: two independent restatements of ggml's quantize_row_*_reference agree byte for byte, for
    every block type, on real weight statistics plus adversarial rows (all-zero block, ties, +-max, denormal scale)."""
    name = "vit_micro_patch16_64"
    hp = pkg.synth.hparams_for(name)
    w = pkg.synth.make_weights(hp, seed=77, head_scale=8.0)
    fc1 = w["blocks.0.mlp.fc1.weight"]
    fc1[0, :32] = 0.0                                        # d == 0 -> id = 0
    fc1[1, :32] = np.tile(np.float32([0.5, -0.5]), 16)       # |x| ties: the FIRST largest wins
    fc1[2, :32] = np.linspace(-1, 1, 32, dtype=np.float32)   # max at the end, min at the start
    fc1[3, :32] = np.float32(1e-7)                           # scale underflows to an fp16 denormal / zero
    fc1[4, :32] = np.float32([65504.0] + [-65504.0] * 31)    # fp16 extremes
    src, ref, out = str(tmp_path / "f16.gguf"), str(tmp_path / "ref.gguf"), str(tmp_path / "out.gguf")
    pkg.ggml_file.write_model(src, hp, w, ftype=1)
    pkg.ggml_file.write_model(ref, hp, w, ftype=ftype)
    binding.quantize_file(src, out, ftype)
    a, b = open(ref, "rb").read(), open(out, "rb").read()
    assert len(a) == len(b)
    assert a == b
    # and the result loads and dequantises through the product loader
    m = binding.Model(out)
    assert m.hparams.ftype == ftype
    names = {t[0]: t[1] for t in m.tensors()}
    assert names["blocks.0.mlp.fc1.weight"] == ftype and names["head.weight"] == ftype
    assert names["patch_embed.proj.weight"] == 1 and names["pos_embed"] == 0 and names["norm.weight"] == 0


def test_native_quantize_tool_errors(pkg, binding, tmp_path):
    name = "vit_micro_patch16_64"
    src = str(tmp_path / "f16.gguf"); pkg.synth.write_synthetic(src, name, ftype=1)
    q = str(tmp_path / "q.gguf")
    with pytest.raises(binding.VitxError):
        binding.quantize_file(src, q, 1)                     # f16 is not a quantisation target (quantize.cpp:296-300)
    with pytest.raises(binding.VitxError):
        binding.quantize_file(str(tmp_path / "missing.gguf"), q, 2)
    binding.quantize_file(src, q, 2)
    with pytest.raises(binding.VitxError):
        binding.quantize_file(q, str(tmp_path / "qq.gguf"), 8)   # already quantised input
    # a failed run leaves nothing behind (validated before the first write, written to a temporary file, renamed on success) ...
    assert not os.path.exists(str(tmp_path / "qq.gguf")) and not [f for f in os.listdir(tmp_path) if ".tmp" in f]
    # ... and the input can never be its own output
    before = open(src, "rb").read()
    with pytest.raises(binding.VitxError):
        binding.quantize_file(src, src, 2)
    assert open(src, "rb").read() == before


def _write_raw(path, hp7, labels, tensors):
    with open(path, "wb") as f:
        f.write(struct.pack("<i", 0x67676D6C))
        for v in hp7: f.write(struct.pack("<i", v))
        f.write(struct.pack("<i", len(labels)))
        for k, s in labels.items():
            b = s.encode(); f.write(struct.pack("<ii", k, len(b))); f.write(b)
        for name, ttype, ne, data in tensors:
            nb = name.encode()
            f.write(struct.pack("<iii", len(ne), len(nb), ttype))
            for d in ne: f.write(struct.pack("<i", d))
            f.write(nb); f.write(data)


def test_loader_rejects_what_the_reference_rejects(pkg, binding, tmp_path):
    good = pkg.synth.cached_synthetic("vit_micro_patch16_64", head_scale=4.0)
    raw = open(good, "rb").read()
    # missing file (vit.cpp:312-317)
    with pytest.raises(binding.VitxError, match="io error"):
        binding.Model(str(tmp_path / "nope.gguf"))
    # bad magic (vit.cpp:320-328)
    p = str(tmp_path / "magic.gguf"); open(p, "wb").write(b"GGUF" + raw[4:])
    with pytest.raises(binding.VitxError, match="bad magic"):
        binding.Model(p)
    # truncated tensor payload
    p = str(tmp_path / "trunc.gguf"); open(p, "wb").write(raw[:-1000])
    with pytest.raises(binding.VitxError):
        binding.Model(p)
    # missing tensors (vit.cpp:697-701)
    mf = pkg.ggml_file.read_model(good)
    hp7 = [128, 2, 2, 10, 16, 64, 1]
    tens = [(t.name, t.ttype, t.ne, t.raw) for t in mf.tensors]
    p = str(tmp_path / "few.gguf"); _write_raw(p, hp7, {}, tens[:-1])
    with pytest.raises(binding.VitxError, match="tensors were expected"):
        binding.Model(p)
    # unknown tensor name (vit.cpp:618-622)
    bad = list(tens); bad[0] = ("not_a_tensor", bad[0][1], bad[0][2], bad[0][3])
    p = str(tmp_path / "unk.gguf"); _write_raw(p, hp7, {}, bad)
    with pytest.raises(binding.VitxError, match="unknown tensor"):
        binding.Model(p)
    # wrong shape with the right element count (vit.cpp:634-641)
    idx = [i for i, t in enumerate(tens) if t[0] == "blocks.0.attn.proj.weight"][0]
    bad = list(tens); n, tt, ne, data = bad[idx]; bad[idx] = (n, tt, (ne[0] * 2, ne[1] // 2), data)
    p = str(tmp_path / "shape.gguf"); _write_raw(p, hp7, {}, bad)
    with pytest.raises(binding.VitxError, match="wrong shape"):
        binding.Model(p)
    # bad ftype in the header (vit.cpp:408-413)
    p = str(tmp_path / "ftype.gguf"); _write_raw(p, [128, 2, 2, 10, 16, 64, 5], {}, tens)
    with pytest.raises(binding.VitxError, match="bad ftype"):
        binding.Model(p)
    # f32 patch kernel ("--ftype 0" files): rejected by the reference (vit.cpp:515,680), accepted here by design
    pf32 = str(tmp_path / "f32.gguf")
    pkg.synth.write_synthetic(pf32, "vit_micro_patch16_64", ftype=0)
    pkg.ggml_file.write_model(pf32, pkg.synth.hparams_for("vit_micro_patch16_64", 0), pkg.synth.make_weights(pkg.synth.hparams_for("vit_micro_patch16_64")), ftype=0, patch_f16=False)
    assert binding.Model(pf32).hparams.ftype == 0


def _decode(name):
    from PIL import Image
    return np.asarray(Image.open(os.path.join(GOLD, "assets", name)).convert("RGB"), dtype=np.uint8)


def test_preprocess_bicubic_matches_oracle_on_bundled_assets(pkg, binding, oracle):
    """vitx_preprocess_u8 == the oracle's restatement of vit_image_preprocess_bicubic, bit for bit, on all 10 assets."""
    gold = np.load(os.path.join(GOLD, "preprocess_bicubic.npz"))
    for a in sorted(os.listdir(os.path.join(GOLD, "assets"))):
        img = _decode(a)
        got = binding.preprocess(img, 224, binding.BICUBIC)
        assert np.array_equal(got, oracle.preprocess(img, 224, "bicubic")), a
        assert np.array_equal(got, pkg.synth.normalize_u8(gold[a])), a


@pytest.mark.parametrize("shape", [(37, 53), (224, 224), (500, 31), (1, 1), (640, 480)])
def test_preprocess_edge_shapes(binding, oracle, shape):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    img = rng.integers(0, 256, size=(shape[0], shape[1], 3), dtype=np.uint8)
    for mode, code in (("bicubic", binding.BICUBIC), ("bilinear", binding.BILINEAR)):
        for S in (64, 224):
            assert np.array_equal(binding.preprocess(img, S, code), oracle.preprocess(img, S, mode)), (mode, S)
    with pytest.raises(binding.VitxError):
        binding.preprocess(img, 224, 7)        # unknown interpolation -> false in the reference (vit.cpp:300-304)


def test_topk_orders_like_vit_predict(binding):
    p = np.array([0.1, 0.5, 0.05, 0.3, 0.05], np.float32)
    idx, val = binding.topk(p, 3)
    assert idx == [1, 3, 0] and np.allclose(val, [0.5, 0.3, 0.1])
    idx, _ = binding.topk(p, 10)
    assert idx[:3] == [1, 3, 0] and len(idx) == 5 and idx[3:] == [2, 4]     # ties: lower class id first


def test_context_creation_fails_loudly_without_gpu(pkg, binding):
    """No silent CPU fallback: without a HIP device the context cannot be created."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    m = binding.Model(pkg.synth.cached_synthetic("vit_micro_patch16_64", head_scale=4.0))
    with pytest.raises(binding.VitxError, match="HIP"):
        binding.Context(m, 0, 1)


def test_gflop_accounting_matches_baseline_md(pkg):
    assert abs(pkg.synth.gflop_per_image(pkg.synth.hparams_for("vit_base_patch16_224")) - 35.1277) < 1e-3
    assert abs(pkg.synth.gflop_per_image(pkg.synth.hparams_for("vit_tiny_patch16_224")) - 2.5074) < 1e-3
    assert abs(pkg.synth.gflop_per_image(pkg.synth.hparams_for("vit_large_patch16_384")) - 382.1326) < 1e-3


def _build_example(tmp_path):
    import subprocess
    exe = str(tmp_path / "vit_main")
    pkgdir = os.path.join(ROOT, "vit.cpp_amd")
    cmd = ["g++", "-std=c++17", "-O1", os.path.join(ROOT, "examples", "vit_main.cpp"), "-I" + pkgdir, "-L" + pkgdir, "-lvitx", "-L/opt/rocm/lib",
           "-Wl,-rpath," + pkgdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return exe


def test_cpp_example_main_compiles_against_the_drop_in_header(pkg, tmp_path):
    """examples/vit_main.cpp = the reference's main.cpp flow on vit.cpp_amd/vit.h: it must compile and link with plain g++
    (source compatibility of the mirror header), keep main.cpp's exit code 1 for a missing model / image (main.cpp:57-73),
    and -- on a box without a GPU -- fail loudly at vit_predict instead of falling back to anything."""
    import subprocess
    exe = _build_example(tmp_path)
    r = subprocess.run([exe, "-m", str(tmp_path / "none.gguf"), "-i", "x.ppm"], capture_output=True, text=True)
    assert r.returncode == 1 and "failed to load model" in r.stderr
    model = pkg.synth.cached_synthetic("vit_micro_patch16_64", head_scale=4.0)
    r = subprocess.run([exe, "-m", model, "-i", str(tmp_path / "none.ppm")], capture_output=True, text=True)
    assert r.returncode == 1 and "failed to load image" in r.stderr
    ppm = tmp_path / "img.ppm"
    rng = np.random.default_rng(3)
    with open(ppm, "wb") as f:
        f.write(b"P6\n# comment line\n40 30\n255\n"); f.write(rng.integers(0, 256, size=(30, 40, 3), dtype=np.uint8).tobytes())
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, "-m", model, "-i", str(ppm)], capture_output=True, text=True)
        assert r.returncode == 1 and "loaded image" in r.stderr and "(40 x 30)" in r.stderr
        assert "no HIP device" in r.stderr or "HIP" in r.stderr
