"""The ViTSTR scene-text extension (/root/reference/extensions/vitstr.cpp), host side: file format, grey preprocess and greedy
decode -- pinned to code compiled from the extension's own sources (oracle/_ref/libvitstr_ref_*.so, built by oracle/build_ref.sh
from vitstr.h / vitstr.cpp line ranges; those two functions need no ggml)."""
import ctypes as C
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
ASSET_DIR = os.path.join(HERE, "golden", "assets")


def _ref(build):
    path = os.path.join(REF_DIR, build)
    if not os.path.exists(path):
        if os.path.exists("/root/reference/extensions/vitstr.cpp/vitstr.cpp"):
            import subprocess
            subprocess.check_call(["bash", os.path.join(ROOT, "oracle", "build_ref.sh")])
        if not os.path.exists(path):
            pytest.skip("oracle/_ref is not built and /root/reference is not present")
    L = C.CDLL(path)
    L.ref_vitstr_preprocess.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.ref_vitstr_decode.restype = C.c_double
    L.ref_vitstr_decode.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_char_p), C.c_char_p, C.c_int]
    return L


def _ref_pre(L, img, S):
    img = np.ascontiguousarray(img, np.uint8); ny, nx = img.shape[:2]
    out = np.empty((S, S), np.float32)
    assert L.ref_vitstr_preprocess(img.ctypes.data_as(C.POINTER(C.c_uint8)), nx, ny, S, out.ctypes.data_as(C.POINTER(C.c_float))) == 0
    return out


def _images():
    rng = np.random.default_rng(5)
    out = [rng.integers(0, 256, (ny, nx, 3), dtype=np.uint8) for ny, nx in [(32, 100), (57, 131), (224, 224), (2, 2), (300, 17), (480, 640)]]
    grey = np.repeat(rng.integers(0, 256, (40, 90, 1), dtype=np.uint8), 3, axis=2)     # r == g == b: 0.299 + 0.587 + 0.114 lands on the truncation edge
    out.append(grey)
    return out


def test_file_format_and_model_kind(pkg, binding, tmp_path):
    p = str(tmp_path / "vs.gguf")
    pkg.synth.write_synthetic(p, "vitstr_micro_patch16_64", head_scale=4.0)
    m = binding.Model(p)
    assert m.in_channels == 1 and m.seq_len == 25
    assert m.label(0) == "[GO]" and m.label(1) == "[s]" and m.label(2) == "!" and m.label(95) == "~"
    shapes = {name: ne for name, _, ne, _ in m.tensors()}
    assert shapes["patch_embed.proj.weight"] == (16, 16, 1, 128)        # vitstr.cpp:482
    m.close()
    q = str(tmp_path / "vit.gguf")
    pkg.synth.write_synthetic(q, "vit_micro_patch16_64", head_scale=4.0)
    m = binding.Model(q)
    assert m.in_channels == 3 and m.seq_len == 0
    m.close()
    # a two-channel patch kernel is neither model: rejected like any other shape mismatch (vit.cpp:659-667)
    hp = pkg.synth.hparams_for("vit_micro_patch16_64")
    w = pkg.synth.make_weights(hp, in_chans=2)
    bad = str(tmp_path / "bad.gguf")
    pkg.ggml_file.write_model(bad, hp, w)
    with pytest.raises(binding.VitxError):
        binding.Model(bad)


def test_preprocess_matches_reference_compiled_code(binding, oracle):
    strict, fma = _ref("libvitstr_ref_strict.so"), _ref("libvitstr_ref_fma.so")
    imgs = _images() + [binding.load_image(os.path.join(ASSET_DIR, f)) for f in sorted(os.listdir(ASSET_DIR))[:4]]
    for img in imgs:
        for S in (224, 64):
            got = binding.preprocess_vitstr(img, S)
            assert got.shape == (S, S) and got.min() >= -1.0 - 1e-6 and got.max() <= 1.0 + 1e-6      # the 4-term blend may land one ulp outside
            assert np.array_equal(got, oracle.preprocess_vitstr(img, S))                 # product == restatement
            assert np.array_equal(got, _ref_pre(strict, img, S))                         # == the extension's own code, no FP contraction
            # built the way the extension's CMakeLists builds it (-O3 -march=native -> FMA contraction) the grey conversion
            # (uint8)(0.299 r + 0.587 g + 0.114 b) truncates differently on exact-integer sums: one grey level = 2/255 after scaling
            assert np.abs(got - _ref_pre(fma, img, S)).max() <= 2.0 / 255.0 + 1e-6
    with pytest.raises(binding.VitxError):
        binding.preprocess_vitstr(np.zeros((1, 5, 3), np.uint8), 32)                    # the reference reads pixel (x + 1, y + 1) unconditionally


def test_greedy_decode_matches_reference_compiled_code(pkg, binding):
    strict = _ref("libvitstr_ref_strict.so")
    labels = pkg.synth.VITSTR_LABELS
    arr = (C.c_char_p * 96)(*[labels[i].encode() for i in range(96)])
    rng = np.random.default_rng(11)
    cases = []
    for k in range(6):
        p = rng.random((25, 96)).astype(np.float32); p /= p.sum(1, keepdims=True)
        if k >= 1: p[3 + 4 * k if 3 + 4 * k < 25 else 24, 1] = 2.0      # "[s]" wins at some position: the text ends there
        if k == 2: p[1, 1] = 3.0                                         # ends immediately: empty text, score 1
        if k == 3: p[2, 7] = p[2, 40] = 5.0                              # a tie: the first maximum wins (strict '>')
        if k == 4: p[:, 1] = 0.0                                         # no "[s]" at all: all 24 positions are emitted
        cases.append(p)
    for p in cases:
        ids, score = binding.vitstr_decode(p)
        buf = C.create_string_buffer(256)
        want_score = strict.ref_vitstr_decode(p.ctypes.data_as(C.POINTER(C.c_float)), 96, 25, arr, buf, 256)
        assert "".join(labels[i] for i in ids) == buf.value.decode()
        assert score == want_score


def test_oracle_sequence_head(pkg, oracle, tmp_path):
    """Oracle restatement of the ViTSTR graph differences (one grey input plane, vitstr.cpp:713-731; head on tokens 0..24, :864-904):
    every one of the 25 rows is a softmax, and row t equals the classifier-style head applied to token t alone."""
    p = str(tmp_path / "vs.gguf")
    pkg.synth.write_synthetic(p, "vitstr_tiny_patch16_224", head_scale=4.0)
    om = oracle.OracleModel(p)
    assert om.in_chans == 1 and om.out_rows == 25
    x = np.random.default_rng(3).standard_normal((2, 224, 224)).astype(np.float32)
    logits, probs, xd = om.forward(x, oracle.REF, dump=True)
    assert logits.shape == (2, 25, 96) and np.abs(probs.sum(-1) - 1).max() < 1e-3
    X = xd[-1].reshape(2, om.N, om.D)
    z = oracle.layernorm(X[:, :25].reshape(50, om.D), om.tensor("norm.weight"), om.tensor("norm.bias"))
    want = om.linear("head.weight", "head.bias", z, oracle.REF).reshape(2, 25, 96)
    assert np.array_equal(want, logits)


def test_vitstr_graph_vs_transformers_f32_and_converter(pkg, oracle, binding, tmp_path):
    """Independent implementation of the ViTSTR graph: a HuggingFace ViT with ONE input channel (f32, tanh-GELU, eps 1e-6); ViTSTR's
    forward (extensions/vitstr.cpp/convert-pth-to-ggml.py: forward_features, then the head on x[:, :25]) is the final LayerNorm + the
    classifier applied to the first 25 tokens.  The converter's --vitstr path writes it as a ViTSTR file; the oracle's no-rounding
    mode must reproduce the HF logits of all 25 positions, and the product loader must recognise the model kind and labels."""
    torch = pytest.importorskip("torch")
    tr = pytest.importorskip("transformers")
    torch.manual_seed(3)
    cfg = tr.ViTConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512, hidden_act="gelu_pytorch_tanh",
                       layer_norm_eps=1e-6, image_size=96, patch_size=16, num_channels=1, num_labels=96, hidden_dropout_prob=0.0,
                       attention_probs_dropout_prob=0.0, qkv_bias=True)
    m = tr.ViTForImageClassification(cfg).eval()
    with torch.no_grad():
        for p_ in m.parameters():
            p_.add_(0.02 * torch.randn_like(p_))
    path = str(tmp_path / "vitstr_hf.gguf")
    hp = pkg.convert.convert_hf_model(m, path, ftype=0, vitstr=True)
    assert (hp.hidden_size, hp.num_classes, hp.img_size) == (128, 96, 96)
    pm = binding.Model(path)
    assert pm.in_channels == 1 and pm.seq_len == 25 and pm.label(1) == "[s]" and pm.label(34) == "A"
    x = np.clip(np.random.default_rng(9).standard_normal((2, 96, 96)) * 0.5, -1, 1).astype(np.float32)
    with torch.no_grad():
        hidden = m.vit(pixel_values=torch.from_numpy(x)[:, None]).last_hidden_state          # [2, 37, 128], final LayerNorm applied
        hf = m.classifier(hidden[:, :25]).numpy()
    lg, pr = oracle.OracleModel(path).forward(x, oracle.IDEAL)
    assert lg.shape == (2, 25, 96)
    assert np.abs(lg - hf).max() <= 2e-3, np.abs(lg - hf).max()       # the patch kernel is stored in fp16: the only parameter difference
    with pytest.raises(ValueError):
        pkg.convert.convert_hf_model(tr.ViTForImageClassification(tr.ViTConfig(hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=256,
                                                                              image_size=32, patch_size=16, num_labels=96)).eval(), str(tmp_path / "x.gguf"), vitstr=True)
