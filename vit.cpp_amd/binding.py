"""ctypes binding over the C ABI of libvitx.so (include/vitx.h).

Plumbing for tests/ and bench.py: PyTorch provides device memory and streams,
every computation happens inside libvitx.so.  There is NO fallback: if the HIP
library is missing or no GPU is present, the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# VITX_LIB: development override (the -DVITX_LAB laboratory build, A/B builds).  bench.py marks its line invalid when it is set.
LIB_PATH = os.environ.get("VITX_LIB") or os.path.join(_HERE, "libvitx.so")

F16, BF16 = 0, 1
LN_TEST_KEY = 0x7e570000        # vitx_ctx_options::ln_test is honoured only as LN_TEST_KEY | mode
BICUBIC, BILINEAR = 0, 1
EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESID, EPI_BIAS_F32, EPI_PATCH, EPI_BIAS_HILO = 0, 1, 2, 3, 4, 5
GEMM_AUTO, GEMM_PP, GEMM_AUTO_SPLIT = 0, 1, 2          # vitx_op_gemm_ex `kernel` (or a ring configuration: 945, 445, 245, 122)
ATTN_AUTO, ATTN_SINGLE, ATTN_FLOW, ATTN_PERSIST, ATTN_STREAM = 0, 1, 3, 4, 5   # vitx_op_attention_ex `kernel`

EXPORTS = [
    "vitx_status_str", "vitx_last_error", "vitx_model_load", "vitx_model_free", "vitx_model_uid", "vitx_model_hparams", "vitx_model_num_labels",
    "vitx_model_label", "vitx_model_num_tensors", "vitx_model_tensor_info", "vitx_model_tensor_f32", "vitx_quantize_file", "vitx_image_load", "vitx_image_decode", "vitx_image_free", "vitx_preprocess_u8", "vitx_preprocess_u8_device",
    "vitx_ctx_create", "vitx_ctx_create_ex", "vitx_ctx_free", "vitx_ctx_max_batch", "vitx_forward", "vitx_forward_device", "vitx_ctx_synchronize",
    "vitx_topk", "vitx_group_create", "vitx_group_free", "vitx_group_num_devices", "vitx_group_forward", "vitx_group_out_floats", "vitx_group_forward_device", "vitx_group_result", "vitx_group_result_rows", "vitx_profile_enable", "vitx_profile_read", "vitx_profile_bracket_us", "vitx_op_layernorm", "vitx_op_gemm", "vitx_op_gemm_ex", "vitx_op_attention", "vitx_op_attention_ex", "vitx_op_softmax", "vitx_op_softmax_dt", "vitx_trace_enable", "vitx_trace_read",
    "vitx_op_dequant", "vitx_op_gemm_q4", "vitx_ctx_weight_bytes", "vitx_ctx_shares_weights", "vitx_probe_mfma", "vitx_op_gemm_ln", "vitx_ctx_ln_fallbacks", "vitx_ctx_stream_retries",
    "vitx_model_in_channels", "vitx_model_seq_len", "vitx_ctx_out_rows", "vitx_ctx_split", "vitx_ctx_ln_fusion_active", "vitx_op_attention_f32", "vitx_op_attention_planes", "vitx_op_attention_cls", "vitx_preprocess_vitstr_u8", "vitx_vitstr_decode",
]


class HParams(C.Structure):
    _fields_ = [("hidden_size", C.c_int32), ("num_hidden_layers", C.c_int32), ("num_attention_heads", C.c_int32), ("num_classes", C.c_int32),
                ("patch_size", C.c_int32), ("img_size", C.c_int32), ("ftype", C.c_int32), ("eps", C.c_float)]


class CtxOptions(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("streams", C.c_int32), ("graph", C.c_int32), ("quant_on_host", C.c_int32), ("q4_fused_rows", C.c_int32),
                ("split_first", C.c_int32), ("no_ln_fusion", C.c_int32), ("ln_test", C.c_int32), ("f16_fast_attention", C.c_int32), ("last_layer_all_rows", C.c_int32)]


class ProfEntry(C.Structure):
    _fields_ = [("name", C.c_char_p), ("launches", C.c_int32), ("total_ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double), ("busy_ms", C.c_double)]


class VitxError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile libvitx.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", _HERE, "-j8"] + (["-B"] if force else [])
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VitxError(f"{LIB_PATH} is missing: run __graft_entry__.build() (there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        vp, ip = C.c_void_p, C.c_int
        L.vitx_status_str.restype = C.c_char_p; L.vitx_status_str.argtypes = [ip]
        L.vitx_last_error.restype = C.c_char_p
        L.vitx_model_load.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.vitx_model_free.argtypes = [vp]
        L.vitx_model_hparams.argtypes = [vp, C.POINTER(HParams)]
        L.vitx_model_num_labels.argtypes = [vp]
        L.vitx_model_label.restype = C.c_char_p; L.vitx_model_label.argtypes = [vp, ip]
        L.vitx_model_num_tensors.argtypes = [vp]
        L.vitx_model_tensor_info.argtypes = [vp, ip, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_size_t)]
        L.vitx_model_tensor_f32.argtypes = [vp, ip, C.POINTER(C.c_float), C.c_size_t]
        L.vitx_quantize_file.argtypes = [C.c_char_p, C.c_char_p, ip]
        L.vitx_image_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(ip), C.POINTER(ip)]
        L.vitx_image_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(ip), C.POINTER(ip)]
        L.vitx_image_free.argtypes = [C.POINTER(C.c_uint8)]
        L.vitx_preprocess_u8.argtypes = [C.POINTER(C.c_uint8), ip, ip, ip, ip, C.POINTER(C.c_float)]
        L.vitx_preprocess_u8_device.argtypes = [vp, ip, ip, ip, ip, ip, vp, vp]
        L.vitx_ctx_create.argtypes = [vp, ip, ip, ip, C.POINTER(vp)]
        L.vitx_ctx_create_ex.argtypes = [vp, ip, ip, ip, C.POINTER(CtxOptions), C.POINTER(vp)]
        L.vitx_ctx_free.argtypes = [vp]
        L.vitx_ctx_max_batch.argtypes = [vp]
        L.vitx_forward.argtypes = [vp, C.POINTER(C.c_float), ip, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.vitx_forward_device.argtypes = [vp, vp, ip, vp, vp, vp]
        L.vitx_ctx_synchronize.argtypes = [vp]
        L.vitx_group_create.argtypes = [vp, C.POINTER(C.c_int), ip, ip, ip, C.POINTER(vp)]
        L.vitx_group_free.argtypes = [vp]
        L.vitx_group_num_devices.argtypes = [vp]
        L.vitx_group_forward.argtypes = [vp, C.POINTER(C.c_float), ip, C.POINTER(C.c_float)]
        L.vitx_group_out_floats.argtypes = [vp]
        L.vitx_group_forward_device.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(ip), ip]
        L.vitx_group_result.restype = C.c_void_p; L.vitx_group_result.argtypes = [vp, ip]
        L.vitx_group_result_rows.argtypes = [vp]
        L.vitx_topk.argtypes = [C.POINTER(C.c_float), ip, ip, C.POINTER(C.c_int32), C.POINTER(C.c_float)]
        L.vitx_profile_enable.argtypes = [vp, ip]
        L.vitx_profile_read.argtypes = [vp, C.POINTER(ProfEntry), ip, C.POINTER(ip)]
        if hasattr(L, "vitx_profile_bracket_us"):      # (tools/ab_libs.py also loads builds that predate it)
            L.vitx_profile_bracket_us.argtypes = [vp, C.POINTER(C.c_double)]
        L.vitx_op_layernorm.argtypes = [ip, vp, vp, vp, vp, ip, ip, C.c_float, vp]
        L.vitx_op_gemm.argtypes = [ip, ip, vp, vp, vp, vp, ip, ip, ip, vp]
        L.vitx_op_attention.argtypes = [ip, vp, vp, ip, ip, ip, ip, vp]
        L.vitx_op_softmax.argtypes = [vp, vp, ip, ip, ip, vp]
        L.vitx_op_attention_ex.argtypes = [ip, ip, vp, vp, ip, ip, ip, ip, vp]
        L.vitx_op_softmax_dt.argtypes = [ip, vp, vp, ip, ip, ip, vp]
        L.vitx_op_gemm_ex.argtypes = [ip, ip, ip, vp, vp, vp, vp, vp, ip, ip, ip, ip, ip, vp]
        L.vitx_trace_enable.argtypes = [vp, C.POINTER(C.c_int32), ip]
        L.vitx_trace_read.argtypes = [vp, C.POINTER(C.c_float), C.c_size_t]
        L.vitx_op_dequant.argtypes = [ip, ip, vp, vp, vp, ip, ip, ip, vp]
        L.vitx_op_gemm_q4.argtypes = [ip, ip, vp, vp, vp, vp, vp, ip, ip, ip, ip, vp]
        L.vitx_ctx_weight_bytes.restype = C.c_size_t; L.vitx_ctx_weight_bytes.argtypes = [vp]
        L.vitx_ctx_shares_weights.restype = C.c_int; L.vitx_ctx_shares_weights.argtypes = [vp]
        L.vitx_ctx_ln_fallbacks.restype = C.c_longlong; L.vitx_ctx_ln_fallbacks.argtypes = [vp]
        L.vitx_ctx_stream_retries.argtypes = [vp]
        L.vitx_op_gemm_ln.argtypes = [ip, vp, vp, vp, vp, vp, vp, vp, ip, ip, ip, C.c_float, ip, ip, C.POINTER(ip), vp]
        L.vitx_probe_mfma.argtypes = [ip, ip, ip, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.vitx_model_in_channels.argtypes = [vp]; L.vitx_model_seq_len.argtypes = [vp]; L.vitx_ctx_out_rows.argtypes = [vp]
        L.vitx_ctx_split.argtypes = [vp, ip, C.POINTER(C.c_int32), ip]
        L.vitx_ctx_ln_fusion_active.argtypes = [vp]
        L.vitx_op_attention_f32.argtypes = [vp, vp, ip, ip, ip, ip, vp]
        L.vitx_op_attention_planes.argtypes = [vp, C.c_long, vp, ip, ip, ip, ip, vp]
        if hasattr(L, "vitx_op_attention_cls"):
            L.vitx_op_attention_cls.argtypes = [ip, vp, C.c_long, vp, ip, ip, ip, ip, vp]
        L.vitx_preprocess_vitstr_u8.argtypes = [C.POINTER(C.c_uint8), ip, ip, ip, C.POINTER(C.c_float)]
        L.vitx_vitstr_decode.argtypes = [C.POINTER(C.c_float), ip, ip, C.POINTER(C.c_int32), C.POINTER(ip), C.POINTER(C.c_double)]
        _lib = L
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        L = lib()
        raise VitxError(f"{what}: {L.vitx_status_str(rc).decode()} ({rc}): {L.vitx_last_error().decode()}")


class Model:
    """Parsed weight file (vit_model_load, vit.cpp:308-712)."""

    def __init__(self, path: str):
        self._h = C.c_void_p()
        check(lib().vitx_model_load(path.encode(), C.byref(self._h)), f"vitx_model_load({path})")
        hp = HParams(); lib().vitx_model_hparams(self._h, C.byref(hp))
        self.hparams = hp
        self.path = path

    def close(self):
        if getattr(self, "_h", None) and self._h:
            lib().vitx_model_free(self._h); self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown: module globals may already be gone
            pass

    @property
    def in_channels(self) -> int: return lib().vitx_model_in_channels(self._h)      # 3, or 1 for a ViTSTR file
    @property
    def seq_len(self) -> int: return lib().vitx_model_seq_len(self._h)              # 0 (classifier) or 25 (ViTSTR: rows per image)
    @property
    def num_classes(self) -> int: return self.hparams.num_classes
    @property
    def img_size(self) -> int: return self.hparams.img_size

    def label(self, i: int) -> Optional[str]:
        s = lib().vitx_model_label(self._h, i)
        return s.decode() if s is not None else None

    def tensors(self) -> List[Tuple[str, int, Tuple[int, ...], int]]:
        out = []
        for i in range(lib().vitx_model_num_tensors(self._h)):
            name = C.c_char_p(); t = C.c_int32(); ne = (C.c_int64 * 4)(); nb = C.c_size_t()
            check(lib().vitx_model_tensor_info(self._h, i, C.byref(name), C.byref(t), ne, C.byref(nb)))
            out.append((name.value.decode(), t.value, tuple(ne), nb.value))
        return out

    def tensor_f32(self, index: int) -> np.ndarray:
        _, _, ne, _ = self.tensors()[index]
        n = int(np.prod(ne)); a = np.empty(n, np.float32)
        check(lib().vitx_model_tensor_f32(self._h, index, a.ctypes.data_as(C.POINTER(C.c_float)), n))
        return a.reshape(tuple(reversed(ne)))


def quantize_file(path_in: str, path_out: str, ftype: int) -> None:
    """Native `quantize` (quantize.cpp:34-353): f16/f32 file -> q4_0/q4_1/q5_0/q5_1/q8_0 file."""
    check(lib().vitx_quantize_file(path_in.encode(), path_out.encode(), ftype), "vitx_quantize_file")


def load_image(path: str) -> np.ndarray:
    """load_image_from_file (vit.cpp:109-127): JPEG / PNG / PPM file -> HWC u8 RGB, decoded by libvitx.so itself."""
    data = C.POINTER(C.c_uint8)(); nx = C.c_int(); ny = C.c_int()
    check(lib().vitx_image_load(path.encode(), C.byref(data), C.byref(nx), C.byref(ny)), f"vitx_image_load({path})")
    try:
        return np.ctypeslib.as_array(data, shape=(ny.value, nx.value, 3)).copy()
    finally:
        lib().vitx_image_free(data)


def decode_image(blob: bytes) -> np.ndarray:
    data = C.POINTER(C.c_uint8)(); nx = C.c_int(); ny = C.c_int()
    check(lib().vitx_image_decode(blob, len(blob), C.byref(data), C.byref(nx), C.byref(ny)), "vitx_image_decode")
    try:
        return np.ctypeslib.as_array(data, shape=(ny.value, nx.value, 3)).copy()
    finally:
        lib().vitx_image_free(data)


def preprocess(img_u8: np.ndarray, img_size: int, interp: int = BICUBIC) -> np.ndarray:
    """vit_image_preprocess (vit.cpp:289-305): HWC u8 any size -> HWC f32 [S,S,3]."""
    img = np.ascontiguousarray(img_u8, np.uint8); ny, nx = img.shape[:2]
    out = np.empty((img_size, img_size, 3), np.float32)
    check(lib().vitx_preprocess_u8(img.ctypes.data_as(C.POINTER(C.c_uint8)), nx, ny, img_size, interp, out.ctypes.data_as(C.POINTER(C.c_float))), "vitx_preprocess_u8")
    return out


def preprocess_vitstr(img_u8: np.ndarray, img_size: int) -> np.ndarray:
    """vit_image_preprocess of extensions/vitstr.cpp (vitstr.cpp:135-201): HWC u8 RGB -> [S,S] f32 grey in [-1, 1]."""
    img = np.ascontiguousarray(img_u8, np.uint8); ny, nx = img.shape[:2]
    out = np.empty((img_size, img_size), np.float32)
    check(lib().vitx_preprocess_vitstr_u8(img.ctypes.data_as(C.POINTER(C.c_uint8)), nx, ny, img_size, out.ctypes.data_as(C.POINTER(C.c_float))), "vitx_preprocess_vitstr_u8")
    return out


def vitstr_decode(probs: np.ndarray):
    """Greedy decode of one image's [25, C] probabilities (vitstr.cpp:1025-1051) -> (class ids, score)."""
    p = np.ascontiguousarray(probs, np.float32); R, Cn = p.shape
    ids = (C.c_int32 * R)(); n = C.c_int(); score = C.c_double()
    check(lib().vitx_vitstr_decode(p.ctypes.data_as(C.POINTER(C.c_float)), R, Cn, ids, C.byref(n), C.byref(score)), "vitx_vitstr_decode")
    return list(ids)[:n.value], score.value


def preprocess_device(d_u8: int, n: int, nx: int, ny: int, img_size: int, d_out: int, interp: int = BICUBIC, stream: int = 0) -> None:
    """vit_image_preprocess on the GPU (device pointers; enqueue only)."""
    check(lib().vitx_preprocess_u8_device(d_u8, n, nx, ny, img_size, interp, d_out, stream or None), "vitx_preprocess_u8_device")


def _check_image_shape(model: "Model", x: np.ndarray) -> None:
    """[n, S, S, 3] for a classifier, [n, S, S] (one grey plane) for a ViTSTR file: the C ABI reads n * S * S * in_channels floats."""
    S = model.img_size
    want = (S, S, 3) if model.in_channels == 3 else (S, S)
    if x.ndim != 1 + len(want) or tuple(x.shape[1:]) != want:
        raise ValueError(f"images must be [n, {', '.join(map(str, want))}] for this model, got {tuple(x.shape)}")


class Context:
    """Per-(thread, GPU) execution context (vit_state): weights in HBM + activation scratch."""

    def __init__(self, model: Model, device: int = 0, max_batch: int = 1, dtype: int = F16, **options):
        """options: the fields of vitx_ctx_options (streams, graph, quant_on_host, q4_fused_rows, split_first, no_ln_fusion)."""
        self.model = model; self.device = device; self.max_batch = max_batch; self.dtype = dtype
        self._h = C.c_void_p()
        opt = CtxOptions(struct_size=C.sizeof(CtxOptions))
        for k, v in options.items():
            if k not in dict(CtxOptions._fields_) or k == "struct_size":
                raise TypeError(f"unknown context option {k!r}")
            setattr(opt, k, int(v))
        # struct_size = the smallest prefix that covers every field that is set: a library that predates a trailing field (tools/ab_libs.py loads
        # older builds through VITX_LIB) rejects a struct_size above its own sizeof(vitx_ctx_options) -- zero trailing fields are its defaults anyway
        names = [f for f, _ in CtxOptions._fields_]
        last = max([i for i, f in enumerate(names) if f != "struct_size" and getattr(opt, f) != 0] + [names.index("f16_fast_attention")])
        opt.struct_size = 4 * (last + 1)
        check(lib().vitx_ctx_create_ex(model._h, device, max_batch, dtype, C.byref(opt), C.byref(self._h)), "vitx_ctx_create_ex")

    def close(self):
        if getattr(self, "_h", None) and self._h:
            lib().vitx_ctx_free(self._h); self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forward(self, imgs_hwc: np.ndarray, want_logits: bool = False):
        """Host arrays in/out (copies + sync): [n,S,S,3] f32 -> probs [n,C] (and logits)."""
        x = np.ascontiguousarray(imgs_hwc, np.float32); n = x.shape[0]
        _check_image_shape(self.model, x)
        R = self.model.seq_len                      # ViTSTR: [n, S, S] grey in, [n, 25, C] out
        probs = np.empty((n, self.model.num_classes) if R == 0 else (n, R, self.model.num_classes), np.float32)
        logits = np.empty_like(probs) if want_logits else None
        fp = C.POINTER(C.c_float)
        check(lib().vitx_forward(self._h, x.ctypes.data_as(fp), n, probs.ctypes.data_as(fp), logits.ctypes.data_as(fp) if want_logits else None), "vitx_forward")
        return (probs, logits) if want_logits else probs

    def forward_device(self, d_imgs: int, n: int, d_probs: int, d_logits: int = 0, stream: int = 0) -> None:
        """Device pointers; only enqueues on `stream` (0 = the context's own stream)."""
        check(lib().vitx_forward_device(self._h, d_imgs, n, d_probs, d_logits or None, stream or None), "vitx_forward_device")

    def trace_enable(self, image_ids) -> None:
        """Record the f32 residual stream of these images after the patch embedding and after every layer (vitx_trace_enable)."""
        ids = (C.c_int32 * len(image_ids))(*image_ids)
        check(lib().vitx_trace_enable(self._h, ids, len(image_ids)), "vitx_trace_enable")
        self._trace_n = len(image_ids)

    def trace_read(self) -> np.ndarray:
        """[L + 1, n_ids, tokens, hidden] f32 of the last forward."""
        hp = self.model.hparams
        N = (hp.img_size // hp.patch_size) ** 2 + 1
        out = np.empty((hp.num_hidden_layers + 1, self._trace_n, N, hp.hidden_size), np.float32)
        check(lib().vitx_trace_read(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), out.size), "vitx_trace_read")
        return out

    def synchronize(self) -> None:
        check(lib().vitx_ctx_synchronize(self._h), "vitx_ctx_synchronize")

    def split(self, n: int) -> List[int]:
        """Sizes of the contiguous sub-batches a forward of n images is cut into (vitx_ctx_split); [n] on one stream."""
        m = (C.c_int32 * 4)()
        k = lib().vitx_ctx_split(self._h, n, m, 4)
        if k <= 0:
            raise VitxError(f"vitx_ctx_split({n}) failed")
        return [int(m[i]) for i in range(k)]

    def boundary_rows(self, n: int) -> List[int]:
        """Image ids on either side of every sub-batch boundary of an n-image forward, plus both ends of the batch."""
        ids, off = {0, min(1, n - 1), max(0, n - 2), n - 1}, 0
        for sz in self.split(n)[:-1]:
            off += sz
            ids.update({off - 1, off})
        return sorted(i for i in ids if 0 <= i < n)

    def weight_bytes(self) -> int:
        """Device bytes held by the weight matrices (quantised tensors stay in block form: 4.5 ... 8.5 bits per weight)."""
        return int(lib().vitx_ctx_weight_bytes(self._h))

    def shares_weights(self) -> bool:
        """True when this context attached to a device copy of the weights another context of the same model had uploaded."""
        return bool(lib().vitx_ctx_shares_weights(self._h))

    def ln_fallbacks(self) -> int:
        """GEMM tiles whose fused LayerNorm was left to the fix-up launch since the context was created (vitx_ctx_ln_fallbacks)."""
        return int(lib().vitx_ctx_ln_fallbacks(self._h))

    def ln_fusion_active(self) -> int:
        """1 fused LayerNorms, 0 stand-alone launches, -1 switched off by the fall-back budget (vitx_ctx_ln_fusion_active)."""
        return int(lib().vitx_ctx_ln_fusion_active(self._h))

    def stream_retries(self) -> int:
        """Internal sub-batch streams re-created because they did not run beside the caller's stream (vitx_ctx_stream_retries)."""
        return int(lib().vitx_ctx_stream_retries(self._h))

    def profile_enable(self, on: bool = True) -> None:
        check(lib().vitx_profile_enable(self._h, int(on)))

    def profile_bracket_us(self) -> float:
        """Microseconds one HIP-event bracket adds to a launch (vitx_profile_bracket_us); profile_read() intervals are raw."""
        v = C.c_double()
        check(lib().vitx_profile_bracket_us(self._h, C.byref(v)), "vitx_profile_bracket_us")
        return float(v.value)

    def profile_read(self):
        arr = (ProfEntry * 16)(); n = C.c_int()
        check(lib().vitx_profile_read(self._h, arr, 16, C.byref(n)), "vitx_profile_read")
        return [dict(name=arr[i].name.decode(), launches=arr[i].launches, total_ms=arr[i].total_ms, flops=arr[i].flops, bytes=arr[i].bytes, busy_ms=arr[i].busy_ms) for i in range(n.value)]


class Group:
    """Several GPUs in one process: batch shards + one RCCL all-gather of the probabilities (vitx_group_*)."""

    def __init__(self, model: Model, devices, max_batch_per_device: int, dtype: int = F16):
        self.model = model
        devs = (C.c_int * len(devices))(*devices)
        self._h = C.c_void_p()
        check(lib().vitx_group_create(model._h, devs, len(devices), max_batch_per_device, dtype, C.byref(self._h)), "vitx_group_create")

    def forward(self, imgs_hwc: np.ndarray) -> np.ndarray:
        """Host images in, [n, C] probabilities out ([n, 25, C] for a ViTSTR file: vitx_group_out_floats floats per image)."""
        x = np.ascontiguousarray(imgs_hwc, np.float32); n = x.shape[0]
        _check_image_shape(self.model, x)
        R = self.model.seq_len
        probs = np.empty((n, self.model.num_classes) if R == 0 else (n, R, self.model.num_classes), np.float32)
        assert probs[0].size == lib().vitx_group_out_floats(self._h)
        fp = C.POINTER(C.c_float)
        check(lib().vitx_group_forward(self._h, x.ctypes.data_as(fp), n, probs.ctypes.data_as(fp)), "vitx_group_forward")
        return probs

    def forward_device(self, d_imgs: List[int], n_local: List[int], topk: int = 0) -> Tuple[List[int], int]:
        """Device-resident shards (one device pointer and image count per device).  Returns (per-device pointers to the gathered result,
        n_max): [n_devices][n_max][C] f32, or [n_devices][n_max][rows][topk] {f32, i32} pairs when topk > 0 (vitx_group_forward_device)."""
        nd = lib().vitx_group_num_devices(self._h)
        assert len(d_imgs) == nd and len(n_local) == nd
        ptrs = (C.c_void_p * nd)(*[p or None for p in d_imgs]); cnt = (C.c_int * nd)(*n_local)
        check(lib().vitx_group_forward_device(self._h, ptrs, cnt, topk), "vitx_group_forward_device")
        return [lib().vitx_group_result(self._h, r) for r in range(nd)], lib().vitx_group_result_rows(self._h)

    def close(self):
        if getattr(self, "_h", None) and self._h:
            lib().vitx_group_free(self._h); self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def probe_mfma(device: int = 0, dtype: int = BF16, fill: int = 2, target_ms: float = 150.0) -> Tuple[float, float]:
    """(TFLOP/s, shader MHz) of back-to-back MFMAs on register operands: the power-managed ceiling of the matrix pipe (vitx_probe_mfma)."""
    tf = C.c_double(); mhz = C.c_double()
    check(lib().vitx_probe_mfma(device, dtype, fill, target_ms, C.byref(tf), C.byref(mhz)), "vitx_probe_mfma")
    return tf.value, mhz.value


def topk(probs_row: np.ndarray, k: int = 5):
    p = np.ascontiguousarray(probs_row, np.float32); k = min(k, p.size)
    idx = (C.c_int32 * k)(); val = (C.c_float * k)()
    check(lib().vitx_topk(p.ctypes.data_as(C.POINTER(C.c_float)), p.size, k, idx, val), "vitx_topk")
    return list(idx), list(val)
