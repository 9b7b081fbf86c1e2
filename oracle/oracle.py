"""ctypes wrapper over oracle/libvit_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see oracle/vit_oracle.c header).  PARITY UNPINNED: the reference has
no golden vectors and cannot be built offline; see DESIGN.md "Oracle".
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvit_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "vit_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libvit_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Mode(C.Structure):
    _fields_ = [("act_round", C.c_int), ("lut", C.c_int), ("attn_round", C.c_int), ("w_round", C.c_int), ("quant_act", C.c_int),
                ("dot_exact", C.c_int), ("attn_qk_round", C.c_int), ("attn_v_round", C.c_int)]


@dataclass(frozen=True)
class Mode:
    act_round: int = 1
    lut: int = 1
    attn_round: int = 0
    w_round: int = 0
    quant_act: int = 1
    dot_exact: int = 0        # probe: double-accumulated dot products
    attn_qk_round: int = -1   # probe: override attn_round for q,k only
    attn_v_round: int = -1    # probe: override attn_round for v only

    def c(self) -> _Mode:
        return _Mode(self.act_round, self.lut, self.attn_round, self.w_round, self.quant_act,
                     self.dot_exact, self.attn_qk_round, self.attn_v_round)


REF = Mode()                                   # ggml CPU semantics (the reference)
IDEAL = Mode(act_round=0, lut=0, quant_act=0)  # exact f32 everywhere
GPU_F16 = Mode(act_round=1, lut=1, attn_round=1, quant_act=0)   # what the fp16 HIP engine computes
GPU_BF16 = Mode(act_round=2, lut=2, attn_round=2, w_round=2, quant_act=0)

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        L.oracle_model_load.restype = C.c_void_p; L.oracle_model_load.argtypes = [C.c_char_p]
        L.oracle_model_free.argtypes = [C.c_void_p]
        L.oracle_model_hparams.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.oracle_forward.restype = C.c_int
        L.oracle_forward.argtypes = [C.c_void_p, fp, C.c_int, C.POINTER(_Mode), fp, fp, fp]
        L.oracle_layernorm.argtypes = [fp, fp, fp, fp, C.c_int, C.c_int, C.c_float]
        L.oracle_gelu.argtypes = [fp, C.c_size_t, C.c_int]
        L.oracle_softmax_rows.argtypes = [fp, C.c_int, C.c_int, C.c_int]
        L.oracle_attention.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_Mode)]
        L.oracle_patch_embed.argtypes = [C.c_void_p, fp, fp, C.c_int, C.POINTER(_Mode)]
        L.oracle_layer.argtypes = [C.c_void_p, C.c_int, fp, C.c_int, C.POINTER(_Mode)]
        L.oracle_head.argtypes = [C.c_void_p, fp, fp, fp, C.c_int, C.POINTER(_Mode)]
        L.oracle_linear_named.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, fp, fp, C.c_int, C.POINTER(_Mode)]
        L.oracle_tensor_f32.restype = fp; L.oracle_tensor_f32.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]
        L.oracle_preprocess_bicubic.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, fp]
        L.oracle_preprocess_bilinear.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, fp]
        L.oracle_preprocess_vitstr.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, fp]
        L.oracle_model_in_chans.argtypes = [C.c_void_p]; L.oracle_model_out_rows.argtypes = [C.c_void_p]
        L.oracle_num_threads.restype = C.c_int
        L.oracle_set_num_threads.restype = C.c_int; L.oracle_set_num_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def _fp(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


class OracleModel:
    def __init__(self, path: str):
        self._h = lib().oracle_model_load(path.encode())
        if not self._h:
            raise RuntimeError(f"oracle: failed to load {path}")
        hp = (C.c_int * 7)(); lib().oracle_model_hparams(self._h, hp)
        self.D, self.L, self.H, self.C, self.P, self.S, self.ftype = list(hp)
        self.g = self.S // self.P; self.N = self.g * self.g + 1
        self.in_chans = lib().oracle_model_in_chans(self._h)       # 3, or 1 for a ViTSTR file (extensions/vitstr.cpp)
        self.out_rows = lib().oracle_model_out_rows(self._h)       # head rows per image: 1 (cls token) or 25 (ViTSTR)

    def close(self):
        if self._h:
            lib().oracle_model_free(self._h); self._h = None

    def __del__(self):
        try: self.close()
        except Exception: pass

    def forward(self, img_hwc: np.ndarray, mode: Mode = REF, dump: bool = False):
        """img_hwc: [n,S,S,3] f32 normalised ([n,S,S] grey for a ViTSTR file).  Returns (logits [n,C], probs [n,C][, x_dump
        [L+1,n*N,D]]); a ViTSTR file gives [n,25,C]."""
        img = np.ascontiguousarray(img_hwc, np.float32)
        n = img.shape[0]; assert img.shape[1:] == ((self.S, self.S, 3) if self.in_chans == 3 else (self.S, self.S))
        shape = (n, self.C) if self.out_rows == 1 else (n, self.out_rows, self.C)
        logits = np.empty(shape, np.float32); probs = np.empty(shape, np.float32)
        xd = np.empty((self.L + 1, n * self.N, self.D), np.float32) if dump else None
        md = mode.c()
        rc = lib().oracle_forward(self._h, _fp(img), n, C.byref(md), _fp(logits), _fp(probs), _fp(xd) if dump else None)
        if rc: raise RuntimeError("oracle_forward failed")
        return (logits, probs, xd) if dump else (logits, probs)

    def tensor(self, name: str) -> np.ndarray:
        ne = (C.c_int64 * 4)()
        p = lib().oracle_tensor_f32(self._h, name.encode(), ne)
        if not p: raise KeyError(name)
        shape = tuple(int(v) for v in reversed(list(ne)))
        return np.ctypeslib.as_array(p, shape=shape).copy()

    def linear(self, wname: str, bname: str | None, x: np.ndarray, mode: Mode = REF) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32); M = x.shape[0]
        ne = (C.c_int64 * 4)(); lib().oracle_tensor_f32(self._h, wname.encode(), ne)
        y = np.empty((M, int(ne[1])), np.float32); md = mode.c()
        lib().oracle_linear_named(self._h, wname.encode(), bname.encode() if bname else None, _fp(x), _fp(y), M, C.byref(md))
        return y

    def patch_embed(self, img_hwc: np.ndarray, mode: Mode = REF) -> np.ndarray:
        img = np.ascontiguousarray(img_hwc, np.float32); n = img.shape[0]
        X = np.empty((n * self.N, self.D), np.float32); md = mode.c()
        lib().oracle_patch_embed(self._h, _fp(img), _fp(X), n, C.byref(md))
        return X

    def layer(self, il: int, X: np.ndarray, n_img: int, mode: Mode = REF) -> np.ndarray:
        X = np.array(X, np.float32, order="C"); md = mode.c()
        lib().oracle_layer(self._h, il, _fp(X), n_img, C.byref(md))
        return X

    def head(self, X: np.ndarray, n_img: int, mode: Mode = REF):
        X = np.ascontiguousarray(X, np.float32); md = mode.c()
        shape = (n_img, self.C) if self.out_rows == 1 else (n_img, self.out_rows, self.C)
        logits = np.empty(shape, np.float32); probs = np.empty(shape, np.float32)
        lib().oracle_head(self._h, _fp(X), _fp(logits), _fp(probs), n_img, C.byref(md))
        return logits, probs


def layernorm(x, w, b, eps=1e-6):
    x = np.ascontiguousarray(x, np.float32); y = np.empty_like(x)
    lib().oracle_layernorm(_fp(x), _fp(np.ascontiguousarray(w, np.float32)), _fp(np.ascontiguousarray(b, np.float32)), _fp(y), x.shape[0], x.shape[1], eps)
    return y


def gelu(x, lut=1):
    y = np.array(x, np.float32, order="C"); lib().oracle_gelu(_fp(y), y.size, lut); return y


def softmax_rows(x, lut=1):
    y = np.array(x, np.float32, order="C"); lib().oracle_softmax_rows(_fp(y), y.shape[0], y.shape[1], lut); return y


def attention(qkv, n_img, N, D, H, mode: Mode = REF):
    qkv = np.ascontiguousarray(qkv, np.float32); out = np.empty((n_img * N, D), np.float32); md = mode.c()
    lib().oracle_attention(_fp(qkv), _fp(out), n_img, N, D, H, C.byref(md)); return out


def preprocess(img_u8: np.ndarray, S: int, mode: str = "bicubic") -> np.ndarray:
    img = np.ascontiguousarray(img_u8, np.uint8); ny, nx = img.shape[:2]
    assert nx * ny * 3 < (1 << 24), "reference computes the pixel index in float (vit.cpp:260-263)"
    out = np.empty((S, S, 3), np.float32)
    fn = lib().oracle_preprocess_bicubic if mode == "bicubic" else lib().oracle_preprocess_bilinear
    fn(img.ctypes.data_as(C.POINTER(C.c_uint8)), nx, ny, S, _fp(out))
    return out


def preprocess_vitstr(img_u8: np.ndarray, S: int) -> np.ndarray:
    """vit_image_preprocess of extensions/vitstr.cpp: HWC u8 RGB -> [S,S] f32 grey in [-1, 1]."""
    img = np.ascontiguousarray(img_u8, np.uint8); ny, nx = img.shape[:2]
    out = np.empty((S, S), np.float32)
    lib().oracle_preprocess_vitstr(img.ctypes.data_as(C.POINTER(C.c_uint8)), nx, ny, S, _fp(out))
    return out


def num_threads() -> int:
    return lib().oracle_num_threads()


def set_num_threads(n: int) -> int:
    """Threads of the following oracle calls (the reference's vit_params.n_threads); returns the previous value."""
    return lib().oracle_set_num_threads(int(n))
