"""Legacy-ggml ".gguf" model file: Python writer/reader and block quantisers.

Host-side mirror of the reference's offline tools for the hot path's *input
format* (not part of the product's compute path):
  * writer layout  -- /root/reference/convert-pth-to-ggml.py:105-158
  * reader layout  -- /root/reference/vit.cpp:319-371, 590-695
  * which tensors get quantised and how -- /root/reference/quantize.cpp:207-303
    (2-D tensors whose name matches ".*weight"), block encoders are ggml's
    quantize_row_q{4_0,4_1,5_0,5_1,8_0}_reference.

File (little endian, no padding):
  int32 magic 0x67676d6c ; int32 hidden, layers, heads, classes, patch, img ; int32 ftype
  int32 n_labels ; n_labels x { int32 key ; int32 len ; bytes }
  until EOF: int32 n_dims, name_len, ttype ; int32 ne[n_dims] (reversed torch shape) ; name ; data
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np

GGML_MAGIC = 0x67676D6C
F32, F16, Q4_0, Q4_1, Q5_0, Q5_1, Q8_0 = 0, 1, 2, 3, 6, 7, 8
TYPE_NAMES = {F32: "f32", F16: "f16", Q4_0: "q4_0", Q4_1: "q4_1", Q5_0: "q5_0", Q5_1: "q5_1", Q8_0: "q8_0"}
BLOCK_BYTES = {F32: 4, F16: 2, Q4_0: 18, Q4_1: 20, Q5_0: 22, Q5_1: 24, Q8_0: 34}
BLOCK_ELEMS = {F32: 1, F16: 1, Q4_0: 32, Q4_1: 32, Q5_0: 32, Q5_1: 32, Q8_0: 32}
QK = 32


@dataclass
class HParams:
    hidden_size: int
    num_hidden_layers: int
    num_attention_heads: int
    num_classes: int
    patch_size: int
    img_size: int
    ftype: int = 1

    @property
    def n_tokens(self) -> int:
        g = self.img_size // self.patch_size
        return g * g + 1


@dataclass
class TensorRec:
    name: str
    ttype: int
    ne: Tuple[int, ...]          # ggml order (ne[0] fastest)
    raw: bytes


@dataclass
class ModelFile:
    hparams: HParams
    id2label: Dict[int, str] = field(default_factory=dict)
    tensors: List[TensorRec] = field(default_factory=list)


# --------------------------------------------------------------------------- block encoders
def _f16_bytes(x: np.ndarray) -> np.ndarray:
    return x.astype(np.float16).view(np.uint8)


def quantize_q4_0(x: np.ndarray) -> bytes:
    """quantize_row_q4_0_reference: max = signed value of largest |x|, d = max/-8,
    q = min(15, (int8)(x/d + 8.5f)); low nibbles = first 16, high = last 16."""
    xb = np.ascontiguousarray(x, np.float32).reshape(-1, QK)
    idx = np.argmax(np.abs(xb), axis=1)
    mx = xb[np.arange(xb.shape[0]), idx]
    d = (mx / np.float32(-8.0)).astype(np.float32)
    with np.errstate(divide="ignore"):
        inv = np.where(d != 0, np.float32(1.0) / d, np.float32(0.0)).astype(np.float32)
    q = (xb * inv[:, None] + np.float32(8.5)).astype(np.float32)
    qi = np.minimum(15, q.astype(np.int8).astype(np.int32)).astype(np.uint8)   # C cast truncates toward zero
    qs = (qi[:, :16] | (qi[:, 16:] << 4)).astype(np.uint8)
    out = np.empty((xb.shape[0], 18), np.uint8)
    out[:, 0:2] = _f16_bytes(d).reshape(-1, 2)
    out[:, 2:] = qs
    return out.tobytes()


def quantize_q4_1(x: np.ndarray) -> bytes:
    xb = np.ascontiguousarray(x, np.float32).reshape(-1, QK)
    mn, mx = xb.min(axis=1), xb.max(axis=1)
    d = ((mx - mn) / np.float32(15.0)).astype(np.float32)
    with np.errstate(divide="ignore"):
        inv = np.where(d != 0, np.float32(1.0) / d, np.float32(0.0)).astype(np.float32)
    q = ((xb - mn[:, None]) * inv[:, None] + np.float32(0.5)).astype(np.float32)
    qi = np.minimum(15, q.astype(np.int8).astype(np.int32)).astype(np.uint8)
    qs = (qi[:, :16] | (qi[:, 16:] << 4)).astype(np.uint8)
    out = np.empty((xb.shape[0], 20), np.uint8)
    out[:, 0:2] = _f16_bytes(d).reshape(-1, 2)
    out[:, 2:4] = _f16_bytes(mn).reshape(-1, 2)
    out[:, 4:] = qs
    return out.tobytes()


def _pack_q5(qi: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    qs = ((qi[:, :16] & 0x0F) | ((qi[:, 16:] & 0x0F) << 4)).astype(np.uint8)
    qh = np.zeros(qi.shape[0], np.uint32)
    for j in range(16):
        qh |= ((qi[:, j].astype(np.uint32) & 0x10) >> 4) << (j + 0)
        qh |= ((qi[:, j + 16].astype(np.uint32) & 0x10) >> 4) << (j + 16)
    return qs, qh


def quantize_q5_0(x: np.ndarray) -> bytes:
    xb = np.ascontiguousarray(x, np.float32).reshape(-1, QK)
    idx = np.argmax(np.abs(xb), axis=1)
    mx = xb[np.arange(xb.shape[0]), idx]
    d = (mx / np.float32(-16.0)).astype(np.float32)
    with np.errstate(divide="ignore"):
        inv = np.where(d != 0, np.float32(1.0) / d, np.float32(0.0)).astype(np.float32)
    q = (xb * inv[:, None] + np.float32(16.5)).astype(np.float32)
    qi = np.minimum(31, q.astype(np.int8).astype(np.int32)).astype(np.uint8)
    qs, qh = _pack_q5(qi)
    out = np.empty((xb.shape[0], 22), np.uint8)
    out[:, 0:2] = _f16_bytes(d).reshape(-1, 2)
    out[:, 2:6] = qh.view(np.uint8).reshape(-1, 4)
    out[:, 6:] = qs
    return out.tobytes()


def quantize_q5_1(x: np.ndarray) -> bytes:
    xb = np.ascontiguousarray(x, np.float32).reshape(-1, QK)
    mn, mx = xb.min(axis=1), xb.max(axis=1)
    d = ((mx - mn) / np.float32(31.0)).astype(np.float32)
    with np.errstate(divide="ignore"):
        inv = np.where(d != 0, np.float32(1.0) / d, np.float32(0.0)).astype(np.float32)
    q = ((xb - mn[:, None]) * inv[:, None] + np.float32(0.5)).astype(np.float32)
    qi = q.astype(np.uint8)
    qs, qh = _pack_q5(qi)
    out = np.empty((xb.shape[0], 24), np.uint8)
    out[:, 0:2] = _f16_bytes(d).reshape(-1, 2)
    out[:, 2:4] = _f16_bytes(mn).reshape(-1, 2)
    out[:, 4:8] = qh.view(np.uint8).reshape(-1, 4)
    out[:, 8:] = qs
    return out.tobytes()


def quantize_q8_0(x: np.ndarray) -> bytes:
    xb = np.ascontiguousarray(x, np.float32).reshape(-1, QK)
    amax = np.abs(xb).max(axis=1)
    d = (amax / np.float32(127.0)).astype(np.float32)
    with np.errstate(divide="ignore"):
        inv = np.where(d != 0, np.float32(1.0) / d, np.float32(0.0)).astype(np.float32)
    v = (xb * inv[:, None]).astype(np.float32)
    qi = (np.sign(v) * np.floor(np.abs(v) + np.float32(0.5))).astype(np.int8)   # roundf: half away from zero
    out = np.empty((xb.shape[0], 34), np.uint8)
    out[:, 0:2] = _f16_bytes(d).reshape(-1, 2)
    out[:, 2:] = qi.view(np.uint8)
    return out.tobytes()


QUANTIZERS = {Q4_0: quantize_q4_0, Q4_1: quantize_q4_1, Q5_0: quantize_q5_0, Q5_1: quantize_q5_1, Q8_0: quantize_q8_0}


def dequantize(ttype: int, raw: bytes, n: int) -> np.ndarray:
    """Decode n elements of the given block type to f32 (ggml dequantize_row_*)."""
    if ttype == F32:
        return np.frombuffer(raw, np.float32, n).copy()
    if ttype == F16:
        return np.frombuffer(raw, np.float16, n).astype(np.float32)
    b = np.frombuffer(raw, np.uint8).reshape(-1, BLOCK_BYTES[ttype])
    d = b[:, 0:2].copy().view(np.float16).astype(np.float32)
    out = np.empty((b.shape[0], QK), np.float32)
    if ttype == Q4_0:
        qs = b[:, 2:]
        out[:, :16] = ((qs & 0x0F).astype(np.int32) - 8) * d
        out[:, 16:] = ((qs >> 4).astype(np.int32) - 8) * d
    elif ttype == Q4_1:
        m = b[:, 2:4].copy().view(np.float16).astype(np.float32); qs = b[:, 4:]
        out[:, :16] = (qs & 0x0F).astype(np.float32) * d + m
        out[:, 16:] = (qs >> 4).astype(np.float32) * d + m
    elif ttype in (Q5_0, Q5_1):
        off = 2 if ttype == Q5_0 else 4
        qh = b[:, off:off + 4].copy().view(np.uint32)[:, 0]; qs = b[:, off + 4:]
        lo = np.empty((b.shape[0], 16), np.int32); hi = np.empty((b.shape[0], 16), np.int32)
        for j in range(16):
            lo[:, j] = (qs[:, j] & 0x0F) | (((qh >> j) & 1) << 4)
            hi[:, j] = (qs[:, j] >> 4) | (((qh >> (j + 16)) & 1) << 4)
        if ttype == Q5_0:
            out[:, :16] = (lo - 16) * d; out[:, 16:] = (hi - 16) * d
        else:
            m = b[:, 2:4].copy().view(np.float16).astype(np.float32)
            out[:, :16] = lo * d + m; out[:, 16:] = hi * d + m
    elif ttype == Q8_0:
        out[:] = b[:, 2:].view(np.int8).astype(np.float32) * d
    else:
        raise ValueError(f"unknown tensor type {ttype}")
    return out.reshape(-1)[:n]


# --------------------------------------------------------------------------- write / read
def write_model(path: str, hp: HParams, tensors: Dict[str, np.ndarray], id2label: Dict[int, str] | None = None,
                ftype: int = 1, patch_f16: bool = True) -> None:
    """Write torch-shaped f32 tensors (timm state_dict naming) as the reference converter does:
    1-D tensors, pos_embed and cls_token stay f32; everything else f16 when ftype>=1
    (convert-pth-to-ggml.py:141-148).  ftype in {2,3,6,7,8} additionally quantises the 2-D
    '*weight' tensors exactly like quantize.cpp:207-303 does to an f16 file."""
    id2label = id2label if id2label is not None else {i: f"LABEL_{i}" for i in range(hp.num_classes)}
    with open(path, "wb") as f:
        f.write(struct.pack("<i", GGML_MAGIC))
        for v in (hp.hidden_size, hp.num_hidden_layers, hp.num_attention_heads, hp.num_classes, hp.patch_size, hp.img_size):
            f.write(struct.pack("<i", v))
        f.write(struct.pack("<i", ftype))
        f.write(struct.pack("<i", len(id2label)))
        for k, v in id2label.items():
            b = v.encode("utf-8")
            f.write(struct.pack("<ii", k, len(b))); f.write(b)
        for name, t in tensors.items():
            data = np.asarray(t, np.float32)
            if name == "patch_embed.proj.bias":
                data = data.reshape(1, data.shape[0], 1, 1)              # convert:150-151
            keep_f32 = data.ndim == 1 or name in ("pos_embed", "cls_token") or name == "patch_embed.proj.bias"
            if ftype == 0 and not (patch_f16 and name == "patch_embed.proj.weight"):
                keep_f32 = True
            ttype = F32 if keep_f32 else F16
            if not keep_f32 and ftype in QUANTIZERS and data.ndim == 2 and name.endswith("weight"):
                ttype = ftype
            nb = name.encode("utf-8")
            f.write(struct.pack("<iii", data.ndim, len(nb), ttype))
            for dim in reversed(data.shape):
                f.write(struct.pack("<i", dim))
            f.write(nb)
            if ttype == F32:
                f.write(data.astype("<f4").tobytes())
            elif ttype == F16:
                f.write(data.astype("<f2").tobytes())
            else:
                src = data.astype(np.float16).astype(np.float32)        # quantize.cpp reads the f16 file (:227-235)
                f.write(QUANTIZERS[ttype](src))


def read_model(path: str) -> ModelFile:
    with open(path, "rb") as f:
        buf = f.read()
    off = 0

    def i32() -> int:
        nonlocal off
        v = struct.unpack_from("<i", buf, off)[0]; off += 4
        return v

    if (i32() & 0xFFFFFFFF) != GGML_MAGIC:
        raise ValueError("bad magic")
    vals = [i32() for _ in range(7)]
    hp = HParams(*vals[:6], ftype=vals[6] % 1000)
    mf = ModelFile(hp)
    for _ in range(i32()):
        k, ln = i32(), i32()
        mf.id2label[k] = buf[off:off + ln].decode("utf-8"); off += ln
    while off < len(buf):
        n_dims, name_len, ttype = i32(), i32(), i32()
        ne = tuple(i32() for _ in range(n_dims))
        name = buf[off:off + name_len].decode("utf-8"); off += name_len
        nel = int(np.prod(ne))
        nbytes = nel // BLOCK_ELEMS[ttype] * BLOCK_BYTES[ttype]
        mf.tensors.append(TensorRec(name, ttype, ne, buf[off:off + nbytes])); off += nbytes
    return mf
