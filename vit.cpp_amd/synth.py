"""Seeded synthetic ViT weights and inputs in the reference's file format.

There is no network and no pretrained checkpoint in the build environment
(SURVEY.md 8c), so every parity and bench run uses random-init weights of the
named architecture written exactly like /root/reference/convert-pth-to-ggml.py
would write a timm checkpoint (tensor names, reversed dims, f16/f32 split).
Recipe per SURVEY.md 8d: rng(1234); matrices ~ trunc-N(0, 0.02^2); biases
N(0,0.02^2); LN weight 1+N(0,0.02^2); head.weight scaled so the class softmax
is peaked (a 1e-3 probability tolerance then means something).
"""
from __future__ import annotations

import os
from typing import Dict

import numpy as np

from .ggml_file import HParams, write_model

CONFIGS = {
    # name: (hidden, layers, heads, classes, patch, img)
    "vit_micro_patch16_64": (128, 2, 2, 10, 16, 64),       # test-only toy (N=17)
    "vit_micro_c37_patch16_64": (128, 2, 2, 37, 16, 64),   # test-only toy with an ODD class count (the head GEMM's last column group is ragged)
    "vit_micro_c21843_patch16_64": (128, 2, 2, 21843, 16, 64),   # test-only toy with ImageNet-21k's class count (timm *_in21k heads): 86 column tiles of the head GEMM, a 21843-wide class softmax
    "vit_micro_patch8_224": (128, 2, 2, 10, 8, 224),       # test-only toy with the token count of the reference's default hparams (N=785)
    "vit_micro_hd32_patch16_64": (128, 2, 4, 10, 16, 64),  # test-only toys with head dims other than 64 (the generic attention kernel): 32,
    "vit_micro_hd96_patch16_96": (192, 2, 2, 10, 16, 96),  #   96 (N = 37),
    "vit_mini_hd80_patch14_112": (1280, 2, 16, 10, 14, 112),   # 80 with ViT-H/14's widths (patch 14: N = 65)
    "vit_mini_hd72_patch14_112": (1152, 2, 16, 10, 14, 112),   # 72 with SO400M's widths (1152 is not a multiple of the 256-column GEMM tiles)
    "vit_base_patch8_224": (768, 12, 12, 1000, 8, 224),    # the reference's default hparams (vit.h:22-28)
    "vit_tiny_patch16_224": (192, 12, 3, 1000, 16, 224),
    "vit_small_patch16_224": (384, 12, 6, 1000, 16, 224),
    "vit_base_patch16_224": (768, 12, 12, 1000, 16, 224),
    "vit_large_patch16_224": (1024, 24, 16, 1000, 16, 224),
    "vit_large_patch16_384": (1024, 24, 16, 1000, 16, 384),
    # ViTSTR (extensions/vitstr.cpp): the same encoder on ONE grey channel; the head reads the first 25 tokens (vitstr.cpp:864-904)
    "vitstr_micro_patch16_64": (128, 2, 2, 96, 16, 64),    # test-only toy (N = 17 < 25 is NOT usable for the sequence head: see vitstr_small)
    "vitstr_tiny_patch16_224": (192, 12, 3, 96, 16, 224),
}
IN_CHANS = {"vitstr_micro_patch16_64": 1, "vitstr_tiny_patch16_224": 1}     # everything else: 3 (RGB)
VITSTR_SEQ_LEN = 25
# ViTSTR's character set (vitstr's TokenLabelConverter): [GO], [s], then the 94 printable ASCII characters
VITSTR_LABELS = {0: "[GO]", 1: "[s]", **{i + 2: chr(33 + i) for i in range(94)}}


def hparams_for(name: str, ftype: int = 1) -> HParams:
    return HParams(*CONFIGS[name], ftype=ftype)


def gflop_per_image(hp: HParams) -> float:
    """2 x MACs of every GEMM in the graph (elementwise excluded)."""
    D, L, h, C, P = hp.hidden_size, hp.num_hidden_layers, hp.num_attention_heads, hp.num_classes, hp.patch_size
    g = hp.img_size // P
    N = g * g + 1
    d = D // h
    per_layer = N * D * 3 * D + 2 * h * N * N * d + N * D * D + 2 * N * D * 4 * D
    return 2.0 * (g * g * 3 * P * P * D + L * per_layer + D * C) / 1e9


def make_weights(hp: HParams, seed: int = 1234, head_scale: float = 8.0, in_chans: int = 3) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    D, L, C, P = hp.hidden_size, hp.num_hidden_layers, hp.num_classes, hp.patch_size
    N = hp.n_tokens

    def mat(*shape):
        return np.clip(rng.standard_normal(shape, dtype=np.float32) * np.float32(0.02), -0.04, 0.04).astype(np.float32)

    def vec(n, mean=0.0):
        return (np.float32(mean) + rng.standard_normal(n, dtype=np.float32) * np.float32(0.02)).astype(np.float32)

    t: Dict[str, np.ndarray] = {}
    t["cls_token"] = mat(1, 1, D)
    t["pos_embed"] = mat(1, N, D)
    t["patch_embed.proj.weight"] = mat(D, in_chans, P, P)
    t["patch_embed.proj.bias"] = vec(D)
    for i in range(L):
        p = f"blocks.{i}."
        t[p + "norm1.weight"] = vec(D, 1.0); t[p + "norm1.bias"] = vec(D)
        t[p + "attn.qkv.weight"] = mat(3 * D, D); t[p + "attn.qkv.bias"] = vec(3 * D)
        t[p + "attn.proj.weight"] = mat(D, D); t[p + "attn.proj.bias"] = vec(D)
        t[p + "norm2.weight"] = vec(D, 1.0); t[p + "norm2.bias"] = vec(D)
        t[p + "mlp.fc1.weight"] = mat(4 * D, D); t[p + "mlp.fc1.bias"] = vec(4 * D)
        t[p + "mlp.fc2.weight"] = mat(D, 4 * D); t[p + "mlp.fc2.bias"] = vec(D)
    t["norm.weight"] = vec(D, 1.0); t["norm.bias"] = vec(D)
    t["head.weight"] = (mat(C, D) * np.float32(head_scale)).astype(np.float32)
    t["head.bias"] = vec(C)
    return t


def write_synthetic(path: str, name: str, ftype: int = 1, seed: int = 1234, head_scale: float = 8.0) -> HParams:
    hp = hparams_for(name, ftype)
    ic = IN_CHANS.get(name, 3)
    write_model(path, hp, make_weights(hp, seed, head_scale, in_chans=ic), ftype=ftype, id2label=dict(VITSTR_LABELS) if ic == 1 else None)
    return hp


def cached_synthetic(name: str, ftype: int = 1, seed: int = 1234, head_scale: float = 8.0, cache_dir: str | None = None) -> str:
    """Write (once) and return the path of a synthetic model file under a scratch dir."""
    cache_dir = cache_dir or os.environ.get("VITX_CACHE", "/tmp/vitx_cache")
    os.makedirs(cache_dir, exist_ok=True)
    path = os.path.join(cache_dir, f"{name}-s{seed}-h{head_scale:g}-ft{ftype}.gguf")
    if not os.path.exists(path):
        tmp = path + f".tmp{os.getpid()}"
        write_synthetic(tmp, name, ftype, seed, head_scale)
        os.replace(tmp, path)
    return path


IMAGENET_MEAN = np.array([123.675, 116.280, 103.530], np.float32)
IMAGENET_STD = np.array([58.395, 57.120, 57.375], np.float32)


def synthetic_images_u8(n: int, size: int, seed: int = 4321) -> np.ndarray:
    return np.random.default_rng(seed).integers(0, 256, size=(n, size, size, 3), dtype=np.uint8)


def normalize_u8(img_u8: np.ndarray) -> np.ndarray:
    """(v - mean)/std in f32, HWC -- what vit_image_preprocess emits (vit.cpp:280)."""
    return ((img_u8.astype(np.float32) - IMAGENET_MEAN) / IMAGENET_STD).astype(np.float32)
