"""Converter: a HuggingFace `transformers` ViTForImageClassification -> the reference's legacy-ggml ".gguf" file.

Counterpart of the reference's /root/reference/convert-pth-to-ggml.py (which needs `timm`): same output layout
(writer rules convert-pth-to-ggml.py:105-158 live in ggml_file.write_model), different source naming -- HF splits
the fused qkv projection into query/key/value (concatenated here in timm's q,k,v order, vit.cpp:826-834) and calls
the sub-modules `vit.encoder.layer.N.*` (transformers < 5) or `vit.layers.N.*` (>= 5).  Offline tool, not on the
compute path.  The model must use tanh-GELU or exact GELU weights trained for it: the forward path implements
ggml_gelu (tanh form, vit.cpp:889-893) only.

A timm checkpoint needs no `timm` either: its state_dict already carries the names the file format uses (the reference's converter
writes `timm_model.state_dict()` verbatim, convert-pth-to-ggml.py:121-133), so `--timm-state-dict model.pth` loads the tensors with
torch.load and derives the hyper-parameters the reference reads from the timm module (:96-103) from the tensor shapes.

    python -m ... convert.py <hf_model_dir_or_name> <out.gguf> [--ftype 1]
    python -m ... convert.py --timm-state-dict <checkpoint.pth> <out.gguf> [--ftype 1] [--heads H] [--labels labels.json]
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from .ggml_file import HParams, write_model


def state_dict_to_timm(sd: Dict[str, np.ndarray], num_layers: int) -> Dict[str, np.ndarray]:
    """Rename / fuse an HF ViTForImageClassification state dict (numpy arrays) into timm's names, in the tensor
    order the reference's loader expects to find them (vit.cpp:512-574)."""
    new = "vit.layers.0.attention.q_proj.weight" in sd
    out: Dict[str, np.ndarray] = {}
    out["cls_token"] = sd["vit.embeddings.cls_token"]
    out["pos_embed"] = sd["vit.embeddings.position_embeddings"]
    out["patch_embed.proj.weight"] = sd["vit.embeddings.patch_embeddings.projection.weight"]
    out["patch_embed.proj.bias"] = sd["vit.embeddings.patch_embeddings.projection.bias"]
    for i in range(num_layers):
        q = f"vit.layers.{i}." if new else f"vit.encoder.layer.{i}."
        p = f"blocks.{i}."
        qkv = ("attention.q_proj", "attention.k_proj", "attention.v_proj") if new else \
              ("attention.attention.query", "attention.attention.key", "attention.attention.value")
        o = "attention.o_proj" if new else "attention.output.dense"
        f1 = "mlp.fc1" if new else "intermediate.dense"
        f2 = "mlp.fc2" if new else "output.dense"
        out[p + "norm1.weight"] = sd[q + "layernorm_before.weight"]; out[p + "norm1.bias"] = sd[q + "layernorm_before.bias"]
        out[p + "attn.qkv.weight"] = np.concatenate([sd[q + n + ".weight"] for n in qkv], 0)
        out[p + "attn.qkv.bias"] = np.concatenate([sd[q + n + ".bias"] for n in qkv], 0)
        out[p + "attn.proj.weight"] = sd[q + o + ".weight"]; out[p + "attn.proj.bias"] = sd[q + o + ".bias"]
        out[p + "norm2.weight"] = sd[q + "layernorm_after.weight"]; out[p + "norm2.bias"] = sd[q + "layernorm_after.bias"]
        out[p + "mlp.fc1.weight"] = sd[q + f1 + ".weight"]; out[p + "mlp.fc1.bias"] = sd[q + f1 + ".bias"]
        out[p + "mlp.fc2.weight"] = sd[q + f2 + ".weight"]; out[p + "mlp.fc2.bias"] = sd[q + f2 + ".bias"]
    out["norm.weight"] = sd["vit.layernorm.weight"]; out["norm.bias"] = sd["vit.layernorm.bias"]
    out["head.weight"] = sd["classifier.weight"]; out["head.bias"] = sd["classifier.bias"]
    return {k: np.ascontiguousarray(np.asarray(v, np.float32)) for k, v in out.items()}


def convert_hf_model(model, path: str, ftype: int = 1, vitstr: bool = False) -> HParams:
    """model: transformers.ViTForImageClassification (eval).  Writes `path`; returns the hparams written.
    vitstr=True: the model is a ViTSTR scene-text recogniser (/root/reference/extensions/vitstr.cpp/convert-pth-to-ggml.py: a ViT with ONE
    input channel whose classifier is applied to the first 25 tokens); the file then carries the character set as labels, and the
    one-channel patch kernel is what makes vit_model_load / vitx_model_load treat it as a ViTSTR model (vitstr.cpp:482)."""
    cfg = model.config
    if vitstr and getattr(cfg, "num_channels", 3) != 1:
        raise ValueError("a ViTSTR model takes one (grey) input channel")
    hd = cfg.hidden_size // cfg.num_attention_heads
    if cfg.hidden_size % cfg.num_attention_heads or hd % 8 or not 8 <= hd <= 128:
        raise ValueError(f"head_dim {hd}: the forward path takes multiples of 8 up to 128 (64 runs the tuned attention kernels)")
    hp = HParams(cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.num_labels, cfg.patch_size, cfg.image_size, ftype)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    tensors = state_dict_to_timm(sd, cfg.num_hidden_layers)
    id2label = {int(k): str(v) for k, v in (getattr(cfg, "id2label", None) or {}).items()} or None
    if vitstr:
        from .synth import VITSTR_LABELS
        if cfg.num_labels != len(VITSTR_LABELS):
            raise ValueError(f"ViTSTR's character set has {len(VITSTR_LABELS)} classes ([GO], [s], 94 printable characters), the model has {cfg.num_labels}")
        id2label = dict(VITSTR_LABELS)
    write_model(path, hp, tensors, id2label=id2label, ftype=ftype)
    return hp


_TIMM_UNSUPPORTED = ("fc_norm.", "reg_token", "dist_token", "head_dist.", ".ls1.", ".ls2.", ".q_norm.", ".k_norm.", "attn_pool.")


def convert_timm_state_dict(sd, path: str, ftype: int = 1, heads: int = 0, id2label=None) -> HParams:
    """sd: a timm VisionTransformer state_dict (name -> array / tensor), e.g. torch.load("vit_base_patch16_224.pth").  Mirrors
    /root/reference/convert-pth-to-ggml.py:96-158 without importing timm: hidden size, depth, classes, patch and image size come from the
    tensor shapes (the reference reads them off the timm module), `norm_pre.*` is skipped exactly as there (:117-120), the ViTSTR
    extension's checkpoints lose their "module.vitstr." prefix (extensions/vitstr.cpp/convert-pth-to-ggml.py:226-229) and are recognised
    by their one-channel patch kernel.  Models with tensors the reference's loader has no slot for (fc_norm, layer-scale, register or
    distillation tokens, qk-norm: vit.cpp:618-622 would reject the file) are refused here, by name."""
    sd = {k: v for k, v in (sd.get("model", sd) if isinstance(sd, dict) and "model" in sd and not hasattr(sd["model"], "shape") else sd).items()}
    t: Dict[str, np.ndarray] = {}
    for k, v in sd.items():
        k = k.replace("module.vitstr.", "")
        if k.startswith("norm_pre"):
            continue
        if any(u in k for u in _TIMM_UNSUPPORTED):
            raise ValueError(f"tensor {k!r}: this timm variant has components the reference's file format has no slot for")
        t[k] = np.ascontiguousarray(np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v, np.float32))
    for need in ("cls_token", "pos_embed", "patch_embed.proj.weight", "patch_embed.proj.bias", "norm.weight", "norm.bias", "head.weight", "head.bias"):
        if need not in t:
            raise ValueError(f"not a timm VisionTransformer state_dict: {need!r} is missing")
    D = int(t["cls_token"].shape[-1])
    L = 1 + max(int(k.split(".")[1]) for k in t if k.startswith("blocks."))
    Dw, cin, P, P2 = t["patch_embed.proj.weight"].shape
    n_tok = int(t["pos_embed"].shape[1])
    g = int(round((n_tok - 1) ** 0.5))
    if Dw != D or P != P2 or g * g + 1 != n_tok or cin not in (1, 3):
        raise ValueError(f"unexpected shapes: patch kernel {t['patch_embed.proj.weight'].shape}, pos_embed {t['pos_embed'].shape}")
    # The head count is NOT in a state_dict (the reference reads timm's module attribute, convert-pth-to-ggml.py): it is only inferred for
    # the widths of timm's released ViTs, where it is unambiguous; any other width needs --heads (r03 advisor: D // 64 silently turned
    # ViT-H/14's 16 heads of 80 into 20 heads of 64 -- a file that loads, runs and is wrong).
    _TIMM_HEADS = {192: 3, 384: 6, 768: 12, 1024: 16, 1280: 16, 1152: 16, 1408: 16, 1664: 16}      # tiny, small, base, large, huge, so400m, giant, gigantic
    H = heads or _TIMM_HEADS.get(D, 0)
    if H <= 0:
        raise ValueError(f"hidden size {D} is not a released timm ViT width ({sorted(_TIMM_HEADS)}): pass the head count (--heads)")
    if D % H or (D // H) % 8 or not 8 <= D // H <= 128:
        raise ValueError(f"head_dim {D // H if H and D % H == 0 else '?'}: the forward path takes multiples of 8 up to 128 (pass --heads for a model whose head_dim is not 64)")
    hp = HParams(D, L, H, int(t["head.weight"].shape[0]), int(P), g * int(P), ftype)
    if cin == 1 and id2label is None:
        from .synth import VITSTR_LABELS
        if hp.num_classes == len(VITSTR_LABELS):
            id2label = dict(VITSTR_LABELS)
    expected = 4 + 12 * L + 4
    if len(t) != expected:
        raise ValueError(f"{len(t)} tensors after filtering, the file format holds exactly {expected} for {L} layers (vit.cpp:512-574)")
    write_model(path, hp, t, id2label=id2label, ftype=ftype)
    return hp


def main(argv=None) -> int:
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("model"); ap.add_argument("out"); ap.add_argument("--ftype", type=int, default=1, help="0 f32, 1 f16 (default), 2/3/6/7/8 q4_0/q4_1/q5_0/q5_1/q8_0")
    ap.add_argument("--vitstr", action="store_true", help="one-channel ViTSTR scene-text model: write the character set as labels")
    ap.add_argument("--timm-state-dict", action="store_true", help="`model` is a torch-saved timm VisionTransformer state_dict (.pth); no timm import needed")
    ap.add_argument("--heads", type=int, default=0, help="attention heads of a timm checkpoint (inferred only for the widths of released timm ViTs; required otherwise)")
    ap.add_argument("--labels", default=None, help="JSON file {class id: label} for a timm checkpoint (default: none are written)")
    a = ap.parse_args(argv)
    if a.timm_state_dict:
        import json
        import torch
        sd = torch.load(a.model, map_location="cpu", weights_only=True)
        labels = {int(k): str(v) for k, v in json.load(open(a.labels)).items()} if a.labels else None
        hp = convert_timm_state_dict(sd, a.out, a.ftype, heads=a.heads, id2label=labels)
        print(f"wrote {a.out}: hidden {hp.hidden_size}, layers {hp.num_hidden_layers}, heads {hp.num_attention_heads}, classes {hp.num_classes}, patch {hp.patch_size}, img {hp.img_size}, ftype {a.ftype}")
        return 0
    import transformers
    m = transformers.ViTForImageClassification.from_pretrained(a.model).eval()
    hp = convert_hf_model(m, a.out, a.ftype, vitstr=a.vitstr)
    print(f"wrote {a.out}: hidden {hp.hidden_size}, layers {hp.num_hidden_layers}, heads {hp.num_attention_heads}, classes {hp.num_classes}, patch {hp.patch_size}, img {hp.img_size}, ftype {a.ftype}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
