"""Command-line front end equivalent to the reference's `vit` binary (/root/reference/main.cpp:25-112) and its
directory-walk accuracy harness:

    python vit_cli.py -m model.gguf -i image.jpg [-k 5] [--dtype f16|bf16] [--interp bicubic|bilinear]
    python vit_cli.py -m model.gguf --dir imagenet_val/ [--batch 256]       # top-1 over <dir>/<label>/*.jpg

Same flags as vit_params_parse (vit.cpp:955-1002: -m -i -t -k -s -e; -t, -s and -e are accepted and ignored exactly
as the reference's forward ignores seed and eps), same stdout lines (" > label : 0.xx", vit.cpp:1062-1067) and the
same stderr timing block.  Decoding is PIL here (the reference uses stb_image inside the absent ggml tree); everything
after the decoded u8 RGB array -- preprocess, forward, top-k -- runs through the C ABI of libvitx.so on the GPU.
"""
from __future__ import annotations

import argparse
import os
import sys
import time
from typing import List

import numpy as np


def _decode(path: str) -> np.ndarray:
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)


def main(argv: List[str] | None = None) -> int:
    from . import binding
    ap = argparse.ArgumentParser(prog="vit", description="ViT inference on MI355X (drop-in for staghado/vit.cpp's CLI)")
    ap.add_argument("-m", "--model", default="../ggml-model-f16.gguf")
    ap.add_argument("-i", "--inp", default="../assets/tench.jpg")
    ap.add_argument("-t", "--threads", type=int, default=4, help="accepted for compatibility; the GPU path has no thread count")
    ap.add_argument("-k", "--topk", type=int, default=5)
    ap.add_argument("-s", "--seed", type=int, default=-1, help="accepted for compatibility (unused by the forward, as in the reference)")
    ap.add_argument("-e", "--epsilon", type=float, default=1e-6, help="accepted for compatibility (the reference's forward uses hparams.eps = 1e-6)")
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"], help="MFMA operand type; f16 reproduces the reference's rounding points")
    ap.add_argument("--interp", default="bicubic", choices=["bicubic", "bilinear"])
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--dir", default=None, help="accuracy harness: walk DIR/<label>/* and report top-1 against the directory name")
    ap.add_argument("--batch", type=int, default=256, help="images per forward in --dir mode")
    a = ap.parse_args(argv)

    t_main = time.perf_counter()
    print(f"main: seed = {a.seed if a.seed >= 0 else int(time.time())}", file=sys.stderr)
    try:
        model = binding.Model(a.model)
    except binding.VitxError as e:
        print(f"main: failed to load model from '{a.model}': {e}", file=sys.stderr)
        return 1
    t_load = time.perf_counter() - t_main
    dt = binding.F16 if a.dtype == "f16" else binding.BF16
    interp = binding.BICUBIC if a.interp == "bicubic" else binding.BILINEAR
    S = model.img_size

    if a.dir is None:
        try:
            img0 = _decode(a.inp)
        except Exception as e:                                              # main.cpp:69-73
            print(f"main: failed to load image from '{a.inp}': {e}", file=sys.stderr)
            return 1
        print(f"main: loaded image '{a.inp}' ({img0.shape[1]} x {img0.shape[0]})", file=sys.stderr)
        img1 = binding.preprocess(img0, S, interp)
        print(f"processed, out dims : ({S} x {S})", file=sys.stderr)
        ctx = binding.Context(model, device=a.device, max_batch=1, dtype=dt)
        probs = ctx.forward(img1[None])[0]
        idx, val = binding.topk(probs, a.topk)
        print("", file=sys.stderr)
        for i, p in zip(idx, val):                                          # vit.cpp:1062-1067
            print(f" > {model.label(i)} : {p:.2f}")
        t_all = time.perf_counter() - t_main
        print("\n", file=sys.stderr)
        print(f"main:    model load time = {t_load * 1e3:8.2f} ms", file=sys.stderr)
        print(f"main:    processing time = {(t_all - t_load) * 1e3:8.2f} ms", file=sys.stderr)
        print(f"main:    total time      = {t_all * 1e3:8.2f} ms", file=sys.stderr)
        return 0

    # ---- accuracy harness: DIR/<label>/<image>; the label must be one of the model's id2label strings
    label_id = {model.label(i): i for i in range(model.num_classes) if model.label(i) is not None}
    files, truth = [], []
    for d in sorted(os.listdir(a.dir)):
        sub = os.path.join(a.dir, d)
        if not os.path.isdir(sub) or d not in label_id:
            continue
        for f in sorted(os.listdir(sub)):
            files.append(os.path.join(sub, f)); truth.append(label_id[d])
    if not files:
        print(f"main: no <label>/<image> files under '{a.dir}' match the model's labels", file=sys.stderr)
        return 1
    ctx = binding.Context(model, device=a.device, max_batch=min(a.batch, len(files)), dtype=dt)
    correct = 0
    t0 = time.perf_counter()
    for lo in range(0, len(files), ctx.max_batch):
        chunk = files[lo:lo + ctx.max_batch]
        batch = np.stack([binding.preprocess(_decode(f), S, interp) for f in chunk])
        pred = ctx.forward(batch).argmax(1)
        correct += int((pred == np.asarray(truth[lo:lo + len(chunk)])).sum())
    el = time.perf_counter() - t0
    print(f"top-1 accuracy: {correct / len(files):.4f} ({correct}/{len(files)})  {len(files) / el:.1f} images/s incl. decode + preprocess")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
