"""Generates the committed fixtures under tests/golden/.  Run from the repo root:

    python tests/golden/make_golden.py

What it pins (the reference itself has NO golden vectors and cannot be built here --
its ggml submodule is empty, SURVEY.md 8c -- so these are outputs of the oracle,
oracle/vit_oracle.c, frozen so that later edits to the oracle or to numpy's RNG stream
are caught):
  * assets/*                 the reference's 10 bundled images (/root/reference/assets),
                             copied verbatim; decoded with PIL in the tests (the reference
                             decodes with stb_image; parity is defined from the decoded u8 array on)
  * assets_decoded_sha1.json sha1 + shape of every PIL-decoded RGB array
  * preprocess_bicubic.npz   oracle bicubic resize+normalise of every asset to 224x224 (as rounded u8)
  * tiny_assets_{logits,probs}.npy  oracle (REF = ggml semantics) forward of synthetic
                             vit_tiny_patch16_224 (seed 1234, head x4) on the 10 preprocessed assets
  * tiny_synth_*.npy         same model on 4 synthetic noise images (seed 4321)
  * micro_synth_*.npy        vit_micro_patch16_64 on 5 noise images
  * base_synth_*.npy         vit_base_patch16_224 on 2 noise images (the benchmarked architecture)
  * weights_sha1.json        sha1 of the synthetic weight files (pins the numpy RNG stream + the writer)
"""
import hashlib
import json
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

pkg = _pkg.load()
from oracle import oracle as O  # noqa: E402

ASSETS = sorted(os.listdir(os.path.join(HERE, "assets")))


def decode(name):
    return np.asarray(Image.open(os.path.join(HERE, "assets", name)).convert("RGB"), dtype=np.uint8)


def sha1(b):
    return hashlib.sha1(b).hexdigest()


def main():
    dec = {}
    pre = {}
    for a in ASSETS:
        img = decode(a)
        dec[a] = {"sha1": sha1(img.tobytes()), "shape": list(img.shape)}
        f = O.preprocess(img, 224, "bicubic")
        # store the rounded u8 (exact inverse of the normalisation) to keep the fixture small
        u8 = np.rint(f * pkg.synth.IMAGENET_STD + pkg.synth.IMAGENET_MEAN).astype(np.uint8)
        assert np.array_equal(pkg.synth.normalize_u8(u8), f)
        pre[a] = u8
    json.dump(dec, open(os.path.join(HERE, "assets_decoded_sha1.json"), "w"), indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "preprocess_bicubic.npz"), **pre)

    wsha = {}
    tiny = pkg.synth.cached_synthetic("vit_tiny_patch16_224", head_scale=4.0)
    wsha["vit_tiny_patch16_224-h4"] = sha1(open(tiny, "rb").read())
    om = O.OracleModel(tiny)
    batch = np.stack([pkg.synth.normalize_u8(pre[a]) for a in ASSETS])
    lg, pr = om.forward(batch, O.REF)
    np.save(os.path.join(HERE, "tiny_assets_logits.npy"), lg); np.save(os.path.join(HERE, "tiny_assets_probs.npy"), pr)
    syn = pkg.synth.normalize_u8(pkg.synth.synthetic_images_u8(4, 224))
    lg, pr = om.forward(syn, O.REF)
    np.save(os.path.join(HERE, "tiny_synth_logits.npy"), lg); np.save(os.path.join(HERE, "tiny_synth_probs.npy"), pr)

    micro = pkg.synth.cached_synthetic("vit_micro_patch16_64", head_scale=4.0)
    wsha["vit_micro_patch16_64-h4"] = sha1(open(micro, "rb").read())
    lg, pr = O.OracleModel(micro).forward(pkg.synth.normalize_u8(pkg.synth.synthetic_images_u8(5, 64)), O.REF)
    np.save(os.path.join(HERE, "micro_synth_logits.npy"), lg); np.save(os.path.join(HERE, "micro_synth_probs.npy"), pr)

    base = pkg.synth.cached_synthetic("vit_base_patch16_224", head_scale=4.0)
    wsha["vit_base_patch16_224-h4"] = sha1(open(base, "rb").read())
    lg, pr = O.OracleModel(base).forward(pkg.synth.normalize_u8(pkg.synth.synthetic_images_u8(2, 224)), O.REF)
    np.save(os.path.join(HERE, "base_synth_logits.npy"), lg); np.save(os.path.join(HERE, "base_synth_probs.npy"), pr)
    json.dump(wsha, open(os.path.join(HERE, "weights_sha1.json"), "w"), indent=1, sort_keys=True)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
