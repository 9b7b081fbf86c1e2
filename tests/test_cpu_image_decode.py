"""load_image_from_file (vit.cpp:109-127 = stbi_load(..., 3)) replaced by libvitx.so's own decoder (csrc/image_decode.cpp).

stb_image.h is not in the tree (it lives in the absent ggml submodule), so the decoder is UNPINNED against stb; it is checked
against an independent decoder (PIL / libjpeg-turbo): JPEG within the band two correct decoders differ by (different IDCT,
chroma-upsampling and colour-conversion roundings), PNG and PPM bit for bit.
"""
import io
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ASSET_DIR = os.path.join(HERE, "golden", "assets")
ASSETS = sorted(os.listdir(ASSET_DIR))


def _pil(path_or_bytes):
    from PIL import Image
    f = io.BytesIO(path_or_bytes) if isinstance(path_or_bytes, (bytes, bytearray)) else path_or_bytes
    return np.asarray(Image.open(f).convert("RGB"), dtype=np.uint8)


def _band(got, ref, max_abs, frac_gt1, mean_abs=0.15):
    assert got.shape == ref.shape
    d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= max_abs, d.max()
    assert (d > 1).mean() <= frac_gt1, (d > 1).mean()
    assert d.mean() <= mean_abs, d.mean()


def test_bundled_images_decode_like_an_independent_decoder(binding):
    """The reference's 10 bundled images (3 of them progressive JPEGs, image.png is really a JPEG): same geometry as PIL,
    every value within 3 of PIL's, at most 2.5 % of the values further than 1 away."""
    from PIL import Image
    n_prog = 0
    for a in ASSETS:
        path = os.path.join(ASSET_DIR, a)
        n_prog += int(bool(Image.open(path).info.get("progressive", 0)))
        _band(binding.load_image(path), _pil(path), 3, 0.025)
    assert n_prog == 3


@pytest.mark.parametrize("kw", [dict(subsampling=0), dict(subsampling=1), dict(subsampling=2), dict(subsampling=2, progressive=True),
                                dict(subsampling=0, progressive=True, quality=95), dict(subsampling=2, quality=30, optimize=True),
                                dict(subsampling=2, restart_marker_blocks=3), dict(subsampling=1, progressive=True, restart_marker_rows=1)])
@pytest.mark.parametrize("shape", [(64, 64), (37, 53), (1, 1), (8, 200), (129, 17)])
def test_jpeg_variants(binding, kw, shape):
    """4:4:4 / 4:2:2 / 4:2:0, baseline and progressive, optimised Huffman tables, restart intervals, sizes that are not a
    multiple of the MCU, and grayscale."""
    from PIL import Image
    rng = np.random.default_rng(shape[0] * 131 + shape[1])
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
    img = np.stack([(xx * 5 + yy * 3) % 256, (xx * 2 + 40 * np.sin(yy / 5.0) + 128) % 256, rng.integers(0, 256, shape)], -1).astype(np.uint8)
    for mode in ("RGB", "L"):
        buf = io.BytesIO()
        try:
            Image.fromarray(img if mode == "RGB" else img[..., 0], mode).save(buf, "JPEG", **({k: v for k, v in kw.items() if not (mode == "L" and k == "subsampling")}))
        except TypeError:
            pytest.skip("this PIL cannot write these JPEG options")
        blob = buf.getvalue()
        got, ref = binding.decode_image(blob), _pil(blob)
        if kw.get("subsampling") == 1 and shape[1] > 2:
            # 4:2:2 only: the decoder keeps stb_image's right-edge rule of its horizontal 2x resampler (the second-to-last output
            # column weights the last two chroma samples 3:1 the "wrong" way round, stbi__resample_row_h_2); libjpeg does not, and the
            # random-noise chroma of this test image makes that one column differ visibly.  Checked separately below.
            col = 2 * ((shape[1] + 1) // 2) - 2
            keep = [x for x in range(shape[1]) if x != col]
            got, ref = got[:, keep], ref[:, keep]
        _band(got, ref, 4, 0.08, 0.4)         # noise-filled chroma plane: rounding differences of the two upsamplers show everywhere


def test_png_and_ppm_are_bit_exact(binding, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (45, 67, 3), dtype=np.uint8)
    rgba = np.concatenate([rgb, rng.integers(0, 256, (45, 67, 1), dtype=np.uint8)], -1)
    cases = {"rgb": Image.fromarray(rgb, "RGB"), "rgba": Image.fromarray(rgba, "RGBA"), "gray": Image.fromarray(rgb[..., 0], "L"),
             "gray_alpha": Image.fromarray(rgba[..., [0, 3]], "LA"), "palette": Image.fromarray(rgb, "RGB").quantize(colors=17),
             "bilevel": Image.fromarray((rgb[..., 0] > 127).astype(np.uint8) * 255, "L").convert("1"), "big_gradient": Image.fromarray(np.tile(np.arange(256, dtype=np.uint8), (300, 4)), "L")}
    for name, im in cases.items():
        for level in (0, 6, 9):                               # stored, dynamic-Huffman and best-compression deflate streams
            buf = io.BytesIO(); im.save(buf, "PNG", compress_level=level)
            got = binding.decode_image(buf.getvalue())
            want = np.asarray(im.convert("RGB"), dtype=np.uint8)
            assert np.array_equal(got, want), (name, level)
    p = tmp_path / "x.ppm"
    with open(p, "wb") as f:
        f.write(b"P6\n# a comment\n67 45\n255\n" + rgb.tobytes())
    assert np.array_equal(binding.load_image(str(p)), rgb)


def test_undecodable_input_is_an_error_not_a_crash(binding, tmp_path):
    blob = open(os.path.join(ASSET_DIR, "tench.jpg"), "rb").read()
    with pytest.raises(binding.VitxError):
        binding.load_image(str(tmp_path / "missing.jpg"))
    for bad in (b"", b"hello world, not an image", blob[:2], b"\x89PNG\r\n\x1a\n" + b"\0" * 20):
        with pytest.raises(binding.VitxError):
            binding.decode_image(bad)
    rng = np.random.default_rng(0)
    for cut in (100, 1000, len(blob) // 2, len(blob) - 2):      # truncated files: an error or a (partially grey) image, never a crash
        try:
            img = binding.decode_image(blob[:cut])
            assert img.shape == (408, 612, 3)
        except binding.VitxError:
            pass
    for _ in range(40):                                         # bit flips inside the entropy-coded data
        b = bytearray(blob); i = int(rng.integers(700, len(b) - 2)); b[i] ^= 1 << int(rng.integers(0, 8))
        try:
            binding.decode_image(bytes(b))
        except binding.VitxError:
            pass


def test_ppm_header_without_raster_is_rejected(binding):
    """r02 advisor (high): a P6 file that ends right after maxval made `pos` step past the end, the unsigned `n - pos` wrapped and the
    decoder copied w*h*3 bytes from beyond the buffer.  Every truncation point of a small valid file must be an error or a full image."""
    rgb = np.arange(64 * 64 * 3, dtype=np.uint8)
    for bad in (b"P6 64 64 255", b"P6\n64 64\n255", b"P6 64 64 255\n", b"P6 64 64 255\n" + bytes(100), b"P6 64 64 255X" + rgb.tobytes(), b"P6 64 64", b"P6"):
        with pytest.raises(binding.VitxError):
            binding.decode_image(bad)
    good = b"P6 64 64 255\n" + rgb.tobytes()
    assert binding.decode_image(good).shape == (64, 64, 3)
    for cut in range(2, 40):
        with pytest.raises(binding.VitxError):
            binding.decode_image(good[:cut])


def _adam7_png(img, depth=8, ctype=2):
    """An Adam7-interlaced PNG written by hand (PIL cannot write one): filter type 0 on every row of every pass."""
    import struct, zlib
    h, w = img.shape[:2]
    xo, yo, xs, ys = (0, 4, 0, 2, 0, 1, 0), (0, 0, 4, 0, 2, 0, 1), (8, 8, 4, 4, 2, 2, 1), (8, 8, 8, 4, 4, 2, 2)
    raw = b""
    for p in range(7):
        sub = img[yo[p]::ys[p], xo[p]::xs[p]]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        for row in sub:
            raw += b"\0" + (np.packbits(row).tobytes() if depth == 1 else row.tobytes())
    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data))
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")


@pytest.mark.parametrize("shape", [(45, 67), (1, 1), (3, 2), (8, 8), (9, 17), (5, 1)])
def test_png_adam7_interlaced(binding, shape):
    """stbi_load de-interlaces Adam7 PNGs; so does the replacement (r02 advisor).  RGB, grey and 1-bit images of sizes where some passes are empty."""
    rng = np.random.default_rng(shape[0] * 100 + shape[1])
    rgb = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    assert np.array_equal(binding.decode_image(_adam7_png(rgb, 8, 2)), rgb)
    grey = rgb[..., 0]
    assert np.array_equal(binding.decode_image(_adam7_png(grey, 8, 0)), np.repeat(grey[..., None], 3, -1))
    bits = (grey > 127).astype(np.uint8)
    assert np.array_equal(binding.decode_image(_adam7_png(bits, 1, 0)), np.repeat((bits * 255)[..., None], 3, -1))
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(io.BytesIO(_adam7_png(rgb, 8, 2))).convert("RGB")), rgb)       # the hand-written encoder is a valid PNG


def test_png_zlib_stream_cannot_expand_beyond_the_image(binding):
    """A small IDAT that inflates to far more than (stride + 1) * height bytes is refused while inflating (r02 advisor: no output cap)."""
    import struct, zlib
    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data))
    bomb = zlib.compress(bytes(64 << 20), 9)                   # 64 MiB of zeros in ~64 KiB
    blob = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 4, 4, 8, 2, 0, 0, 0)) + chunk(b"IDAT", bomb) + chunk(b"IEND", b"")
    with pytest.raises(binding.VitxError, match="expands beyond"):
        binding.decode_image(blob)


def test_jpeg_adobe_colour_spaces(binding):
    """stbi_load(..., 3) converts Adobe CMYK / YCCK files (4 components, APP14 transform 0 / 2) to RGB with its x * k / 255 products and
    takes 3-component files marked `transform 0` as plain RGB.  PIL writes CMYK JPEGs as inverted Adobe CMYK; the expected RGB is stb's
    formula applied to PIL's own (lossy-decoded, inverted) planes."""
    from PIL import Image
    rng = np.random.default_rng(5)
    base = np.kron(rng.integers(0, 256, (6, 8, 4), dtype=np.uint8), np.ones((8, 8, 1), dtype=np.uint8))     # blocky: the DCT keeps it nearly exact
    im = Image.fromarray(base, "CMYK")
    buf = io.BytesIO(); im.save(buf, "JPEG", quality=100, subsampling=0)
    got = binding.decode_image(buf.getvalue())
    dec = np.asarray(Image.open(io.BytesIO(buf.getvalue())), dtype=np.int32)      # PIL hands back the CMYK planes (already un-inverted)
    inv = 255 - dec                                                                # the bytes in the file (Adobe stores them inverted)
    t = inv[..., :3] * inv[..., 3:4] + 128
    want = (t + (t >> 8)) >> 8
    assert got.shape == want.shape and np.abs(got.astype(np.int32) - want).max() <= 2
    # a 3-component file with Adobe transform 0 and no JFIF marker is RGB: patch PIL's YCbCr output into that form
    rgb = np.kron(rng.integers(0, 256, (4, 4, 3), dtype=np.uint8), np.ones((8, 8, 1), dtype=np.uint8))
    buf = io.BytesIO(); Image.fromarray(rgb, "RGB").save(buf, "JPEG", quality=100, subsampling=0)
    blob = buf.getvalue()
    ycc = binding.decode_image(blob)
    assert blob[2:4] == b"\xff\xe0"                                              # JFIF APP0 right after SOI
    app0_len = int.from_bytes(blob[4:6], "big")
    app14 = b"\xff\xee" + (14).to_bytes(2, "big") + b"Adobe" + bytes([0, 100, 0, 0, 0, 0, 0])   # transform 0
    as_rgb = binding.decode_image(blob[:2] + app14 + blob[4 + app0_len:])
    planes = np.asarray(Image.open(io.BytesIO(blob)).convert("YCbCr"))             # PIL: RGB -> YCbCr of the decoded image ~ the stored planes
    assert not np.array_equal(as_rgb, ycc)
    assert np.abs(as_rgb.astype(np.int32) - planes.astype(np.int32)).max() <= 3


def test_reference_main_flow_builds_on_the_mirror_header(tmp_path):
    """examples/vit_main.cpp is /root/reference/main.cpp minus its ggml lines (:82-91, :110): it calls load_image_from_file,
    vit_image_preprocess, vit_model_load and vit_predict through vit.cpp_amd/vit.h and must link with plain g++."""
    root = os.path.dirname(HERE)
    pkgdir = os.path.join(root, "vit.cpp_amd")
    exe = str(tmp_path / "vit_main")
    cmd = ["g++", "-std=c++17", "-O1", os.path.join(root, "examples", "vit_main.cpp"), "-I" + pkgdir, "-L" + pkgdir, "-lvitx", "-L/opt/rocm/lib",
           "-Wl,-rpath," + pkgdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    r = subprocess.run([exe, "-m", str(tmp_path / "none.gguf"), "-i", os.path.join(ASSET_DIR, "tench.jpg")], capture_output=True, text=True)
    assert r.returncode == 1 and "failed to load model" in r.stderr
