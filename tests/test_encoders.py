"""CPU: the native output writers of stage 5 (main.py:399-404; csrc/sr_encode.cpp) against Pillow, the library the
reference calls.  TIFF-LZW and PNG must decode (with Pillow's libtiff / zlib) to the input bytes; the JPEG must decode
to exactly what Pillow's own quality-95 file of the same pixels decodes to (same libjpeg coefficients: Annex K tables,
islow DCT, 4:2:0) -- pinned against the reference's third-party writer itself, which IS importable here."""
import os

import numpy as np
import pytest
from PIL import Image

import _native


def _scene(rng, h, w, cn=3, noise=12):
    yy, xx = np.mgrid[0:h, 0:w]
    base = 128 + 64 * np.sin(xx / 37.0) + 48 * np.cos(yy / 23.0)
    if cn:
        base = base[..., None] + np.arange(cn) * 11
    return np.clip(base + rng.integers(-noise, noise + 1, base.shape), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("shape", [(517, 771, 3), (33, 17, 3), (1, 1, 3), (64, 2049, 0), (130, 97, 4), (16, 16, 3), (15, 1000, 3)])
@pytest.mark.parametrize("ext", ["tif", "png"])
def test_lossless_writers_roundtrip(rng, tmp_path, ext, shape):
    h, w, cn = shape
    img = _scene(rng, h, w, cn)
    p = str(tmp_path / f"a.{ext}")
    assert _native.write_image(img, p, threads=3) == ("TIFF" if ext == "tif" else "PNG")
    with Image.open(p) as im:
        back = np.asarray(im)
        assert im.mode == {0: "L", 3: "RGB", 4: "RGBA"}[cn]
        if ext == "tif":
            assert im.info.get("compression") == "tiff_lzw"
    assert back.shape == img.shape and np.array_equal(back, img)


def test_lzw_table_reset_and_flat_data(rng, tmp_path):
    """Noise fills the 4094-entry LZW table many times (clear codes, 12-bit codes); constant data gives the longest
    strings; both must survive, single- and multi-threaded output identical."""
    noise = rng.integers(0, 256, (300, 400, 3), dtype=np.uint8)
    flat = np.full((300, 400, 3), 7, np.uint8)
    for name, img in (("noise", noise), ("flat", flat)):
        p1, p2 = str(tmp_path / f"{name}1.tiff"), str(tmp_path / f"{name}2.tiff")
        _native.write_image(img, p1, threads=1)
        _native.write_image(img, p2, threads=8)
        assert open(p1, "rb").read() == open(p2, "rb").read()
        assert np.array_equal(np.asarray(Image.open(p1)), img)
    assert os.path.getsize(str(tmp_path / "flat1.tiff")) < 8000


def test_png_level_and_threads(rng, tmp_path):
    img = _scene(rng, 700, 900, 3)
    sizes = {}
    for level in (0, 3, 9):
        p = str(tmp_path / f"l{level}.png")
        _native.write_image(img, p, png_level=level, threads=4)
        assert np.array_equal(np.asarray(Image.open(p)), img)
        sizes[level] = os.path.getsize(p)
    assert sizes[0] > sizes[3] and sizes[0] > sizes[9]          # noisy data: level 9 is not always smaller than 3
    ref = str(tmp_path / "pil.png")
    Image.fromarray(img).save(ref, format="PNG", compress_level=3)
    assert sizes[3] < 1.6 * os.path.getsize(ref)            # filter None + chunked deflate costs some ratio, not 2x


@pytest.mark.parametrize("shape", [(517, 771, 3), (16, 16, 3), (17, 33, 3), (1, 1, 3), (240, 8, 3), (100, 130, 0), (9, 7, 0)])
def test_jpeg_matches_pillow_decode(rng, tmp_path, shape):
    h, w, cn = shape
    img = _scene(rng, h, w, cn)
    mine, ref = str(tmp_path / "m.jpg"), str(tmp_path / "p.jpg")
    assert _native.write_image(img, mine, threads=3) == "JPEG"
    Image.fromarray(img).save(ref, quality=95)                       # the reference's call (main.py:403)
    with Image.open(mine) as a, Image.open(ref) as b:
        assert a.mode == b.mode and a.size == b.size
        assert a.quantization == b.quantization                      # same scaled Annex K tables
        assert np.array_equal(np.asarray(a), np.asarray(b))          # same coefficients -> same decoded pixels
    for q in (50, 100):
        Image.fromarray(img).save(ref, quality=q)
        _native.write_image(img, mine, jpeg_quality=q)
        assert np.array_equal(np.asarray(Image.open(mine)), np.asarray(Image.open(ref)))


def test_writer_errors(tmp_path):
    img = np.zeros((4, 4, 3), np.uint8)
    with pytest.raises(ValueError):
        _native.write_image(img.astype(np.float32), str(tmp_path / "x.png"))
    with pytest.raises(OSError):                                      # Pillow: "cannot write mode RGBA as JPEG"
        _native.write_image(np.zeros((4, 4, 4), np.uint8), str(tmp_path / "x.jpg"))
    with pytest.raises(ValueError):
        _native.write_image(img, str(tmp_path / "no" / "such" / "dir" / "x.png"))


def test_other_extensions_go_to_pillow(tmp_path):
    """main.py:403's else-branch is ``img.save(path, quality=95)``: Pillow picks the format from the extension.  The native
    JPEG writer must only take the JPEG extensions -- a .bmp / .webp path gets that format, an unknown extension raises, an
    RGBA array written as .jpg raises OSError as Pillow does."""
    from PIL import Image
    import _native
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    p = str(tmp_path / "o.bmp")
    assert _native.write_image(img, p) == "BMP"
    with Image.open(p) as im:
        assert im.format == "BMP" and np.array_equal(np.asarray(im), img)
    for ext in ("jpeg", "jpe", "jfif", "JPG"):
        q = str(tmp_path / f"o.{ext}")
        assert _native.write_image(img, q) == "JPEG"
        with Image.open(q) as im:
            assert im.format == "JPEG"
    with pytest.raises(ValueError):
        _native.write_image(img, str(tmp_path / "o.unknownext"))
    with pytest.raises(OSError):
        _native.write_image(np.dstack([img, img[..., :1]]), str(tmp_path / "rgba.jpg"))
