"""GPU parity for SURVEY 8(f) rank 4: BlendingModule.feather_blend and color_correction vs the NumPy oracle
(oracle/oracle_np.py: chamfer distance transform, histogram / mean-std tables, box-filter guided filter).
Bit-exact u8 results.  PARITY UNPINNED at cv2.distanceTransform / cv2.blur (no cv2 here)."""
import numpy as np
import pytest

from oracle import oracle_np as onp

pytestmark = pytest.mark.gpu


def _scene(rng, h, w, cn=3):
    yy, xx = np.mgrid[0:h, 0:w]
    base = 128 + 70 * np.sin(xx / 19.0) + 40 * np.cos(yy / 13.0)
    img = base[..., None] + rng.integers(-25, 26, (h, w, cn)) + np.arange(cn) * 9
    return np.clip(img, 0, 255).astype(np.uint8)


def test_feather_blend_matches_oracle(rng):
    """feather_blend's distance-transform weights are evaluated by the oracle with the chamfer algorithm itself (small
    tiles) -- they come out as 1 everywhere -- and the GPU's unit-weight gather must reproduce the blend bit for bit."""
    from blending_module import BlendingModule, TileInfo
    bm = BlendingModule()
    tiles = [_scene(rng, 40, 56) for _ in range(4)]
    pos = [(0, 0), (0, 40), (28, 0), (28, 40)]
    infos = [TileInfo(t, x, y, i // 2, i % 2) for i, (t, (y, x)) in enumerate(zip(tiles, pos))]
    want = onp.feather_blend(tiles, pos)
    assert np.all(onp.feather_weight_map(40, 56) == 1.0)
    got = bm.feather_blend(infos, feather_width=13)
    assert got.shape == want.shape == (68, 96, 3) and np.array_equal(got, want)
    # bare arrays: grid positions guessed even without output_shape (blending_module.py:1299-1303); cropped canvas; gray
    got = bm.feather_blend(tiles, output_shape=(70, 100))
    assert np.array_equal(got, onp.feather_blend(tiles, [(0, 0), (0, 56), (40, 0), (40, 56)], (70, 100)))
    gray = [t[..., 0].copy() for t in tiles]
    assert np.array_equal(bm.feather_blend([TileInfo(g, x, y, 0, 0) for g, (y, x) in zip(gray, pos)]),
                          onp.feather_blend(gray, pos))
    big = [_scene(rng, 300, 200) for _ in range(2)]
    assert np.array_equal(bm.feather_blend([TileInfo(big[0], 0, 0, 0, 0), TileInfo(big[1], 150, 20, 0, 1)]),
                          onp.feather_blend(big, [(0, 0), (20, 150)]))


@pytest.mark.parametrize("method", ["histogram", "mean_std", "other"])
@pytest.mark.parametrize("local_filter", [True, False])
def test_color_correction_matches_oracle(rng, method, local_filter):
    from blending_module import BlendingModule
    bm = BlendingModule()
    img = _scene(rng, 150, 203)
    ref = np.clip(_scene(rng, 90, 77).astype(np.int16) // 2 + 60, 0, 255).astype(np.uint8)     # a different palette
    got = bm.color_correction(img, ref, method=method, local_filter=local_filter)
    want = onp.color_correction(img, ref, method=method, local_filter=local_filter)
    assert got.dtype == np.uint8 and got.shape == img.shape
    assert np.array_equal(got, want), (method, local_filter, int((got != want).sum()))
    if method == "histogram" and not local_filter:
        # the matched image's histogram follows the reference's: medians agree within a few levels
        for c in range(3):
            assert abs(float(np.median(got[..., c])) - float(np.median(ref[..., c]))) <= 6
    assert bm.color_correction(img, ref, method="none") is img


def test_color_correction_gray_edges_and_histogram(ctx, rng):
    from blending_module import BlendingModule
    bm = BlendingModule()
    g = _scene(rng, 37, 130, cn=1)[..., 0]
    r = _scene(rng, 20, 20, cn=1)[..., 0]
    assert np.array_equal(bm.color_correction(g, r), onp.color_correction(g, r))
    tiny = _scene(rng, 5, 7)                                     # smaller than the 8x8 box: iterated reflection
    assert np.array_equal(bm.color_correction(tiny, tiny[::-1].copy()), onp.color_correction(tiny, tiny[::-1].copy()))
    img = _scene(rng, 211, 97)
    d = ctx.upload(img)
    hist = ctx.histogram_u8(d.ptr, 97 * 3, 211, 97, 3)
    d.free()
    for c in range(3):
        assert np.array_equal(hist[c], np.bincount(img[..., c].ravel(), minlength=256))
    with pytest.raises(NotImplementedError):
        bm.color_correction(img.astype(np.float32), img)
    with pytest.raises(ValueError):
        bm.color_correction(img, img[..., 0])


@pytest.mark.parametrize("shape", [(16, 16, 3), (17, 64, 3), (40, 71, 1), (97, 211, 3), (33, 130, 4), (200, 300, 3), (64, 128, 3),
                                   (46, 78, 3), (47, 79, 1)])
def test_color_correction_fused_guided_filter(rng, shape, monkeypatch):
    """The one-kernel guided filter (k_cc_fused8: integer guide table -> exact 32-bit sliding sums for the first stage, a / b
    kept in LDS) against the oracle AND against the two-pass kernels it replaces (SR_CC_FUSED=0): block interiors, every
    border, ragged right / bottom edges, images barely larger than the window, 1 / 3 / 4 channels.  'none' keeps the identity
    table (guide == source), 'histogram' a non-trivial one."""
    from blending_module import BlendingModule
    bm = BlendingModule()
    h, w, cn = shape
    img = _scene(rng, h, w, cn)
    ref = np.clip(_scene(rng, 50, 60, cn).astype(np.int16) // 2 + 70, 0, 255).astype(np.uint8)
    if cn == 1:
        img, ref = img[..., 0], ref[..., 0]
    for method in ("histogram", "other"):
        want = onp.color_correction(img, ref, method=method)
        got = bm.color_correction(img, ref, method=method)
        assert np.array_equal(got, want), (shape, method, int((got != want).sum()))
        monkeypatch.setenv("SR_CC_FUSED", "0")
        two_pass = bm.color_correction(img, ref, method=method)
        monkeypatch.delenv("SR_CC_FUSED")
        assert np.array_equal(two_pass, want), (shape, method)
    flat = np.full_like(img, 77)                                 # zero variance everywhere: a = 0 / eps, b = mean
    assert np.array_equal(bm.color_correction(flat, ref, method="other"), onp.color_correction(flat, ref, method="other"))
    noise = rng.integers(0, 256, img.shape, dtype=np.uint8)      # full-range noise
    assert np.array_equal(bm.color_correction(noise, ref), onp.color_correction(noise, ref))


@pytest.mark.parametrize("shape", [(16, 16, 3), (17, 64, 3), (40, 71, 1), (97, 211, 3), (33, 130, 4), (200, 300, 3), (64, 128, 2),
                                   (46, 78, 3), (47, 79, 1)])
def test_color_correction_fused_guided_filter_float_table(ctx, rng, shape, monkeypatch):
    """The one-kernel guided filter with a FLOAT guide table (k_cc_fused8f: 'mean_std' tables whose box sums are exact in
    fp64 slide; sr_color_table_class == 2) against the oracle's ordered sums AND against the pass-structured kernels
    (SR_CC_FUSED_F=0): interiors, borders, ragged edges, 1-4 channels; tables with negative entries and entries above 255;
    a table with an entry next to zero (class 0) must take the ordered kernels and still agree."""
    import _native
    h, w, cn = shape
    img = _scene(rng, h, w, cn)
    src = img.astype(np.float32)
    base = np.arange(256, dtype=np.float32)[None, :]
    tables = {
        "mean_std": (base - rng.uniform(60, 180, (cn, 1)).astype(np.float32)) * rng.uniform(0.3, 1.9, (cn, 1)).astype(np.float32)
                    + rng.uniform(60, 180, (cn, 1)).astype(np.float32),
        "wide": (base - np.float32(140.25)) * np.float32(2.5) + np.float32(100.5),          # -250 .. 387
        "near_zero": (base - np.float32(100.0)) * np.float32(1.0009765625) + np.float32(2.0 ** -12),
    }
    classes = {k: _native.color_table_class(v if v.shape[0] == cn else np.repeat(v, cn, 0)) for k, v in tables.items()}
    assert classes["wide"] == 2 and classes["near_zero"] == 0, classes
    for name, lut in tables.items():
        lut = np.ascontiguousarray(lut if lut.shape[0] == cn else np.repeat(lut, cn, 0))
        corrected = np.stack([lut[c][img[..., c]] for c in range(cn)], axis=-1).astype(np.float32)
        want = np.clip(onp.simple_guided_filter(corrected, src, 8, 0.01), 0, 255).astype(np.uint8)

        def run():
            d, o = ctx.upload(img), ctx.alloc(img.size)
            ctx.color_correct_u8(d.ptr, w * cn, h, w, cn, lut, 1, 8, 0.01, o.ptr, w * cn)
            got = ctx.download(o.ptr, img.shape, np.uint8)
            d.free(); o.free()
            return got
        got = run()
        assert np.array_equal(got, want), (shape, name, classes[name], int((got != want).sum()))
        monkeypatch.setenv("SR_CC_FUSED_F", "0")
        two_pass = run()
        monkeypatch.delenv("SR_CC_FUSED_F")
        assert np.array_equal(two_pass, want), (shape, name)


def test_color_correction_small_eps_takes_ieee_division(ctx, rng):
    """eps below 0.005 (the integer kernel's bare fma-chain division is only proven for the reference's 0.01): an integer
    table then runs through the float-table kernel and its IEEE division; fused and pass-structured results agree with the
    oracle."""
    img = _scene(rng, 70, 131)
    lut = np.tile(np.arange(256, dtype=np.float32), (3, 1))
    for eps in (1e-3, 1e-4):
        d, o = ctx.upload(img), ctx.alloc(img.size)
        ctx.color_correct_u8(d.ptr, 131 * 3, 70, 131, 3, lut, 1, 8, eps, o.ptr, 131 * 3)
        got = ctx.download(o.ptr, img.shape, np.uint8)
        d.free(); o.free()
        want = np.clip(onp.simple_guided_filter(img.astype(np.float32), img.astype(np.float32), 8, eps), 0, 255).astype(np.uint8)
        assert np.array_equal(got, want), (eps, int((got != want).sum()))


@pytest.mark.parametrize("mode,guided", [(1, "simple"), (2, "ximgproc")])
def test_color_correct_in_place(ctx, rng, mode, guided):
    """sr_color_correct_u8 with the output buffer ON the input: the fused kernels read a block's halo while other blocks store,
    so an in-place call must take the pass-structured kernels (which convert / filter before anything is stored) and still
    give the oracle's bytes."""
    img = _scene(rng, 96, 160)
    lut = np.tile(np.arange(256, dtype=np.float32), (3, 1))
    lut[1] = np.clip(np.arange(256) * 0.5 + 40, 0, 255).astype(np.int32)          # an integer table that is not the identity
    d = ctx.upload(img)
    ctx.color_correct_u8(d.ptr, 160 * 3, 96, 160, 3, lut, mode, 8, 0.01, d.ptr, 160 * 3)
    got = ctx.download(d.ptr, img.shape, np.uint8)
    d.free()
    corrected = np.stack([lut[c][img[..., c]] for c in range(3)], axis=-1).astype(np.float32)
    f = onp.simple_guided_filter if guided == "simple" else onp.guided_filter_ximgproc
    want = np.clip(f(corrected, img.astype(np.float32), 8, 0.01), 0, 255).astype(np.uint8)
    assert np.array_equal(got, want), int((got != want).sum())


def test_mixed_channel_tiles_raise(rng):
    from blending_module import BlendingModule, TileInfo
    bm = BlendingModule()
    a, b = _scene(rng, 32, 32, 3), _scene(rng, 32, 32, 4)
    with pytest.raises(ValueError):
        bm.laplacian_fusion([TileInfo(a, 0, 0, 0, 0), TileInfo(b, 16, 0, 0, 1)])


@pytest.mark.parametrize("method", ["histogram", "mean_std"])
def test_color_correction_ximgproc_branch(rng, method, monkeypatch):
    """The try-branch of _guided_filter (blending_module.py:1108-1111: cv2.ximgproc.guidedFilter, what runs with
    opencv-contrib installed), selected with guided_filter='ximgproc': (2 r + 1)^2 window, colour guide with the per-pixel
    3 x 3 covariance inverse.  Bit-exact vs oracle_np.guided_filter_ximgproc, which restates it -- PARITY UNPINNED (no cv2
    here, the reference holds no fixture).  Also: gray images, an image smaller than the 17 x 17 window, and that the two
    branches really differ."""
    from blending_module import BlendingModule
    bx = BlendingModule(guided_filter="ximgproc")
    img = _scene(rng, 120, 171)
    ref = np.clip(_scene(rng, 90, 77).astype(np.int16) // 2 + 60, 0, 255).astype(np.uint8)
    got = bx.color_correction(img, ref, method=method)
    want = onp.color_correction(img, ref, method=method, guided="ximgproc")
    assert got.dtype == np.uint8 and got.shape == img.shape
    assert np.array_equal(got, want), int((got != want).sum())
    assert not np.array_equal(got, BlendingModule().color_correction(img, ref, method=method))
    g, r = _scene(rng, 41, 99, cn=1)[..., 0], _scene(rng, 20, 20, cn=1)[..., 0]
    assert np.array_equal(bx.color_correction(g, r, method=method), onp.color_correction(g, r, method=method, guided="ximgproc"))
    tiny = _scene(rng, 6, 9)
    assert np.array_equal(bx.color_correction(tiny, tiny[::-1].copy(), method=method),
                          onp.color_correction(tiny, tiny[::-1].copy(), method=method, guided="ximgproc"))
    wide = _scene(rng, 130, 256)                                 # width a multiple of 4: the 17 x 17 box kernel's interior blocks
    want_wide = onp.color_correction(wide, ref, method=method, guided="ximgproc")
    assert np.array_equal(bx.color_correction(wide, ref, method=method), want_wide)
    # histogram matching has an integer guide table: its first stage is ONE kernel of exact sliding sums (k_gfx_coeff17);
    # SR_GF_FUSED=0 takes the box-mean launches instead -- the same bytes either way
    monkeypatch.setenv("SR_GF_FUSED", "0")
    assert np.array_equal(bx.color_correction(wide, ref, method=method), want_wide)
    assert np.array_equal(bx.color_correction(img, ref, method=method), want)
    monkeypatch.delenv("SR_GF_FUSED")
    noise = rng.integers(0, 256, (97, 203, 3), dtype=np.uint8)
    assert np.array_equal(bx.color_correction(noise, ref, method=method), onp.color_correction(noise, ref, method=method, guided="ximgproc"))
    with pytest.raises(ValueError):
        BlendingModule(guided_filter="bilateral")
