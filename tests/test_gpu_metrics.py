"""GPU parity for the quality metrics: HIP reductions vs the CPU oracle and vs scikit-image 0.18.3
golden values (tests/golden/metrics_skimage.npz).

Tolerances: PSNR is an exact integer sum -> identical doubles; SSIM is fp64 with a different
summation order than scipy's running-sum filters -> 1e-9 relative (north_star bar: 1e-4)."""
import os

import numpy as np
import pytest

from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "metrics_skimage.npz")


def _pair(rng, h, w, sigma=6.0):
    yy, xx = np.mgrid[0:h, 0:w]
    base = (128 + 64 * np.sin(xx / 37.0) + 48 * np.cos(yy / 23.0))[..., None]
    a = np.clip(base + rng.integers(-12, 13, (h, w, 3)), 0, 255).astype(np.uint8)
    b = np.clip(a.astype(np.float32) + rng.normal(0, sigma, (h, w, 3)), 0, 255).astype(np.uint8)
    return a, b


def _gpu_psnr(ctx, a, b):
    import _native
    da, db = ctx.upload(a), ctx.upload(b)
    rowlen = a.size // a.shape[0]
    sse = ctx.sse_u8(da.ptr, rowlen, db.ptr, rowlen, a.shape[0], rowlen)
    da.free(); db.free()
    return sse, _native.psnr_from_sse(sse, a.size, 255.0)


def _gpu_ssim(ctx, a, b, mode, **kw):
    cn = a.shape[2] if a.ndim == 3 else 1
    da, db = ctx.upload(a), ctx.upload(b)
    s, n = ctx.ssim_u8(da.ptr, a.shape[1] * cn, db.ptr, b.shape[1] * cn, a.shape[0], a.shape[1], cn, mode, **kw)
    da.free(); db.free()
    return s / n, n


@pytest.mark.parametrize("shape", [(64, 64), (193, 257), (512, 512), (1000, 333)])
def test_psnr_matches_oracle(ctx, rng, shape):
    a, b = _pair(rng, *shape)
    sse, p = _gpu_psnr(ctx, a, b)
    assert sse == int(((a.astype(np.int64) - b.astype(np.int64)) ** 2).sum())
    assert p == oc.psnr(a, b)
    assert _gpu_psnr(ctx, a, a)[1] == float("inf")


def test_psnr_strided_crop(ctx, rng):
    """calculate_psnr crops both images to the common top-left rectangle (quality_assessment_module.py:304-308)."""
    a, _ = _pair(rng, 90, 120)
    _, b = _pair(rng, 100, 110)
    da, db = ctx.upload(a), ctx.upload(b)
    sse = ctx.sse_u8(da.ptr, 120 * 3, db.ptr, 110 * 3, 90, 110 * 3)
    ac, bc = a[:90, :110], b[:90, :110]
    assert sse == int(((ac.astype(np.int64) - bc.astype(np.int64)) ** 2).sum())


@pytest.mark.parametrize("mode", ["uniform", "gauss", "simple"])
@pytest.mark.parametrize("shape", [(64, 80), (193, 257), (300, 411)])
def test_ssim_matches_oracle(ctx, rng, mode, shape):
    a, b = _pair(rng, *shape)
    for shift in (15, 14):
        ga, gb = oc.rgb2gray_u8(a, shift), oc.rgb2gray_u8(b, shift)
        ref = oc.ssim(ga, gb, mode)
        got, n = _gpu_ssim(ctx, a, b, mode, gray_shift=shift)
        pad = {"uniform": 3, "gauss": 5, "simple": 0}[mode]
        assert n == (shape[0] - 2 * pad) * (shape[1] - 2 * pad)
        assert abs(got - ref) <= 1e-9 * abs(ref), (got, ref)
        got_gray, _ = _gpu_ssim(ctx, ga, gb, mode)
        assert abs(got_gray - ref) <= 1e-9 * abs(ref)


def test_ssim_row_ranges_add_up(ctx, rng):
    a, b = _pair(rng, 200, 150)
    full, n = _gpu_ssim(ctx, a, b, "gauss")
    parts, cnt = 0.0, 0
    for r0, r1 in [(0, 37), (37, 120), (120, 200)]:
        m, k = _gpu_ssim(ctx, a, b, "gauss", row_begin=r0, row_end=r1)
        parts += m * k
        cnt += k
    assert cnt == n
    assert abs(parts / cnt - full) < 1e-12


def test_metrics_vs_skimage_golden(ctx):
    z = np.load(GOLD)
    for name in z["cases"]:
        a, b = z[f"{name}_a"], z[f"{name}_b"]
        assert _gpu_psnr(ctx, a, b)[1] == pytest.approx(float(z[f"{name}_psnr"]), rel=1e-12)
        ga, gb = np.ascontiguousarray(a[..., 1]), np.ascontiguousarray(b[..., 1])
        assert _gpu_ssim(ctx, ga, gb, "uniform")[0] == pytest.approx(float(z[f"{name}_ssim_uniform"]), rel=1e-9)
        assert _gpu_ssim(ctx, ga, gb, "gauss")[0] == pytest.approx(float(z[f"{name}_ssim_gauss"]), rel=1e-9)
    np.random.seed(42)
    o = np.random.randint(0, 256, (512, 512, 3), dtype=np.uint8)
    u = np.clip(o.astype(np.float32) + np.random.randn(512, 512, 3) * 5, 0, 255).astype(np.uint8)
    assert _gpu_psnr(ctx, o, u)[1] == pytest.approx(float(z["ex_psnr"]), rel=1e-12)
    g0, g1 = np.ascontiguousarray(o[..., 0]), np.ascontiguousarray(u[..., 0])
    assert _gpu_ssim(ctx, g0, g1, "uniform")[0] == pytest.approx(float(z["ex_ssim_uniform_ch0"]), rel=1e-9)
    assert _gpu_ssim(ctx, g0, g1, "gauss")[0] == pytest.approx(float(z["ex_ssim_gauss_ch0"]), rel=1e-9)


def test_rgb2gray_and_resize(ctx, rng):
    a, _ = _pair(rng, 120, 171)
    da = ctx.upload(a)
    dg = ctx.alloc(120 * 171)
    for shift in (15, 14):
        ctx.rgb2gray_u8(da.ptr, 171 * 3, 120, 171, dg.ptr, 171, shift)
        assert np.array_equal(ctx.download(dg.ptr, (120, 171), np.uint8), oc.rgb2gray_u8(a, shift))
    for (dh, dw) in [(12, 17), (48, 68), (247, 353), (120, 171)]:
        dd = ctx.alloc(dh * dw * 3)
        ctx.resize_cubic_u8(da.ptr, 171 * 3, 120, 171, 3, dd.ptr, dw * 3, dh, dw)
        ref = oc.resize_cubic_u8(a, dw, dh)
        assert np.array_equal(ctx.download(dd.ptr, (dh, dw, 3), np.uint8), ref)
        # window form == crop of the full result
        x0, y0, ww, wh = dw // 3, dh // 4, dw // 2, dh // 2
        dwin = ctx.alloc(ww * wh * 3)
        ctx.resize_cubic_window_u8(da.ptr, 171 * 3, 120, 171, 3, dh, dw, x0, y0, ww, wh, dwin.ptr, ww * 3)
        assert np.array_equal(ctx.download(dwin.ptr, (wh, ww, 3), np.uint8), ref[y0:y0 + wh, x0:x0 + ww])


@pytest.mark.parametrize("mode", ["mirror", "replicate", "reflect", "constant"])
def test_tile_extract_pad(ctx, rng, mode):
    import _native
    img = rng.integers(0, 256, (150, 211, 3), dtype=np.uint8)
    block, ov = 64, 12
    xywh = _native.tile_plan(211, 150, block, ov)
    dimg = ctx.upload(img)
    dt = ctx.alloc(len(xywh) * block * block * 3)
    ctx.tile_extract_pad(dimg.ptr, 150, 211, 3, 211 * 3, xywh, block, mode, dt.ptr)
    got = ctx.download(dt.ptr, (len(xywh), block, block, 3), np.uint8)
    for i, (x, y, w, h) in enumerate(xywh):
        assert np.array_equal(got[i], oc.tile_extract_pad(img, x, y, w, h, block, mode)), (i, x, y, w, h)


def test_tile_extract_pad_iterated_reflection(ctx, rng):
    """1280x720 / block 2048: the pad (1328 rows) exceeds the image, reflection iterates."""
    import _native
    img = rng.integers(0, 256, (45, 80, 3), dtype=np.uint8)
    block = 128
    xywh = _native.tile_plan(80, 45, block, 25)
    assert xywh == [(0, 0, 80, 45)]
    dimg, dt = ctx.upload(img), ctx.alloc(block * block * 3)
    for mode in ("mirror", "reflect", "replicate"):
        ctx.tile_extract_pad(dimg.ptr, 45, 80, 3, 80 * 3, xywh, block, mode, dt.ptr)
        assert np.array_equal(ctx.download(dt.ptr, (block, block, 3), np.uint8),
                              oc.tile_extract_pad(img, 0, 0, 80, 45, block, mode))


@pytest.mark.parametrize("shape,dst", [((300, 411, 3), (111, 152)), ((193, 257, 3), (19, 25)), ((128, 640), (51, 256)),
                                         ((97, 83, 3), (97, 83)), ((64, 64, 3), (5, 6)),
                                         # gentle down-sampling (the LDS-free column march, k_resize_gray_pair_march): ragged
                                         # segments of 16 rows / 64 columns, border columns, scales 0.3 .. 0.9
                                         ((257, 300, 3), (129, 151)), ((100, 700, 3), (77, 333)), ((333, 90, 3), (100, 81)),
                                         ((70, 200, 3), (65, 66)),
                                         # windows further apart than four rows (the march jumps), mixed steps around 4
                                         ((400, 330, 3), (80, 66)), ((391, 345, 3), (100, 90)), ((300, 640, 3), (61, 128)),
                                         # scales near 0.1: 32 destination columns per wave
                                         ((600, 1000, 3), (60, 100)), ((410, 777, 3), (50, 90))])
def test_assess_resized_equals_resize_then_assess(ctx, rng, shape, dst):
    """sr_assess_resized_u8 (bicubic resize sampled on the fly inside the metric kernel) gives the
    sums of sr_resize_cubic_u8 on both images followed by sr_assess_u8 -- and those of the oracle's resize + metrics."""
    import _native
    a = rng.integers(0, 256, shape, dtype=np.uint8)
    b = np.clip(a.astype(np.int16) + rng.integers(-9, 10, shape), 0, 255).astype(np.uint8)
    h, w = shape[:2]
    cn = 3 if len(shape) == 3 else 1
    dh, dw = dst
    da, db = ctx.upload(a), ctx.upload(b)
    flags = _native.ASSESS_ALL
    got = ctx.assess_resized_u8(da.ptr, w * cn, db.ptr, w * cn, h, w, cn, dh, dw, flags=flags)
    ra, rb = ctx.alloc(dh * dw * cn), ctx.alloc(dh * dw * cn)
    ctx.resize_cubic_u8(da.ptr, w * cn, h, w, cn, ra.ptr, dw * cn, dh, dw)
    ctx.resize_cubic_u8(db.ptr, w * cn, h, w, cn, rb.ptr, dw * cn, dh, dw)
    want = ctx.assess_u8(ra.ptr, dw * cn, rb.ptr, dw * cn, dh, dw, cn, flags=flags)
    assert got["sse"] == want["sse"]
    for k in ("ssim_uniform", "ssim_gauss", "ssim_simple"):
        assert got[k] == pytest.approx(want[k], rel=1e-13, abs=1e-300), k
    # against the CPU oracle
    oa, ob = oc.resize_cubic_u8(a, dw, dh), oc.resize_cubic_u8(b, dw, dh)
    assert got["sse"] == float(np.sum((oa.astype(np.int64) - ob.astype(np.int64)) ** 2))
    if dh >= 7 and dw >= 7:
        g0 = oc.rgb2gray_u8(oa) if cn == 3 else oa
        g1 = oc.rgb2gray_u8(ob) if cn == 3 else ob
        n = _native.ssim_count(dh, dw, "uniform")
        assert got["ssim_uniform"] / n == pytest.approx(oc.ssim(g0, g1, "uniform"), rel=1e-9)
    for buf in (da, db, ra, rb):
        buf.free()
