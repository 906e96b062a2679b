"""Cross-checks of the oracle's OpenCV-defined functions against INDEPENDENT library implementations that are
importable offline (scipy.ndimage, torch): not pins of OpenCV's bit pattern -- cv2 is absent and the reference holds no
pixel fixtures, so those stay "parity unpinned" -- but of the semantics the restatement assumes: kernel coefficients,
sample positions, border rules.  Each check says what it covers and how closely the two must agree."""
import numpy as np
import pytest

from oracle import oracle_np as onp


def test_pyr_up_vs_transposed_conv():
    """cv2.pyrUp = zero-insertion + 5x5 binomial / 64: F.conv_transpose2d with stride 2 is that operation.  Interior
    only (OpenCV's own border rule at the last row / column is what the restatement spells out)."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(21)
    a = rng.uniform(0, 255, (23, 31)).astype(np.float32)
    k1 = torch.tensor([1., 4., 6., 4., 1.], dtype=torch.float64)
    k = (k1[:, None] * k1[None, :] / 64.0)[None, None]
    up = F.conv_transpose2d(torch.from_numpy(a.astype(np.float64))[None, None], k, stride=2, padding=2, output_padding=1)[0, 0].numpy()
    got = onp.pyr_up(a, (46, 62))
    np.testing.assert_allclose(got[2:-3, 2:-3], up[2:-3, 2:-3], rtol=2e-6, atol=1e-4)


def test_gaussian_blur_branch_b_vs_scipy():
    """SSIM branch B (_calculate_ssim_simple, quality_assessment_module.py:391-417): cv2.GaussianBlur((11, 11), 1.5) on
    float64 with BORDER_REFLECT_101 == scipy.ndimage.gaussian_filter(sigma 1.5, radius 5, mode='mirror') (same
    normalised 11-tap kernel, same border): the whole SSIM value to 1e-12."""
    from scipy import ndimage
    rng = np.random.default_rng(22)
    x = rng.integers(0, 256, (57, 83)).astype(np.float64)
    y = np.clip(x + rng.normal(0, 9, x.shape), 0, 255).round()
    f = lambda a: ndimage.gaussian_filter(a, 1.5, truncate=3.5, mode="mirror")
    assert int(3.5 * 1.5 + 0.5) == 5
    np.testing.assert_allclose(onp._filter_sep(x, onp.cv_gaussian_kernel(11, 1.5), "reflect101"), f(x), rtol=1e-12, atol=1e-10)
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    mu1, mu2 = f(x), f(y)
    s1, s2, s12 = f(x * x) - mu1 ** 2, f(y * y) - mu2 ** 2, f(x * y) - mu1 * mu2
    want = (((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 ** 2 + mu2 ** 2 + c1) * (s1 + s2 + c2))).mean()
    assert abs(onp.ssim(x.astype(np.uint8), y.astype(np.uint8), "simple") - want) < 1e-12


def test_ssim_branch_a_filters_vs_scipy():
    """Branch A (skimage): uniform_filter(size 7) and gaussian_filter(sigma 1.5, truncate 3.5) with scipy's default
    mode='reflect' -- the filters skimage 0.18.3 calls -- rebuilt here from scipy and compared with the oracle's SSIM
    (the skimage golden values in tests/golden pin the same thing on fixed images)."""
    from scipy import ndimage
    rng = np.random.default_rng(23)
    x = rng.integers(0, 256, (64, 71)).astype(np.float64)
    y = np.clip(x + rng.normal(0, 12, x.shape), 0, 255).round()
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    for mode, f, norm, pad in (("uniform", lambda a: ndimage.uniform_filter(a, size=7), 49.0 / 48.0, 3),
                               ("gauss", lambda a: ndimage.gaussian_filter(a, 1.5, truncate=3.5), 1.0, 5)):
        ux, uy = f(x), f(y)
        vx, vy, vxy = norm * (f(x * x) - ux * ux), norm * (f(y * y) - uy * uy), norm * (f(x * y) - ux * uy)
        s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
        want = s[pad:-pad, pad:-pad].mean()
        assert abs(onp.ssim(x.astype(np.uint8), y.astype(np.uint8), mode) - want) < 1e-10, mode


@pytest.mark.parametrize("dst", [(160, 120), (200, 77), (40, 30), (33, 50)])
def test_resize_vs_torch_interpolate(dst):
    """cv2.resize INTER_CUBIC / INTER_LINEAR sample at (i + 0.5) * scale - 0.5 with the a = -0.75 cubic and replicate
    borders, no antialiasing -- what F.interpolate(mode='bicubic' / 'bilinear', align_corners=False) computes in float.
    OpenCV's u8 path works in 11-bit fixed point and rounds to u8: the restatement stays within 0.8 grey levels of the
    float result (rounding is 0.5 of that; observed maxima 0.50-0.76), mean deviation 0.25 = pure rounding."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(24)
    yy, xx = np.mgrid[0:60, 0:80]
    a = np.clip(128 + 70 * np.sin(xx / 7.0)[..., None] + 50 * np.cos(yy / 5.0)[..., None] + rng.integers(-20, 21, (60, 80, 3)), 0, 255).astype(np.uint8)
    dw, dh = dst
    t = torch.from_numpy(a.astype(np.float64)).permute(2, 0, 1)[None]
    cub = F.interpolate(t, size=(dh, dw), mode="bicubic", align_corners=False)[0].permute(1, 2, 0).numpy()
    lin = F.interpolate(t, size=(dh, dw), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
    got_c = onp.resize_cubic_u8(a, dw, dh).astype(np.float64)
    got_l = onp.resize_linear_u8(a, dw, dh).astype(np.float64)
    assert np.abs(got_c - np.clip(cub, 0, 255)).max() <= 0.8
    assert np.abs(got_l - lin).max() <= 0.8
    assert np.abs(got_c - np.clip(cub, 0, 255)).mean() < 0.3 and np.abs(got_l - lin).mean() < 0.3


def test_gray_vs_float_formula():
    """cv2.cvtColor RGB2GRAY: 0.299 R + 0.587 G + 0.114 B in 15-bit (14-bit in older builds) fixed point, rounded."""
    rng = np.random.default_rng(25)
    a = rng.integers(0, 256, (50, 60, 3), dtype=np.uint8)
    want = a[..., 0] * 0.299 + a[..., 1] * 0.587 + a[..., 2] * 0.114
    for shift in (14, 15):
        assert np.abs(onp.rgb2gray_u8(a, shift).astype(np.float64) - want).max() <= 0.51


def test_distance_transform_vs_scipy_edt():
    """cv2.distanceTransform(DIST_L2, 5) is a 5x5 chamfer approximation of the Euclidean distance: within 2.5 % of
    scipy's exact EDT (the published bound of the (1, 1.4, 2.1969) mask), exact along rows and columns."""
    from scipy import ndimage
    rng = np.random.default_rng(26)
    m = np.ones((41, 53), np.uint8)
    m[rng.integers(0, 41, 6), rng.integers(0, 53, 6)] = 0
    d = onp.distance_transform_l2_5(m)
    e = ndimage.distance_transform_edt(m)
    nz = e > 0
    assert np.abs(d[nz] / e[nz] - 1.0).max() < 0.025
    assert np.array_equal(d == 0, e == 0)


def test_histogram_lut_vs_cdf_matching():
    """color_correction's histogram matching (blending_module.py:1040-1075): for every source level the first reference
    level whose CDF reaches the source CDF -- np.searchsorted on the normalised CDFs."""
    rng = np.random.default_rng(27)
    src = np.bincount(rng.integers(0, 200, 5000), minlength=256)
    ref = np.bincount(rng.integers(40, 256, 7000), minlength=256)
    lut = onp.histogram_lut(src, ref)
    cs, cr = np.cumsum(src) / src.sum(), np.cumsum(ref) / ref.sum()
    want = np.clip(np.searchsorted(cr, cs, side="left"), 0, 255)
    occupied = src > 0
    assert np.abs(lut[occupied].astype(int) - want[occupied]).max() <= 1


def test_guided_filters_vs_scipy_uniform_filter():
    """Both branches of BlendingModule._guided_filter (blending_module.py:1092-1146) rebuilt from scipy.ndimage.uniform_filter in
    float64 -- He et al.'s formulas, nothing shared with the oracle's box sums: `_simple_guided_filter`'s cv2.blur((8, 8)) is an
    8-wide window anchored at 4 with BORDER_REFLECT_101 (scipy: size 8, mode 'mirror' -- its even window also spans
    i - 4 .. i + 3), cv2.ximgproc.guidedFilter's a (2 r + 1)-wide window with BORDER_REFLECT (scipy 'reflect'), gray and colour
    guide.  Fixes window, anchor, border and the algebra; float32 vs float64 rounding is the only difference left."""
    from scipy.ndimage import uniform_filter
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:61, 0:83]
    src = np.clip(120 + 60 * np.sin(xx / 9.0) + 40 * np.cos(yy / 7.0) + rng.integers(-20, 21, (61, 83)), 0, 255).astype(np.float32)
    guide = np.clip(src * 0.8 + 30 + rng.integers(-6, 7, src.shape), 0, 255).astype(np.float32)
    eps = 0.01

    def gf(g, p, box):
        g, p = g.astype(np.float64), p.astype(np.float64)
        mg, mp = box(g), box(p)
        a = (box(g * p) - mg * mp) / (box(g * g) - mg * mg + eps)
        b = mp - a * mg
        return box(a) * g + box(b)

    want = gf(guide, src, lambda x: uniform_filter(x, size=8, mode="mirror"))
    got = onp.simple_guided_filter(guide, src, 8, eps)
    assert np.max(np.abs(got - want)) < 2e-3 and np.mean(np.abs(got - want)) < 2e-4      # measured 1.4e-4 / 1.3e-5: float32 rounding
    want = gf(guide, src, lambda x: uniform_filter(x, size=17, mode="reflect"))
    got = onp.guided_filter_ximgproc(guide, src, 8, eps)
    assert np.max(np.abs(got - want)) < 2e-3 and np.mean(np.abs(got - want)) < 2e-4      # measured 5e-5 / 7e-6
    # colour guide: per pixel (Sigma + eps I)^-1 cov(I, p), solved with numpy.linalg in float64
    I = np.stack([guide, np.clip(guide[::-1] * 0.5 + 60, 0, 255), np.clip(255 - guide * 0.7, 0, 255)], axis=-1).astype(np.float32)
    P = np.stack([src, src[:, ::-1].copy(), np.clip(src * 0.5 + 40, 0, 255)], axis=-1).astype(np.float32)
    box = lambda x: uniform_filter(x, size=17, mode="reflect")
    I64, P64 = I.astype(np.float64), P.astype(np.float64)
    m = np.stack([box(I64[..., i]) for i in range(3)], axis=-1)
    S = np.empty(I.shape[:2] + (3, 3))
    for i in range(3):
        for j in range(3):
            S[..., i, j] = box(I64[..., i] * I64[..., j]) - m[..., i] * m[..., j] + (eps if i == j else 0.0)
    want3 = np.empty_like(P64)
    for c in range(3):
        mp = box(P64[..., c])
        cov = np.stack([box(I64[..., i] * P64[..., c]) - m[..., i] * mp for i in range(3)], axis=-1)
        a = np.linalg.solve(S, cov[..., None])[..., 0]
        b = mp - np.sum(a * m, axis=-1)
        want3[..., c] = sum(box(a[..., i]) * I64[..., i] for i in range(3)) + box(b)
    got3 = onp.guided_filter_ximgproc(I, P, 8, eps)
    d3 = np.abs(got3 - want3)                                    # measured 0.025 / 0.003 grey levels: the float32 cofactor inverse of
    assert d3.max() < 0.1 and d3.mean() < 6e-3                   # a covariance of three strongly correlated guide channels
