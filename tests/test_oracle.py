"""CPU: the oracle against the golden vectors, against itself (NumPy vs C, bit for bit), against an
independent torch convolution, and against the algebraic properties the domain offers."""
import json
import os

import numpy as np
import pytest

from oracle import oracle_c as oc
from oracle import oracle_np as onp

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_bookkeeping_golden():
    g = json.load(open(os.path.join(GOLD, "bookkeeping.json")))
    for case in g["tiling"]:
        w, h, block, ov = case["w"], case["h"], case["block"], case["overlap_px"]
        pos = onp.tile_positions(w, h, block, ov)
        assert [list(p) for p in pos] == case["positions"]
        assert [list(onp.tile_overlaps(*p, w, h, block, ov)) for p in pos] == case["overlaps"]
    for case in g["target_size"]:
        assert list(onp.target_size(tuple(case["size"]), case["preset"])) == case["target"]
    assert onp.target_size((800, 600), "4000x3000") == (4000, 3000)
    assert onp.target_size((800, 600), "garbage") == (12245, 8163)


def test_metrics_vs_skimage_golden():
    z = np.load(os.path.join(GOLD, "metrics_skimage.npz"))
    assert str(z["skimage_version"]) == "0.18.3"
    for name in z["cases"]:
        a, b = z[f"{name}_a"], z[f"{name}_b"]
        for impl in (onp, oc):
            assert impl.psnr(a, b) == pytest.approx(float(z[f"{name}_psnr"]), rel=1e-13)
        ga, gb = np.ascontiguousarray(a[..., 1]), np.ascontiguousarray(b[..., 1])
        assert oc.ssim(ga, gb, "uniform") == pytest.approx(float(z[f"{name}_ssim_uniform"]), rel=1e-10)
        assert oc.ssim(ga, gb, "gauss") == pytest.approx(float(z[f"{name}_ssim_gauss"]), rel=1e-10)
        if ga.size <= 64 * 64:
            assert onp.ssim(ga, gb, "uniform") == pytest.approx(float(z[f"{name}_ssim_uniform"]), rel=1e-10)
            assert onp.ssim(ga, gb, "gauss") == pytest.approx(float(z[f"{name}_ssim_gauss"]), rel=1e-10)
    # the reference's own example pair (quality_assessment_module.py:1394-1400)
    np.random.seed(42)
    o = np.random.randint(0, 256, (512, 512, 3), dtype=np.uint8)
    u = np.clip(o.astype(np.float32) + np.random.randn(512, 512, 3) * 5, 0, 255).astype(np.uint8)
    assert oc.psnr(o, u) == pytest.approx(34.19200765819827, rel=1e-14)
    assert oc.ssim(np.ascontiguousarray(o[..., 0]), np.ascontiguousarray(u[..., 0]), "uniform") == \
        pytest.approx(0.9977095724090179, rel=1e-10)
    assert oc.ssim(np.ascontiguousarray(o[..., 0]), np.ascontiguousarray(u[..., 0]), "gauss") == \
        pytest.approx(0.9976671706617938, rel=1e-10)
    assert oc.psnr(o, o) == float("inf")


@pytest.mark.parametrize("shape", [(37, 53, 3), (64, 64, 3), (130, 96), (2, 2, 3), (3, 5), (9, 2, 3), (1, 7)])
def test_numpy_and_c_oracles_agree_bitwise(shape):
    rng = np.random.default_rng(3)
    a = rng.uniform(0, 255, shape).astype(np.float32)
    if min(shape[:2]) >= 1:
        d1, d2 = onp.pyr_down(a), oc.pyr_down(a)
        assert np.array_equal(d1, d2)
        assert np.array_equal(onp.pyr_up(d1, a.shape[:2]), oc.pyr_up(d1, a.shape[:2]))


def test_pyr_down_vs_torch_conv():
    """Independent implementation: reflect pad + strided conv2d with the 5x5 binomial kernel."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(5)
    a = rng.uniform(0, 255, (67, 90)).astype(np.float32)
    k1 = torch.tensor([1., 4., 6., 4., 1.], dtype=torch.float64)
    k = (k1[:, None] * k1[None, :] / 256.0)[None, None]
    t = F.pad(torch.from_numpy(a.astype(np.float64))[None, None], (2, 2, 2, 2), mode="reflect")
    ref = F.conv2d(t, k, stride=2)[0, 0].numpy()
    np.testing.assert_allclose(oc.pyr_down(a), ref, rtol=2e-6, atol=1e-4)


def test_pyramid_properties():
    rng = np.random.default_rng(11)
    img = rng.uniform(0, 255, (75, 101, 3)).astype(np.float32)
    gp = onp.build_gaussian_pyramid(img, 6)
    assert [g.shape[:2] for g in gp] == [(75, 101), (38, 51), (19, 26), (10, 13), (5, 7), (3, 4)]
    lp = onp.build_laplacian_pyramid(gp)
    np.testing.assert_allclose(onp.collapse_laplacian_pyramid(lp), img, atol=1e-3)      # perfect reconstruction
    const = np.full((40, 56), 77.0, np.float32)
    assert np.all(oc.pyr_down(const) == 77.0)                                           # partition of unity
    assert np.all(oc.pyr_up(oc.pyr_down(const), const.shape) == 77.0)
    # stop rule: a side < 2 ends the pyramid (blending_module.py:250-252)
    assert len(onp.build_gaussian_pyramid(np.zeros((3, 40), np.float32), 6)) == 3


def test_weight_map_and_lut():
    for (h, w) in [(64, 64), (97, 61), (30, 200)]:
        for wt in ("cosine", "linear", "sigmoid"):
            m = onp.distance_weight_map(h, w, wt)
            fw = min(h, w) // 8
            lut = oc.weight_lut(fw, wt)
            y, x = np.arange(h)[:, None], np.arange(w)[None, :]
            d = np.minimum(np.minimum(y, h - 1 - y), np.minimum(x, w - 1 - x))
            assert np.array_equal(lut[np.minimum(d, fw)], m)
    m = onp.distance_weight_map(64, 64, "cosine")
    assert m[0, :].max() == 0 and m[:, 0].max() == 0 and m[32, 32] == 1      # outer ring 0, interior 1


def test_blend_oracles_agree_and_fixture_regenerates():
    z = np.load(os.path.join(GOLD, "blend_small.npz"))
    tiles = [z[f"tile{i}"] for i in range(int(z["n"]))]
    pos = [tuple(p) for p in z["pos"]]
    for impl in (onp, oc):
        u8, f = impl.laplacian_fusion(tiles, pos, tuple(z["shape"]), int(z["levels"]), "cosine", return_float=True)
        assert np.array_equal(u8, z["canvas_u8"]) and np.array_equal(f, z["canvas_f32"])
    a = onp.weighted_average_fusion(tiles, pos, tuple(z["shape"]), "cosine", return_float=True)
    b = oc.weighted_average_fusion(tiles, pos, tuple(z["shape"]), "cosine", return_float=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_blend_properties():
    """With one level the blend is a weighted average, so identical overlapping content comes back
    unchanged away from the zero-weight rim (with more levels the reference's algorithm does NOT have
    this property for small tiles: coarse levels are weighted by blurred weights but normalised by the
    level-0 weights only -- DESIGN.md 'reference quirks').  A missing tile leaves a hole of zeros
    (sum w = 0 -> clamp 1e-6, SURVEY 5 'failed tiles')."""
    rng = np.random.default_rng(2)
    yy, xx = np.mgrid[0:200, 0:260]
    img = np.clip(128 + 60 * np.sin(xx / 19.0) + 50 * np.cos(yy / 13.0) + rng.integers(-5, 6, (200, 260)), 0, 255)
    img = np.stack([img, img[::-1], img[:, ::-1]], -1).astype(np.uint8)
    rects = [(0, 0, 150, 120), (110, 0, 150, 120), (0, 80, 150, 120), (110, 80, 150, 120)]
    tiles = [np.ascontiguousarray(img[y:y + h, x:x + w]) for (x, y, w, h) in rects]
    pos = [(y, x) for (x, y, _, _) in rects]
    out, f = oc.laplacian_fusion(tiles, pos, (200, 260), 1, "cosine", return_float=True)
    inner = (slice(2, -2), slice(2, -2))
    assert np.abs(f[inner] - img[inner]).max() < 1e-3
    assert np.array_equal(out[inner], img[inner]) or np.abs(out[inner].astype(int) - img[inner]).max() <= 1
    out3 = oc.laplacian_fusion(tiles[:3], pos[:3], (200, 260), 6, "cosine")
    assert np.all(out3[125:, 155:] == 0)


def test_padding_modes_match_numpy_pad():
    rng = np.random.default_rng(9)
    t = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)
    for mode, npmode in [("mirror", "reflect"), ("reflect", "symmetric"), ("replicate", "edge"), ("constant", "constant")]:
        ref = np.pad(t, ((0, 11), (0, 9), (0, 0)), mode=npmode)
        assert np.array_equal(onp.apply_padding(t, 11, 9, mode), ref)
        full = np.zeros((20, 20, 3), np.uint8)
        full[2:7, 3:10] = t
        assert np.array_equal(oc.tile_extract_pad(full, 3, 2, 7, 5, 16, mode), ref[:16, :16])


def test_resize_and_gray_oracles():
    rng = np.random.default_rng(4)
    a = rng.integers(0, 256, (60, 80, 3), dtype=np.uint8)
    for (dw, dh) in [(8, 6), (32, 24), (160, 120), (80, 60)]:
        assert np.array_equal(onp.resize_cubic_u8(a, dw, dh), oc.resize_cubic_u8(a, dw, dh))
    assert np.array_equal(oc.resize_cubic_u8(a, 80, 60), a)          # identity scale keeps the image
    flat = np.full((20, 30, 3), 93, np.uint8)
    assert np.all(oc.resize_cubic_u8(flat, 77, 51) == 93) and np.all(onp.resize_linear_u8(flat, 77, 51) == 93)
    for shift in (14, 15):
        assert np.array_equal(onp.rgb2gray_u8(a, shift), oc.rgb2gray_u8(a, shift))
    g = onp.rgb2gray_u8(np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0]]], np.uint8))
    assert g.tolist() == [[255, 0, 76]]


def test_merge_tiles_oracle_partition():
    """Feather ramps of neighbouring tiles must reproduce a constant image (no resize case)."""
    metas = [dict(global_x=0, global_y=0, output_w=60, output_h=40, overlap_top=0, overlap_bottom=0, overlap_left=0, overlap_right=20),
             dict(global_x=40, global_y=0, output_w=60, output_h=40, overlap_top=0, overlap_bottom=0, overlap_left=20, overlap_right=0)]
    tiles = [np.full((40, 60, 3), 200, np.uint8)] * 2
    out = onp.merge_tiles(tiles, metas, 100, 40, 1.0, True)
    assert out.shape == (40, 100, 3) and set(np.unique(out)) <= {199, 200}


def test_feather_weights_are_one_and_distance_transform_restatement():
    """feather_blend (blending_module.py:1313-1337) takes cv2.distanceTransform of an all-ones mask: no zero pixel, every
    distance saturates, weight == 1.  The chamfer restatement itself: exact on axis-aligned distances, 1.4 / 2.1969
    metrics on the knight / diagonal moves (16.16 fixed point)."""
    from oracle import oracle_np as onp
    d = onp.distance_transform_l2_5(np.ones((7, 9), np.uint8))
    assert d.min() == d.max() == np.float32(8192.0)
    assert np.all(onp.feather_weight_map(7, 9) == 1.0) and np.all(onp.feather_weight_map(200, 300) == 1.0)
    m = np.ones((9, 9), np.uint8)
    m[4, 4] = 0
    d = onp.distance_transform_l2_5(m)
    assert d[4, 4] == 0 and d[4, 8] == 4.0 and d[0, 4] == 4.0
    assert abs(d[3, 3] - 1.4) < 1e-4 and abs(d[2, 3] - 2.1969) < 1e-4 and abs(d[0, 0] - 4 * 1.4) < 1e-3


def test_color_correction_oracle_properties():
    from oracle import oracle_np as onp
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (40, 48, 3), dtype=np.uint8)
    # matching an image to itself: the table is the identity on every value that occurs
    for c in range(3):
        h = np.bincount(img[..., c].ravel(), minlength=256)
        lut = onp.histogram_lut(h, h)
        assert np.array_equal(lut[img[..., c]], img[..., c])
    assert np.array_equal(onp.color_correction(img, img, "histogram", local_filter=False), img)
    # box mean of a constant is the constant; the 8x8 window is anchored at 4 (x-4 .. x+3)
    assert np.all(onp.box_blur_f32(np.full((9, 11), 3.25, np.float32), 8) == np.float32(3.25))
    imp = np.zeros((20, 20), np.float32)
    imp[10, 10] = 64.0
    b = onp.box_blur_f32(imp, 8)
    assert b[10 - 3, 10 - 3] == 1.0 and b[10 + 4, 10 + 4] == 1.0 and b[10 + 5, 10] == 0.0 and b[10 - 4, 10] == 0.0
    t = onp.mean_std_table(np.bincount(img[..., 0].ravel(), minlength=256), np.bincount(img[..., 0].ravel(), minlength=256))
    assert np.allclose(t, np.arange(256), atol=1e-3)
