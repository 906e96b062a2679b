"""GPU: non-uint8 inputs where the reference takes them (round-2 verdict item 8).

* TilingModule.merge_tiles casts whatever dtype the tiles carry (tiling_module.py:1104-1135): float32 / float64 / int16
  tile data at output size, float32 data through the INTER_LINEAR resize branch -- vs oracle_np.merge_tiles (the float
  resize arithmetic is a restatement of cv2's float path: parity unpinned).
* QualityAssessmentModule.calculate_ssim hands non-u8 images on as they are (quality_assessment_module.py:169-195,
  351-417): float images whose maximum exceeds 1 -- 2-D float32 / float64 pairs straight to the float64 SSIM, float32 RGB
  through cv2's float RGB2GRAY -- vs oracle_np.ssim, 1e-9 relative.
"""
import numpy as np
import pytest

from oracle import oracle_np as onp

pytestmark = pytest.mark.gpu


def _img(rng, h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    base = (128 + 64 * np.sin(xx / 37.0) + 48 * np.cos(yy / 23.0))[..., None]
    return np.clip(base + rng.integers(-12, 13, (h, w, 3)), 0, 255).astype(np.uint8)


def test_merge_tiles_float_tile_data(rng, tmp_path):
    import tiling_module as tm
    tw, th, ovx, ovy = 300, 220, 60, 44
    H, W = 2 * th - ovy, 2 * tw - ovx
    img = _img(rng, H, W)
    t = tm.TilingModule(block_size=300, overlap_ratio=0.2, output_scale=1.0, l2_cache_dir=str(tmp_path))
    tiles, metas = [], []
    for i in range(4):
        r, c = divmod(i, 2)
        x, y = c * (tw - ovx), r * (th - ovy)
        md = tm.TileMetadata(global_x=x, global_y=y, input_w=tw, input_h=th, output_w=tw, output_h=th,
                             overlap_top=ovy if r else 0, overlap_bottom=ovy if r == 0 else 0,
                             overlap_left=ovx if c else 0, overlap_right=ovx if c == 0 else 0)
        tiles.append(tm.Tile(metadata=md, data=None))
        metas.append(dict(global_x=x, global_y=y, output_w=tw, output_h=th, overlap_top=md.overlap_top,
                          overlap_bottom=md.overlap_bottom, overlap_left=md.overlap_left, overlap_right=md.overlap_right))
    u8 = [np.ascontiguousarray(img[m["global_y"]:m["global_y"] + th, m["global_x"]:m["global_x"] + tw]) for m in metas]

    def run(datas, blending=True):
        for tl, d in zip(tiles, datas):
            tl.data = d
        return t.merge_tiles(tiles, W, H, blending=blending), onp.merge_tiles(datas, metas, W, H, 1.0, blending)

    # float32 with fractional values and values beyond 255 (astype(uint8) wraps, no clip)
    f32 = [d.astype(np.float32) * np.float32(1.07) + np.float32(0.37 * i) for i, d in enumerate(u8)]
    got, want = run(f32)
    assert got.dtype == np.uint8 and np.array_equal(got, want)
    got, want = run(f32, blending=False)
    assert np.array_equal(got, want)
    # float64 and int16 at output size: the reference's astype(float32)
    got, want = run([d.astype(np.float64) * 0.93 + 1.25 for d in u8])
    assert np.array_equal(got, want)
    got, want = run([d.astype(np.int16) + 40 for d in u8])
    assert np.array_equal(got, want)
    # float32 data at half size: the float INTER_LINEAR branch
    half = [np.ascontiguousarray(d[::2, ::2]) for d in f32]
    got, want = run(half)
    assert np.array_equal(got, want)
    # a float64 tile that needs resizing would run cv2's double-precision path: refused, not approximated
    with pytest.raises(NotImplementedError):
        run([d.astype(np.float64) for d in half])


@pytest.mark.parametrize("branch,multiscale,mode", [("A", True, "gauss"), ("A", False, "uniform"), ("B", True, "simple")])
def test_calculate_ssim_float_images(rng, branch, multiscale, mode):
    """Float images that stay float after _preprocess_image (quality_assessment_module.py:169-195,351-417).
    PARITY UNPINNED for float32 pairs: the 1e-9 here is against this repository's float64 oracle (= scikit-image 0.18.3's
    promotion); scikit-image >= 0.19 computes float32 inputs in float32 and differs by about 1e-6 (include/sr_hip.h)."""
    import quality_assessment_module as qam
    q = qam.QualityAssessmentModule(device='cpu', ssim_branch=branch)
    a8, b8 = _img(rng, 97, 131), _img(rng, 97, 131)
    # 2-D float32 and float64 (max > 1: no rescale): straight to the float64 SSIM
    fa, fb = a8[..., 0].astype(np.float32) * np.float32(0.9) + np.float32(3.3), b8[..., 0].astype(np.float32) * np.float32(1.02)
    assert q.calculate_ssim(fa, fb, multiscale=multiscale) == pytest.approx(onp.ssim(fa, fb, mode), rel=1e-9)
    da, db = a8[..., 1].astype(np.float64) * 0.77 + 11.125, b8[..., 1].astype(np.float64) * 0.81 + 9.5
    assert q.calculate_ssim(da, db, multiscale=multiscale) == pytest.approx(onp.ssim(da, db, mode), rel=1e-9)
    assert q.calculate_ssim(da, db, multiscale=multiscale, data_range=200.0) == \
        pytest.approx(onp.ssim(da, db, mode, data_range=200.0), rel=1e-9)
    # float32 RGB: cv2's float RGB2GRAY first; shapes differ -> common top-left rectangle
    ra, rb = a8.astype(np.float32) * np.float32(0.98) + np.float32(1.5), b8[:90, :120].astype(np.float32)
    want = onp.ssim(onp.rgb2gray_f32(ra[:90, :120]), onp.rgb2gray_f32(rb), mode)
    assert q.calculate_ssim(ra, rb, multiscale=multiscale) == pytest.approx(want, rel=1e-9)
    # a float image in [0, 1] is still rescaled to u8 first (the reference's preprocess rule), and u8 results are unchanged
    assert q.calculate_ssim(a8 / 255.0, b8, multiscale=multiscale) == \
        pytest.approx(q.calculate_ssim((a8 / 255.0 * 255).astype(np.uint8), b8, multiscale=multiscale), rel=1e-12)
    # float64 RGB: cv2.cvtColor has no such conversion
    with pytest.raises(ValueError):
        q.calculate_ssim(a8.astype(np.float64) * 1.5, b8.astype(np.float64), multiscale=multiscale)


def test_calculate_ssim_mixed_u8_float_rgb_is_refused(rng):
    """cv2.cvtColor grays a uint8 image with the rounded fixed-point formula and a float32 one with the float formula
    (quality_assessment_module.py:359-360); sr_ssim_float has one formula for both planes, so the mixed RGB pair is refused
    instead of being answered up to half a grey level per pixel differently."""
    from quality_assessment_module import QualityAssessmentModule
    qa = QualityAssessmentModule()
    a = rng.integers(0, 256, (40, 52, 3), dtype=np.uint8)
    b = a.astype(np.float32) + 1.5                     # max > 1: stays float after _preprocess_image
    with pytest.raises(NotImplementedError):
        qa.calculate_ssim(a, b)
    with pytest.raises(NotImplementedError):
        qa.calculate_ssim(b, a, multiscale=False)
    assert 0.0 < qa.calculate_ssim(b, b + 1.0) <= 1.0  # one dtype: answered
