"""CPU: the host-side mirrors of the reference's modules -- names, defaults, enums, error behaviour
and the pure bookkeeping (no GPU compute is called here)."""
import dataclasses
import os

import numpy as np
import pytest

import blending_module as bm
import main as sr_main
import quality_assessment_module as qa
import tiling_module as tm
from oracle import oracle_np as onp


def test_reference_names_exist():
    for mod, names in [
        (tm, ["PaddingMode", "TileStatus", "CacheLevel", "TileMetadata", "Tile", "TilingModule"]),
        (bm, ["FusionMethod", "PoissonMode", "WeightType", "Seam", "TileInfo", "OverlapRegion", "BlendingModule",
              "create_tile_grid", "ParallelBlender"]),
        (qa, ["AssessmentLevel", "QualityThresholds", "ScaleConfig", "QualityAssessmentModule"]),
        (sr_main, ["PipelineConfig", "PipelineResult", "SuperResolutionPipeline"]),
    ]:
        for n in names:
            assert hasattr(mod, n), (mod.__name__, n)


def test_pipeline_config_defaults_match_reference():
    c = sr_main.PipelineConfig()
    ref = dict(block_size=2048, overlap_ratio=0.2, padding_mode='mirror', target_resolution="100MP",
               seedream_strength=0.5, seedream_steps=50, blend_method='laplacian', num_pyramid_levels=6,
               max_agents=60, max_concurrent=30, enable_qa=True, qa_device='cpu', volc_ak="", volc_sk="",
               volc_region="cn-beijing")
    for k, v in ref.items():
        assert getattr(c, k) == v
    assert [f.name for f in dataclasses.fields(sr_main.PipelineResult)] == [
        "success", "output_path", "processing_time", "total_blocks", "successful_blocks", "failed_blocks",
        "quality_score", "quality_report", "error_message"]


def test_target_size_rule(tmp_path):
    p = sr_main.SuperResolutionPipeline(sr_main.PipelineConfig(block_size=256))
    p.tiling_module.l2_cache_dir  # constructed
    assert p._calculate_target_size((1280, 720), "200MP") == (17320, 9742)
    assert p._calculate_target_size((1080, 720), "200MP") == (17320, 11546)
    assert p._calculate_target_size((1280, 720), "3000x2000") == (3000, 2000)
    assert p._calculate_target_size((1280, 720), "bogus") == (12245, 8163)


def test_tiling_module_init_rules(tmp_path):
    with pytest.raises(ValueError):
        tm.TilingModule(overlap_ratio=0.05, l2_cache_dir=str(tmp_path))
    with pytest.raises(ValueError):
        tm.TilingModule(overlap_ratio=0.35, l2_cache_dir=str(tmp_path))
    with pytest.raises(ValueError):
        tm.TilingModule(padding_mode="wrap", l2_cache_dir=str(tmp_path))
    t = tm.TilingModule(block_size=1024, overlap_ratio=0.2, l2_cache_dir=str(tmp_path / "c"))
    assert (t.overlap_pixels, t.output_size) == (204, 2048) and (tmp_path / "c").is_dir()
    pos = t._calculate_tile_positions(4096, 4096)
    assert len(pos) == 25 and pos[-1] == (3280, 3280, 816, 816)
    assert t._calculate_overlap_for_tile(3280, 3280, 816, 816, 4096, 4096) == (204, 4, 204, 4)


def test_tile_metadata_roundtrip_and_effective_region():
    m = tm.TileMetadata(global_x=10, global_y=20, input_w=100, input_h=80, overlap_top=5, overlap_bottom=6,
                        overlap_left=7, overlap_right=8)
    d = m.to_dict()
    assert d["status"] == "PENDING"
    assert tm.TileMetadata.from_dict(d) == m
    t = tm.Tile(metadata=m)
    assert t.get_overlap_region() == (5, 6, 7, 8)
    assert t.get_effective_region() == (17, 25, 17 + 100 - 15, 25 + 80 - 11)


def test_checkpoint_probe_only(tmp_path):
    """The tile caches / checkpoints are out of scope; what remains is the probe of main.py:299-304."""
    t = tm.TilingModule(l2_cache_dir=str(tmp_path))
    assert t.restore_from_cache("deadbeef") is None
    (tmp_path / "checkpoint_abc.json").write_text('{"num_tiles": 4}')
    assert t.restore_from_cache("abc") == {"num_tiles": 4}
    assert not hasattr(tm, "LRUCache") and not hasattr(t, "save_tile_cache")


def test_blending_module_config_and_grid():
    with pytest.raises(ValueError):
        bm.BlendingModule(method="nope")
    b = bm.BlendingModule(method="weighted", num_levels=4)
    assert b.method is bm.FusionMethod.WEIGHTED_AVERAGE and b.num_levels == 4
    imgs = [np.zeros((48, 64, 3), np.uint8)] * 6
    infos, regions = bm.create_tile_grid(imgs, (2, 3), overlap=10)
    ref = onp.create_tile_grid_positions(6, (2, 3), (48, 64), 10)
    assert [(i.x, i.y, i.row, i.col) for i in infos] == ref
    want = onp.overlap_regions(ref, [(48, 64)] * 6)
    got = [(r.tile1_idx, r.tile2_idx, r.x1_start, r.y1_start, r.x2_start, r.y2_start, r.width, r.height, r.direction)
           for r in regions]
    assert got == want and len(got) == 7
    assert bm.Seam(0, 0, 1, 1, 0.80).severity == "high" and bm.Seam(0, 0, 1, 1, 0.99).suggested_fix == "none"
    # weight map helper is the reference formula
    for wt in bm.WeightType:
        assert np.array_equal(b._create_distance_weight_map(40, 72, wt), onp.distance_weight_map(40, 72, wt.value))
    # reference quirk: bare arrays without output_shape -> max() of an empty sequence
    with pytest.raises(ValueError):
        b.laplacian_fusion([np.zeros((32, 32, 3), np.uint8)])
    with pytest.raises(NotImplementedError):
        b.poisson_fusion(None, None)


def test_qa_module_host_logic():
    q = qa.QualityAssessmentModule()
    assert q.lpips_model_vgg is None
    assert q._assess_psnr(41) == "excellent" and q._assess_psnr(36) == "good" and q._assess_psnr(10) == "poor"
    assert q._assess_ssim(0.97) == "good" and q._assess_lpips(0.2) == "poor"
    assert q._calculate_overall_score({"psnr": 30.0, "ms_ssim": 0.9}) == pytest.approx(60.0)
    assert q._calculate_overall_score({"psnr": 130.0, "ms_ssim": 0.9, "lpips_vgg": 0.5}) == pytest.approx(80.0)
    assert q._calculate_overall_score({}) == 0.0
    with pytest.raises(RuntimeError):
        q.calculate_lpips(None, None)
    with pytest.raises(ValueError):
        q.downsample_bicubic(np.zeros((10, 10, 3), np.uint8), 1.5)
    # preprocess rule: max <= 1.0 -> *255 -> u8 (incl. the all-black quirk)
    assert q._preprocess_image(np.full((2, 2), 0.5)).dtype == np.uint8
    assert q._preprocess_image(np.full((2, 2), 0.5))[0, 0] == 127
    assert q._preprocess_image(np.zeros((2, 2), np.uint8)).dtype == np.uint8
    a, b = q._crop_pair(np.zeros((5, 9, 3)), np.zeros((7, 4, 3)))
    assert a.shape == b.shape == (5, 4, 3)
    with pytest.raises(ValueError):
        qa.QualityAssessmentModule(ssim_branch="C")


def test_mean_std_table_vs_reference_float32_expression():
    """_mean_std_matching (blending_module.py:1062-1086) takes np.mean / np.std of a float32 image: NumPy accumulates those
    in float32 (over axes (0, 1) of an HWC array: one running float32 sum per channel, millions of terms), so its moments
    differ from the exact histogram moments the table is built from by up to ~1e-4 relative, image-size dependent.
    After clip + astype(uint8) truncation a source VALUE whose mapped value lies that close to an integer flips by one
    grey level -- for every pixel of that value.  Documented bound (parity unpinned: NumPy's accumulation order is not
    reproduced): never more than 1 LSB, on the pixels of a few of the 256 source values.  Checked on a large image against
    the literal expression of the reference (no guided filter: that isolates the table)."""
    import blending_module as bm
    rng = np.random.default_rng(11)
    h, w = 1500, 2200
    yy, xx = np.mgrid[0:h, 0:w]
    src = np.clip(120 + 50 * np.sin(xx / 91.0)[..., None] + 30 * np.cos(yy / 57.0)[..., None]
                  + rng.integers(-20, 21, (h, w, 3)), 0, 255).astype(np.uint8)
    ref = np.clip(src.astype(np.int16) * 0.8 + 40 + rng.integers(-9, 10, (h, w, 3)), 0, 255).astype(np.uint8)
    sf, rf = src.astype(np.float32), ref.astype(np.float32)
    want = (sf - np.mean(sf, axis=(0, 1))) * (np.std(rf, axis=(0, 1)) / (np.std(sf, axis=(0, 1)) + 1e-6)) + np.mean(rf, axis=(0, 1))
    want = np.clip(want, 0, 255).astype(np.uint8)
    got = np.empty_like(src)
    for c in range(3):
        tab = bm.BlendingModule._mean_std_table(np.bincount(src[..., c].ravel(), minlength=256),
                                                np.bincount(ref[..., c].ravel(), minlength=256))
        got[..., c] = np.clip(tab, 0, 255).astype(np.uint8)[src[..., c]]
    diff = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert int(diff.max()) <= 1
    values_flipped = {int(v) for c in range(3) for v in np.unique(src[..., c][diff[..., c] != 0])}
    assert len(values_flipped) <= 24 and float((diff != 0).mean()) < 0.10


def test_non_u8_seams_and_color_correction_raise_pinned_types():
    """Inputs the HIP path does not take are refused before any device call, with the exception the reference's own calls
    pin: cv2.cvtColor(BGR2GRAY) (blending_module.py:873-876) rejects colour windows that are not 8-bit / 16-bit / float32
    -> ValueError here (cv2.error is not importable); what cv2 / numpy would accept -> NotImplementedError."""
    import numpy as np
    from blending_module import BlendingModule, TileInfo
    bm = BlendingModule()
    u8 = np.zeros((32, 32, 3), np.uint8)
    for bad in (np.float64, np.int32):
        with pytest.raises(ValueError):
            bm.detect_seams(u8.astype(bad), [TileInfo(u8, 0, 0, 0, 0)])
        with pytest.raises(ValueError):
            bm.detect_seams(u8, [TileInfo(u8.astype(bad), 0, 0, 0, 0)])
    for ok_in_cv2 in (np.float32, np.uint16):
        with pytest.raises(NotImplementedError):
            bm.detect_seams(u8.astype(ok_in_cv2), [TileInfo(u8, 0, 0, 0, 0)])
    with pytest.raises(NotImplementedError):
        bm.detect_seams(np.zeros((32, 32), np.float64), [TileInfo(np.zeros((32, 32), np.uint8), 0, 0, 0, 0)])   # gray: astype only
    with pytest.raises(NotImplementedError):
        bm.color_correction(u8.astype(np.float32), u8)
    with pytest.raises(NotImplementedError):
        bm.color_correction(u8, u8.astype(np.int16), method="mean_std")
    assert bm.color_correction(u8.astype(np.float64), u8, method="none").dtype == np.float64                     # 'none' returns the input


def test_color_table_class_and_the_exactness_it_promises():
    """sr_color_table_class (host only): 1 = whole numbers, 2 = a float table whose box sums of 64 values of g, fl32(g * v),
    fl32(g * g) are exact in fp64 (the fused guided filter then slides them), 0 = ordered sums.  For class-2 tables the
    promise is checked directly: 64 random table values summed in float64 in three different orders (the oracle's
    sequential order, reversed, sorted by magnitude) and with a sliding update give one and the same double."""
    import _native
    rng = np.random.default_rng(5)
    ident = np.tile(np.arange(256, dtype=np.float32), (3, 1))
    assert _native.color_table_class(ident) == 1
    assert _native.color_table_class(ident[:1] * np.float32(0.5)) in (0, 2) and _native.color_table_class(ident * np.float32(0.5)) == 2
    bad = ident.copy()
    bad[1, 7] = np.float32(2.0 ** -20)
    assert _native.color_table_class(bad) == 0
    bad[1, 7] = np.nan
    assert _native.color_table_class(bad) == 0
    bad[1, 7] = np.inf
    assert _native.color_table_class(bad) == 0
    with pytest.raises(ValueError):
        _native.color_table_class(ident, terms=0)
    n2 = 0
    for _ in range(200):
        t = ((np.arange(256, dtype=np.float32) - np.float32(rng.uniform(20, 230))) * np.float32(rng.uniform(0.2, 3.0))
             + np.float32(rng.uniform(0, 255)))[None, :]
        cls = _native.color_table_class(t)
        assert cls in (0, 2)
        if cls != 2:
            continue
        n2 += 1
        v = rng.integers(0, 256, 72)
        g = t[0][v]
        for m in (g, g * v.astype(np.float32), g * g):
            assert m.dtype == np.float32
            x = m.astype(np.float64)
            seq = 0.0
            for e in x[:64]:
                seq += e
            rev = 0.0
            for e in x[:64][::-1]:
                rev += e
            srt = 0.0
            for e in sorted(x[:64], key=abs):
                srt += e
            assert seq == rev == srt
            slide = seq
            for k in range(8):                                   # the window moves on by 8: add the entering, drop the leaving value
                slide += x[64 + k] - x[k]
            want = 0.0
            for e in x[8:72]:
                want += e
            assert slide == want
    assert n2 > 100            # most mean_std tables qualify; those with an entry next to zero do not
