"""GPU parity: HIP pyramid / blend path vs the CPU oracle (oracle/sr_oracle.c), through the C ABI.

Bar: the fp32 canvas before quantisation is bit-exact with the oracle (same expression order,
no FMA contraction); the u8 canvas is identical.  The OpenCV-defined semantics themselves are
'parity unpinned' (see oracle header)."""
import numpy as np
import pytest

from oracle import oracle_c as oc
from oracle import oracle_np as onp

pytestmark = pytest.mark.gpu


def _tiles(rng, n, h, w, cn=3, dtype=np.uint8):
    out = []
    for i in range(n):
        yy, xx = np.mgrid[0:h, 0:w]
        base = 128 + 64 * np.sin(xx / 37.0 + i) + 48 * np.cos(yy / 23.0 + 0.5 * i)
        img = base[..., None] + rng.integers(-12, 13, (h, w, cn)) + 7 * i
        img = np.clip(img, 0, 255)
        out.append(img.astype(dtype) if cn > 1 else img[..., 0].astype(dtype))
    return out


@pytest.mark.parametrize("shape", [(37, 53, 3), (64, 64, 3), (130, 96), (2, 2, 3), (9, 2, 3), (301, 517, 3)])
def test_pyr_down_up_dense(ctx, rng, shape):
    a = rng.uniform(0, 255, shape).astype(np.float32)
    d_ref = oc.pyr_down(a)
    d_gpu = ctx.pyr_down_np(a)
    assert d_gpu.shape == d_ref.shape
    assert np.array_equal(d_gpu, d_ref)
    u_ref = oc.pyr_up(d_ref, a.shape[:2])
    assert np.array_equal(ctx.pyr_up_np(d_ref, a.shape[:2]), u_ref)
    assert np.array_equal(ctx.pyr_up_np(d_ref, a.shape[:2], a, "sub"), a - u_ref)
    assert np.array_equal(ctx.pyr_up_np(d_ref, a.shape[:2], a, "add"), u_ref + a)


def test_pyr_up_rejects_bad_size(ctx):
    import _native
    with pytest.raises(_native.SrShapeError):
        ctx.pyr_up_np(np.zeros((4, 4), np.float32), (9, 8))


GRIDS = [
    # (tile_h, tile_w, rows, cols, overlap, levels, weight)
    (96, 130, 2, 2, 30, 6, "cosine"),
    (64, 64, 1, 3, 16, 6, "linear"),
    (97, 61, 3, 2, 20, 4, "sigmoid"),
    (512, 512, 2, 2, 100, 6, "cosine"),   # blending_module.py:1774-1814 demo geometry (924^2 canvas)
    (40, 200, 2, 1, 8, 1, "cosine"),      # levels = 1: R0 = G0 * W0
]


# the marched gather forced (its remainder as rectangles of cells: k_final_rect) / the same with the remainder through the masked
# blocks + edge blocks / with uniform item segments cut for six rounds / the default of a small canvas: the block kernel alone
MARCH_FORMS = [("2", {}), ("2", {"SR_RECT": "0"}), ("2", {"SR_MARCH_TAIL": "0", "SR_MARCH_ROUNDS": "6,6,6", "SR_RECT_CELLS": "64"}), ("1", {})]


@pytest.mark.parametrize("march,env", MARCH_FORMS)
@pytest.mark.parametrize("th,tw,rows,cols,ov,levels,wt", GRIDS)
def test_laplacian_fusion_grid(ctx, rng, th, tw, rows, cols, ov, levels, wt, march, env, monkeypatch):
    monkeypatch.setenv("SR_MARCH", march)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    tiles = _tiles(rng, rows * cols, th, tw)
    pos = [((i // cols) * (th - ov), (i % cols) * (tw - ov)) for i in range(rows * cols)]
    shape = (rows * th - (rows - 1) * ov, cols * tw - (cols - 1) * ov)
    ref_u8, ref_f = oc.laplacian_fusion(tiles, pos, shape, levels, wt, return_float=True)
    out_u8, out_f = ctx.fusion_np(tiles, pos, shape, levels, wt, laplacian=True, return_float=True)
    assert np.array_equal(out_f, ref_f), float(np.nanmax(np.abs(out_f - ref_f)))
    assert np.array_equal(out_u8, ref_u8)


def test_laplacian_fusion_ragged_and_cropped(ctx, rng):
    """Different tile sizes, arbitrary positions, a hole (sum w = 0) and a tile sticking out of the
    canvas (cropped like blending_module.py:477-484)."""
    sizes = [(80, 120), (100, 90), (64, 150), (90, 90)]
    pos = [(0, 0), (10, 100), (70, 20), (75, 160)]
    tiles = [_tiles(rng, 1, h, w)[0] for h, w in sizes]
    shape = (150, 230)
    ref_u8, ref_f = oc.laplacian_fusion(tiles, pos, shape, 6, "cosine", return_float=True)
    out_u8, out_f = ctx.fusion_np(tiles, pos, shape, 6, "cosine", return_float=True)
    assert np.array_equal(out_f, ref_f)
    assert np.array_equal(out_u8, ref_u8)


def test_laplacian_fusion_gray_and_float_tiles(ctx, rng):
    tiles = _tiles(rng, 2, 72, 88, cn=1)
    pos = [(0, 0), (0, 60)]
    ref = oc.laplacian_fusion(tiles, pos, (72, 148), 5, "cosine", return_float=True)
    out = ctx.fusion_np(tiles, pos, (72, 148), 5, "cosine", return_float=True)
    assert np.array_equal(out[1], ref[1]) and np.array_equal(out[0], ref[0])
    ftiles = [t.astype(np.float32) * 0.75 + 3.25 for t in _tiles(rng, 2, 72, 88)]
    ref = oc.laplacian_fusion(ftiles, pos, (72, 148), 6, "cosine", return_float=True)
    out = ctx.fusion_np(ftiles, pos, (72, 148), 6, "cosine", return_float=True)
    assert np.array_equal(out[1], ref[1]) and np.array_equal(out[0], ref[0])


@pytest.mark.parametrize("wt", ["cosine", "linear", "sigmoid"])
def test_weighted_average_fusion(ctx, rng, wt):
    tiles = _tiles(rng, 4, 96, 130)
    pos = [(0, 0), (0, 100), (70, 0), (70, 100)]
    ref = oc.weighted_average_fusion(tiles, pos, (166, 230), wt, return_float=True)
    out = ctx.fusion_np(tiles, pos, (166, 230), 6, wt, laplacian=False, return_float=True)
    assert np.array_equal(out[1], ref[1]) and np.array_equal(out[0], ref[0])


def test_fusion_rejects_tiny_tile(ctx):
    with pytest.raises(ValueError):
        ctx.fusion_np([np.zeros((6, 40, 3), np.uint8)], [(0, 0)], (6, 40), 6, "cosine")


def test_golden_blend_fixture(ctx):
    """Committed fixture (restatement-derived, OpenCV-unverified): see tests/golden/README.md."""
    import os
    p = os.path.join(os.path.dirname(__file__), "golden", "blend_small.npz")
    z = np.load(p)
    tiles = [z[f"tile{i}"] for i in range(int(z["n"]))]
    pos = [tuple(p) for p in z["pos"]]
    out_u8, out_f = ctx.fusion_np(tiles, pos, tuple(z["shape"]), int(z["levels"]), "cosine", return_float=True)
    assert np.array_equal(out_u8, z["canvas_u8"])
    np.testing.assert_allclose(out_f, z["canvas_f32"], rtol=1e-6, atol=1e-5)


def test_strip_windows_bit_identical(ctx, rng):
    """Multi-GPU decomposition on one GPU: each 'virtual rank' blends only its canvas strip from
    tiles whose rows outside sr_blend_plan_tile_rows are poisoned; strips must equal the monolithic
    result bit for bit."""
    import _native
    th, tw, rows, cols, ov = 300, 260, 3, 2, 60
    tiles = _tiles(rng, rows * cols, th, tw)
    rects = [((i % cols) * (tw - ov), (i // cols) * (th - ov), tw, th) for i in range(rows * cols)]
    H, W = rows * th - (rows - 1) * ov, cols * tw - (cols - 1) * ov
    pos = [(r[1], r[0]) for r in rects]
    ref_u8, ref_f = oc.laplacian_fusion(tiles, pos, (H, W), 6, "cosine", return_float=True)
    n_ranks = 5
    bounds = [H * r // n_ranks for r in range(n_ranks + 1)]
    canvas = ctx.alloc(H * W * 3)
    canvas_f = ctx.alloc(H * W * 3 * 4)
    ctx.memset(canvas.ptr, 0, H * W * 3)
    for r in range(n_ranks):
        plan = _native.BlendPlan(ctx, rects, 3, H, W, 6, "cosine", bounds[r], bounds[r + 1])
        bufs, ptrs = [], []
        for t, tile in enumerate(tiles):
            a, b = plan.tile_rows(t)
            poisoned = np.full_like(tile, 0xAA)
            poisoned[a:b] = tile[a:b]
            if a < b:
                assert b - a <= th
            bufs.append(ctx.upload(poisoned))
            ptrs.append(bufs[-1].ptr)
        plan.blend(ptrs, [tw * 3] * len(tiles), canvas.ptr, W * 3, _native.SR_U8, canvas_f.ptr)
        ctx.sync()
        plan.close()
        for b_ in bufs:
            b_.free()
    out_u8 = ctx.download(canvas.ptr, (H, W, 3), np.uint8)
    out_f = ctx.download(canvas_f.ptr, (H, W, 3), np.float32)
    assert np.array_equal(out_f, ref_f)
    assert np.array_equal(out_u8, ref_u8)
    # halo is bounded: a strip in the middle must not need whole tiles
    plan = _native.BlendPlan(ctx, [(0, 0, 2000, 3000)], 3, 3000, 2000, 6, "cosine", 1400, 1600)
    a, b = plan.tile_rows(0)
    below, above = _native.pyramid_halo(6)               # analytic bound of the window planner: 155 / 125
    assert 1400 - below <= a <= 1400 and 1600 <= b <= 1600 + above
    plan.close()


def test_staged_blend_equals_monolithic(ctx, rng):
    """sr_blend_pyramids on two disjoint tile subsets + sr_blend_gather == sr_laplacian_blend (the overlap of
    the multi-GPU row exchange with compute relies on this)."""
    import _native
    sizes = [(120, 160), (120, 160), (100, 90), (140, 200), (120, 160)]
    rects = [(0, 0, 160, 120), (120, 10, 160, 120), (30, 100, 90, 100), (100, 90, 200, 140), (250, 40, 160, 120)]
    tiles = [_tiles(rng, 1, h, w)[0] for (h, w) in sizes]
    H, W = 240, 420
    bufs = [ctx.upload(t) for t in tiles]
    ptrs, strides = [b.ptr for b in bufs], [t.shape[1] * 3 for t in tiles]
    plan = _native.BlendPlan(ctx, rects, 3, H, W, 6, "cosine")
    c1, c2 = ctx.alloc(H * W * 3), ctx.alloc(H * W * 3)
    plan.blend(ptrs, strides, c1.ptr, W * 3)
    mono = ctx.download(c1.ptr, (H, W, 3), np.uint8)
    ref = oc.laplacian_fusion(tiles, [(r[1], r[0]) for r in rects], (H, W), 6, "cosine")
    assert np.array_equal(mono, ref)
    plan2 = _native.BlendPlan(ctx, rects, 3, H, W, 6, "cosine")
    plan2.pyramids(ptrs, strides, [3, 0], first=True)
    plan2.pyramids(ptrs, strides, [1, 2, 4], first=False)
    plan2.gather(ptrs, strides, c2.ptr, W * 3)
    assert np.array_equal(ctx.download(c2.ptr, (H, W, 3), np.uint8), mono)
    # empty subset is allowed (a rank that owns none of the tiles it needs)
    plan2.pyramids(ptrs, strides, [], first=True)
    plan2.pyramids(ptrs, strides, [0, 1, 2, 3, 4], first=False)
    plan2.gather(ptrs, strides, c2.ptr, W * 3)
    assert np.array_equal(ctx.download(c2.ptr, (H, W, 3), np.uint8), mono)
    plan.close(); plan2.close()


def test_plan_reuse_with_changing_tile_buffers(ctx, rng):
    """One plan, several calls whose tile pointers / strides / subsets change between calls: the descriptor tables are
    uploaded only when they change (cached host shadows), so every change must be picked up -- each result equals the
    oracle's for the tiles actually passed.  Also the tile extract with changing rectangles on one context."""
    import _native
    rects = [(0, 0, 160, 120), (120, 10, 160, 120), (60, 90, 160, 120)]
    H, W = 210, 280
    plan = _native.BlendPlan(ctx, rects, 3, H, W, 5, "cosine")
    canvas = ctx.alloc(H * W * 3)
    pos = [(r[1], r[0]) for r in rects]
    sets = []
    for k in range(3):
        tiles = _tiles(rng, 3, 120, 160)
        if k == 2:                                        # same pixels as set 1, but padded rows (other strides)
            tiles = [t.copy() for t in sets[1][0]]
            padded = [np.zeros((120, 160 + 7, 3), np.uint8) for _ in tiles]
            for p_, t in zip(padded, tiles):
                p_[:, :160] = t
            bufs = [ctx.upload(p_) for p_ in padded]
            strides = [(160 + 7) * 3] * 3
        else:
            bufs = [ctx.upload(t) for t in tiles]
            strides = [160 * 3] * 3
        sets.append((tiles, bufs, strides))
    for order in ([0, 1, 0, 2, 1], ):
        for k in order:
            tiles, bufs, strides = sets[k]
            ptrs = [b.ptr for b in bufs]
            if k == 1:                                    # staged form with alternating subsets
                plan.pyramids(ptrs, strides, [2, 0], first=True)
                plan.pyramids(ptrs, strides, [1], first=False)
                plan.gather(ptrs, strides, canvas.ptr, W * 3)
            else:
                plan.blend(ptrs, strides, canvas.ptr, W * 3)
            got = ctx.download(canvas.ptr, (H, W, 3), np.uint8)
            assert np.array_equal(got, oc.laplacian_fusion(tiles, pos, (H, W), 5, "cosine")), k
    plan.close()
    img = rng.integers(0, 256, (90, 140, 3), dtype=np.uint8)
    d_img = ctx.upload(img)
    for xywh in ([(0, 0, 50, 40), (30, 20, 64, 48)], [(10, 5, 50, 40), (30, 20, 64, 48)], [(10, 5, 50, 40)]):
        outs = [ctx.alloc(w * h * 3) for (_, _, w, h) in xywh]
        ctx.tile_extract(d_img.ptr, 90, 140, 3, 140 * 3, xywh, [o.ptr for o in outs], [w * 3 for (_, _, w, _) in xywh])
        for (x, y, w, h), o in zip(xywh, outs):
            assert np.array_equal(ctx.download(o.ptr, (h, w, 3), np.uint8), img[y:y + h, x:x + w])


# Shapes chosen against k_down2_march (csrc/sr_down2.inc): 1, 2, 3 and many column groups; widths that leave 0 .. 7 border
# columns on the right; level-1 heights odd and even (the two closing formulas of level 2); heights around the segment length;
# row strides of every alignment (the dword shift of a row's window changes from row to row when 3 w is not a multiple of 4).
DOWN2_SHAPES = [(33, 18), (40, 26), (41, 27), (50, 34), (51, 35), (64, 42), (97, 43), (98, 50), (99, 51), (131, 66), (200, 203),
                (201, 404), (67, 805), (405, 90), (406, 91), (407, 92), (408, 93)]


@pytest.mark.parametrize("h,w", DOWN2_SHAPES)
def test_down2_march_shapes(ctx, rng, h, w, monkeypatch):
    """Levels 1 and 2 from one march (default) against the oracle AND against the two launches of round 3 (SR_DOWN2=0):
    fp32 canvas bit-exact, u8 canvas identical; two overlapping tiles so that every level of both is used."""
    tiles = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8), rng.integers(0, 256, (h, w, 3), dtype=np.uint8)]
    tiles[1][: h // 2] = 255                       # saturated and zero areas: the 16-bit row and column sums at their maxima
    tiles[0][h // 2:, : w // 3] = 0
    ov = max(w // 3, 4)
    pos = [(0, 0), (3, w - ov)]
    shape = (h + 3, 2 * w - ov)
    ref_u8, ref_f = oc.laplacian_fusion(tiles, pos, shape, 6, "cosine", return_float=True)
    out_u8, out_f = ctx.fusion_np(tiles, pos, shape, 6, "cosine", laplacian=True, return_float=True)
    assert np.array_equal(out_f, ref_f), float(np.nanmax(np.abs(out_f - ref_f)))
    assert np.array_equal(out_u8, ref_u8)
    monkeypatch.setenv("SR_DOWN2", "0")            # read when the plan is made
    old_u8, old_f = ctx.fusion_np(tiles, pos, shape, 6, "cosine", laplacian=True, return_float=True)
    assert np.array_equal(old_f, out_f) and np.array_equal(old_u8, out_u8)
