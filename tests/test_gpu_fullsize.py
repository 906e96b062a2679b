"""GPU parity at the FULL size of every BASELINE.json configuration (SURVEY.md 8(d) geometries): the HIP result is
compared with the CPU oracle itself, not through properties.

  cfg1   4 MP  2x2 of 1366x911, ov 273/182 -> 2459x1640   laplacian_fusion AND merge_tiles (its "plumbing" blend)
  cfg2 100 MP  3x3 of 4710x3349                            device pipeline: canvas, PSNR, 3 x SSIM
  cfg4 150 MP  4x4 of 4412x3162
  cfg3 200 MP  5x5 of 4124x2970
  cfg5 200 MP  non-uniform k-d tiling (32 rectangles)

Bar: u8 canvas bit-identical to oracle/sr_oracle.c (reference: blending_module.py:369-506), PSNR equal to the last
bit (exact integer SSE), SSIM within 1e-9 relative (fp64, different summation order).  The OpenCV-defined semantics
of the oracle itself stay 'parity unpinned' (oracle header).  The C oracle needs a few seconds per 100 MP on the
box's 16 host threads.
"""
import numpy as np
import pytest

from oracle import oracle_c as oc
from oracle import oracle_np as onp

pytestmark = pytest.mark.gpu


def _device_image(ctx, H, W, seed, coarse=12):
    """A synthetic H x W x 3 image with structure at every pyramid level, made on the GPU (bicubic upscale of seeded
    noise -- generating 200 MP with NumPy would take longer than the oracle) and returned as a [H, W*3] u8 tensor."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(seed)
    small = torch.randint(0, 256, (H // coarse + 2, (W // coarse + 2) * 3), dtype=torch.uint8, device="cuda", generator=g)
    image = torch.empty((H, W * 3), dtype=torch.uint8, device="cuda")
    ctx.resize_cubic_u8(small.data_ptr(), small.stride(0), small.shape[0], small.shape[1] // 3, 3,
                        image.data_ptr(), W * 3, H, W)
    torch.cuda.synchronize()
    return image


@pytest.mark.parametrize("workload", ["4MP", "100MP", "150MP", "200MP", "200MP-kd"])
def test_baseline_config_equals_oracle(workload):
    """Tile extract -> Laplacian blend -> PSNR / SSIM of bench.py's pipeline object at full size vs the oracle."""
    import torch
    import device_pipeline as dp
    geo = dp.workload_geometry(workload)
    H, W = geo.canvas_h, geo.canvas_w
    pipe = dp.DevicePipeline(geo, 0, 1, 0)
    image = _device_image(pipe.ctx, H, W, seed=11)
    noise = torch.randint(-3, 4, image.shape, dtype=torch.int16, device="cuda",
                          generator=torch.Generator(device="cuda").manual_seed(12))
    reference = torch.clamp(image.to(torch.int16) + noise, 0, 255).to(torch.uint8)
    del noise
    pipe.step(image, reference)
    torch.cuda.synchronize()
    got = pipe.canvas.cpu().numpy().reshape(H, W, 3)
    m = pipe.metrics()
    h_img = image.cpu().numpy().reshape(H, W, 3)
    h_ref = reference.cpu().numpy().reshape(H, W, 3)
    # the tile stage: every extracted tile equals the slice the reference takes (tiling_module.py:713-715)
    for t in (0, len(geo.rects) // 2, len(geo.rects) - 1):
        x, y, w, h = geo.rects[t]
        assert np.array_equal(pipe.local_tiles[t].cpu().numpy().reshape(h, w, 3), h_img[y:y + h, x:x + w]), t
    pipe.close()
    del image, reference
    torch.cuda.empty_cache()

    tiles = [np.ascontiguousarray(h_img[y:y + h, x:x + w]) for (x, y, w, h) in geo.rects]
    want = oc.laplacian_fusion(tiles, [(y, x) for (x, y, _, _) in geo.rects], (H, W), geo.levels, geo.weight_type)
    del tiles
    diff = int(np.count_nonzero(got != want))
    assert diff == 0, f"{workload}: {diff} canvas bytes differ from the oracle"
    assert m["psnr"] == oc.psnr(h_ref, want)
    g0, g1 = oc.rgb2gray_u8(h_ref), oc.rgb2gray_u8(want)
    for mode in ("uniform", "gauss", "simple"):
        assert m[f"ssim_{mode}"] == pytest.approx(oc.ssim(g0, g1, mode), rel=1e-9), mode


def test_config1_blending_module_full_size(ctx):
    """cfg1 through the reference-shaped call surface: create_tile_grid (blending_module.py:1492-1560) ->
    laplacian_fusion / weighted_average_fusion, fp32 canvas bit-exact before quantisation."""
    import blending_module as bm
    tw, th, ovx, ovy = 1366, 911, 273, 182
    H, W = 2 * th - ovy, 2 * tw - ovx                                   # 1640 x 2459
    img = _device_image(ctx, H, W, seed=21, coarse=9).cpu().numpy().reshape(H, W, 3)
    pos = [(r * (th - ovy), c * (tw - ovx)) for r in range(2) for c in range(2)]
    tiles = [np.ascontiguousarray(img[y:y + th, x:x + tw]) + np.uint8(3 * i) for i, (y, x) in enumerate(pos)]
    infos = [bm.TileInfo(image=t, x=x, y=y, row=i // 2, col=i % 2) for i, (t, (y, x)) in enumerate(zip(tiles, pos))]
    b = bm.BlendingModule(method='laplacian', num_levels=6)
    want_u8, want_f = oc.laplacian_fusion(tiles, pos, (H, W), 6, "cosine", return_float=True)
    got = b.laplacian_fusion(infos, output_shape=(H, W))
    assert got.shape == (H, W, 3) and np.array_equal(got, want_u8)
    assert np.array_equal(b.laplacian_fusion(infos), want_u8)          # default output_shape = bounding box
    out_u8, out_f = ctx.fusion_np(tiles, pos, (H, W), 6, "cosine", laplacian=True, return_float=True)
    assert np.array_equal(out_f, want_f) and np.array_equal(out_u8, want_u8)
    assert np.array_equal(b.weighted_average_fusion(infos, output_shape=(H, W)),
                          oc.weighted_average_fusion(tiles, pos, (H, W), "cosine"))


def test_config1_merge_tiles_full_size(ctx, tmp_path):
    """cfg1's own blend is TilingModule.merge_tiles (tiling_module.py:1074-1175, the CPU 'plumbing' path): 2x2 tiles of
    1366x911 feathered into 2459x1640, (a) tile data already at output size, (b) tile data at half size so the
    INTER_LINEAR resize branch (:1104-1109) runs, (c) blending=False.  Bit-exact vs oracle_np.merge_tiles."""
    import tiling_module as tm
    tw, th, ovx, ovy = 1366, 911, 273, 182
    H, W = 2 * th - ovy, 2 * tw - ovx
    img = _device_image(ctx, H, W, seed=22, coarse=9).cpu().numpy().reshape(H, W, 3)
    t = tm.TilingModule(block_size=1366, overlap_ratio=0.2, output_scale=1.0, l2_cache_dir=str(tmp_path))
    tiles, metas = [], []
    for i in range(4):
        r, c = divmod(i, 2)
        x, y = c * (tw - ovx), r * (th - ovy)
        md = tm.TileMetadata(global_x=x, global_y=y, input_w=tw, input_h=th, output_w=tw, output_h=th,
                             overlap_top=ovy if r else 0, overlap_bottom=ovy if r == 0 else 0,
                             overlap_left=ovx if c else 0, overlap_right=ovx if c == 0 else 0)
        tiles.append(tm.Tile(metadata=md, data=np.ascontiguousarray(img[y:y + th, x:x + tw]) + np.uint8(5 * i)))
        metas.append(dict(global_x=x, global_y=y, output_w=tw, output_h=th, overlap_top=md.overlap_top,
                          overlap_bottom=md.overlap_bottom, overlap_left=md.overlap_left, overlap_right=md.overlap_right))
    got = t.merge_tiles(tiles, W, H, blending=True)
    assert got.shape == (H, W, 3)
    assert np.array_equal(got, onp.merge_tiles([x.data for x in tiles], metas, W, H, 1.0, True))
    assert np.array_equal(t.merge_tiles(tiles, W, H, blending=False),
                          onp.merge_tiles([x.data for x in tiles], metas, W, H, 1.0, False))
    half = [np.ascontiguousarray(x.data[::2, ::2]) for x in tiles]       # 456 x 683 -> resized to 911 x 1366
    for x, d in zip(tiles, half):
        x.data = d
    assert np.array_equal(t.merge_tiles(tiles, W, H, blending=True), onp.merge_tiles(half, metas, W, H, 1.0, True))
