"""Worker of tests/test_main_sharded.py::test_laplacian_fusion_across_ranks: one rank of an SPMD caller of
BlendingModule.laplacian_fusion (every rank holds the tile list; gloo process group; rank 0 saves the canvas)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "super-resolution-system_amd"))


def tiles_of(seed=11):
    rng = np.random.default_rng(seed)
    H, W, th, tw, ov = 300, 420, 180, 250, 80
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.clip((128 + 64 * np.sin(xx / 37.0) + 48 * np.cos(yy / 23.0))[..., None] + rng.integers(-12, 13, (H, W, 3)), 0, 255).astype(np.uint8)
    from blending_module import TileInfo
    pos = [(0, 0), (W - tw, 0), (0, H - th), (W - tw, H - th)]
    return [TileInfo(np.ascontiguousarray(img[y:y + th, x:x + tw]), x, y, i // 2, i % 2) for i, (x, y) in enumerate(pos)], (H, W)


def main():
    out = sys.argv[1]
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=world)
    from blending_module import BlendingModule
    tiles, shape = tiles_of()
    canvas = BlendingModule().laplacian_fusion(tiles, None, output_shape=shape)
    if int(os.environ.get("RANK", "0")) == 0:
        np.save(out, canvas)
    if world > 1:
        dist.destroy_process_group()
    del torch
    return 0


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--launch":       # the test's entry: start the ranks, each runs main()
        import _launch
        sys.exit(_launch.launch_ranks(int(sys.argv[3]), os.path.abspath(__file__), [sys.argv[1]], 600.0, who="fusion_worker"))
    sys.exit(main())
