#!/usr/bin/env python3
"""Device time of TilingModule.merge_tiles' feather path (sr_feather_merge, tiling_module.py:1074-1175) on the 200 MP
geometry: 25 tiles of 4124 x 2970 (no resize: out == src), 825 px ramps.  usage (GPU box): python tools/feather_merge_timing.py"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "super-resolution-system_amd")):
    sys.path.insert(0, p)
import torch                  # noqa: E402
import _native                # noqa: E402
import device_pipeline as dp  # noqa: E402

geo = dp.workload_geometry("200MP")
H, W = geo.canvas_h, geo.canvas_w
dev = torch.device("cuda", 0)
ctx = _native.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
n = len(geo.rects)
tiles = [torch.randint(0, 256, (h, w * 3), dtype=torch.uint8, device=dev) for (_, _, w, h) in geo.rects]
canvas = torch.zeros((H, W * 3), dtype=torch.uint8, device=dev)
ov = 825
cols = 5
mt = (_native.MergeTile * n)(*[_native.MergeTile(x, y, w, h, w, h, ov if i // cols > 0 else 0, ov if i // cols < cols - 1 else 0,
                                                  ov if i % cols > 0 else 0, ov if i % cols < cols - 1 else 0)
                               for i, (x, y, w, h) in enumerate(geo.rects)])
ptrs = (C.c_void_p * n)(*[C.c_void_p(t.data_ptr()) for t in tiles])
st = (C.c_int64 * n)(*[t.stride(0) for t in tiles])


def once():
    _native.check(ctx.lib.sr_feather_merge(ctx.handle, mt, n, ptrs, st, 1, C.c_void_p(canvas.data_ptr()), W * 3, H, W))


once(); once()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    once()
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / 10
alg = 3.0 * geo.tile_pixels + 3.0 * geo.canvas_pixels
print(json.dumps({"feather_merge_ms": round(ms, 4), "alg_GB": round(alg / 1e9, 3), "GBps": round(alg / 1e9 / (ms / 1e3), 1),
                  "MP_per_s": round(geo.canvas_pixels / 1e6 / (ms / 1e3), 1)}))
