#!/usr/bin/env python3
"""Device time of the colour-correction kernels (blending_module.color_correction: 256-bin histograms, LUT apply, guided
filter) and the complexity-score moments on one 200 MP image / on the 25 tiles of the 200 MP workload.
usage (GPU box): python tools/adjust_timing.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "super-resolution-system_amd")):
    sys.path.insert(0, p)
import numpy as np            # noqa: E402
import torch                  # noqa: E402
import _native                # noqa: E402
import device_pipeline as dp  # noqa: E402

geo = dp.workload_geometry("200MP")
H, W, cn = geo.canvas_h, geo.canvas_w, 3
dev = torch.device("cuda", 0)
ctx = _native.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
g = torch.Generator(device=dev).manual_seed(3)
img = torch.randint(0, 256, (H, W * cn), dtype=torch.uint8, device=dev, generator=g)
out = torch.empty_like(img)
lut = np.tile(np.arange(256, dtype=np.float32), (cn, 1))
tw, th = geo.rects[0][2], geo.rects[0][3]
tiles = torch.randint(0, 256, (len(geo.rects), th, tw * cn), dtype=torch.uint8, device=dev, generator=g)


# a `mean_std` table (blending_module._mean_std_table): float values, not integers -- class 2 of sr_color_table_class: fp64 sliding sums (k_cc_fused8f)
lut_f = np.clip((lut - 120.0) * np.float32(1.07) + np.float32(131.3), 0, 255).astype(np.float32)
EPS = 0.01                    # the reference's eps (blending_module.py:1092-1146)


def run(table):
    ctx.histogram_u8(img.data_ptr(), W * cn, H, W, cn)
    ctx.color_correct_u8(img.data_ptr(), W * cn, H, W, cn, table, False, 8, EPS, out.data_ptr(), W * cn)
    ctx.color_correct_u8(img.data_ptr(), W * cn, H, W, cn, table, 1, 8, EPS, out.data_ptr(), W * cn)      # _simple_guided_filter
    ctx.color_correct_u8(img.data_ptr(), W * cn, H, W, cn, table, 2, 8, EPS, out.data_ptr(), W * cn)      # the ximgproc branch
    ctx.gray_std_u8(tiles.data_ptr(), len(geo.rects), th * tw * cn, tw * cn, th, tw)


def timed(table):
    run(table)
    torch.cuda.synchronize()
    ctx.prof_enable(True)
    ctx.prof_reset()
    e0, e1, e2, e3 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
    e0.record()
    ctx.color_correct_u8(img.data_ptr(), W * cn, H, W, cn, table, 1, 8, EPS, out.data_ptr(), W * cn)
    e1.record()
    ctx.color_correct_u8(img.data_ptr(), W * cn, H, W, cn, table, 2, 8, EPS, out.data_ptr(), W * cn)
    e2.record()
    run(table)
    torch.cuda.synchronize()
    k = {k: round(ms, 3) for k, (ms, n) in ctx.prof_get().items()}
    ctx.prof_enable(False)
    return {"kernel_ms_two_calls_each": k, "call_ms": {"simple": round(e0.elapsed_time(e1), 3), "ximgproc": round(e1.elapsed_time(e2), 3)}}


res_int, res_float = timed(lut), timed(lut_f)
print(json.dumps({"image": f"{W}x{H}", "tiles": f"{len(geo.rects)} x {tw}x{th}", "eps": EPS,
                  "integer_table": res_int, "float_table": res_float,
                  "table_class": {"integer_table": _native.color_table_class(lut), "float_table": _native.color_table_class(lut_f)},
                  "fused_float": os.environ.get("SR_CC_FUSED_F", "1") != "0",
                  "bytes_GB": {"histogram": round(H * W * cn / 1e9, 3), "lut_apply": round(2 * H * W * cn / 1e9, 3)}}))
