#!/usr/bin/env python3
"""Device time of the resized assessment (sr_assess_resized_u8: sampler `resize_gray` + one-channel march `assess_resized`) per
scale on a 200 MP pair -- the multi-scale comparison of QualityAssessmentModule._evaluate_downsample_comparison
(quality_assessment_module.py:518-555) uses 0.1 / 0.2 / 0.4.  SR_RESIZE_MARCH=0 selects the block-staged sampler.
usage (GPU box): python tools/sampler_timing.py"""
import json
import os
import sys

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
g = torch.Generator(device=dev).manual_seed(1)
a = torch.randint(0, 256, (H, W * 3), dtype=torch.uint8, device=dev, generator=g)
b = torch.randint(0, 256, (H, W * 3), dtype=torch.uint8, device=dev, generator=g)
out = {"image": f"{W}x{H}", "march": os.environ.get("SR_RESIZE_MARCH", "1") != "0", "kernel_ms_by_scale": {}}
for s in (0.1, 0.2, 0.4, 0.7):
    dh, dw = int(H * s), int(W * s)
    ctx.assess_resized_u8(a.data_ptr(), W * 3, b.data_ptr(), W * 3, H, W, 3, dh, dw)
    ctx.prof_enable(True)
    ctx.prof_reset()
    for _ in range(3):
        ctx.assess_resized_u8(a.data_ptr(), W * 3, b.data_ptr(), W * 3, H, W, 3, dh, dw)
    torch.cuda.synchronize()
    p = ctx.prof_get()
    ctx.prof_enable(False)
    out["kernel_ms_by_scale"][str(s)] = {k: round(ms / n, 4) for k, (ms, n) in p.items()}
print(json.dumps(out))
