#!/usr/bin/env python3
"""Device time of the API-parity QA of one 200 MP pair (QualityAssessmentModule.evaluate_full_reference,
quality_assessment_module.py:467-555): the three-scale bicubic comparison fused (sr_assess_resized_u8) against the
unfused sequence (2 resizes + PSNR + SSIM per scale), plus the full-size PSNR / SSIM / MS-SSIM pass.
usage (on the GPU box): python tools/qa_api_timing.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "super-resolution-system_amd")):
    sys.path.insert(0, p)
import numpy as np            # noqa: E402
import torch                  # noqa: E402
import _native                # noqa: E402
import bench                  # noqa: E402
import device_pipeline as dp  # noqa: E402

geo = dp.workload_geometry("200MP")
H, W, cn = geo.canvas_h, geo.canvas_w, 3
dev = torch.device("cuda", 0)
ctx = _native.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
src = bench.synthetic_source()
src2 = np.clip(src.astype(np.int16) + np.random.default_rng(7).integers(-3, 4, src.shape), 0, 255).astype(np.uint8)
t = torch.from_numpy(np.stack([src, src2])).to(dev)
a = torch.empty((H, W * cn), dtype=torch.uint8, device=dev)
b = torch.empty_like(a)
ctx.resize_cubic_u8(t[0].data_ptr(), src.shape[1] * cn, src.shape[0], src.shape[1], cn, a.data_ptr(), W * cn, H, W)
ctx.resize_cubic_u8(t[1].data_ptr(), src.shape[1] * cn, src.shape[0], src.shape[1], cn, b.data_ptr(), W * cn, H, W)
scales = (0.1, 0.2, 0.4)
res = torch.zeros(4, dtype=torch.float64, device=dev)


def fused():
    out = []
    for s in scales:
        out.append(ctx.assess_resized_u8(a.data_ptr(), W * cn, b.data_ptr(), W * cn, H, W, cn, int(H * s), int(W * s)))
    return out


def unfused():
    out = []
    for s in scales:
        dh, dw = int(H * s), int(W * s)
        ra = torch.empty((dh, dw * cn), dtype=torch.uint8, device=dev)
        rb = torch.empty_like(ra)
        ctx.resize_cubic_u8(a.data_ptr(), W * cn, H, W, cn, ra.data_ptr(), dw * cn, dh, dw)
        ctx.resize_cubic_u8(b.data_ptr(), W * cn, H, W, cn, rb.data_ptr(), dw * cn, dh, dw)
        out.append(ctx.assess_u8(ra.data_ptr(), dw * cn, rb.data_ptr(), dw * cn, dh, dw, cn,
                                 flags=_native.ASSESS_SSE | _native.ASSESS_UNIFORM7))
    return out


def full():
    return ctx.assess_u8(a.data_ptr(), W * cn, b.data_ptr(), W * cn, H, W, cn,
                         flags=_native.ASSESS_SSE | _native.ASSESS_UNIFORM7 | _native.ASSESS_GAUSS11)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps, r


tf, rf = timed(fused)
tu, ru = timed(unfused)
tfull, _ = timed(full)
ctx.prof_enable(True)
ctx.prof_reset()
fused(); unfused(); full()
torch.cuda.synchronize()
kern = {k: round(ms, 4) for k, (ms, n) in ctx.prof_get().items()}
ctx.prof_enable(False)
same = all(x["sse"] == y["sse"] and abs(x["ssim_uniform"] - y["ssim_uniform"]) <= 1e-12 * abs(y["ssim_uniform"]) for x, y in zip(rf, ru))
print(json.dumps({"image": f"{W}x{H}", "scales": scales, "downsample_comparison_fused_ms": round(tf, 3),
                  "downsample_comparison_unfused_ms": round(tu, 3), "full_size_psnr_ssim_msssim_ms": round(tfull, 3),
                  "fused_equals_unfused": bool(same), "kernel_ms_one_call_each": kern,
                  "note": "wall time per call incl. one host sync per scale (the synchronous API the mirror module uses)"}))
