#!/usr/bin/env python3
"""LPIPS throughput on the GPU: time of sr_lpips_u8 on a synthetic image pair, per kernel family (HIP events), useful
convolution FLOPs of the untiled forward and the fraction of the fp32 matrix peak (157.3 TFLOP/s: v_mfma_f32_32x32x2_f32
runs at the fp32 vector rate) -- the MFMA-side roofline of the path's one dense contraction, reported apart from the
HBM roofline of the blend.   usage: tools/lpips_timing.py [--size HxW | --workload 200MP-kd] [--tile 2048] [--nets vgg,alex]
Synthetic seeded weights (no pretrained weights offline): timing does not depend on their values."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "super-resolution-system_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

F32_MATRIX_PEAK_TFLOPS = 157.3


def synthetic_weights(net, seed=20260313):
    import _native
    return _native.lpips_synthetic_weights(net, seed)


def conv_flops(net, h, w):
    """2 * MACs of every convolution of the untiled forward (one image)."""
    import _native
    strides = {"alex": [(4, 2, 11), (1, 2, 5), (1, 1, 3), (1, 1, 3), (1, 1, 3)]}
    total, mfma = 0.0, 0.0
    pools_before = {"alex": {1: (3, 2), 2: (3, 2)}, "vgg": {2: (2, 2), 4: (2, 2), 7: (2, 2), 10: (2, 2)}}[net]
    for i, (co, ci, k) in enumerate(_native.LPIPS_CONV_SHAPES[net]):
        if i in pools_before:
            pk, ps = pools_before[i]
            h, w = (h - pk) // ps + 1, (w - pk) // ps + 1
        s, p = (strides["alex"][i][0], strides["alex"][i][1]) if net == "alex" else (1, 1)
        h, w = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        f = 2.0 * co * ci * k * k * h * w
        total += f
        if ci != 3:
            mfma += f
    return total, mfma


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="4096x4096")
    ap.add_argument("--workload", default=None, help="take the canvas size of a bench workload (e.g. 200MP-kd)")
    ap.add_argument("--tile", type=int, default=4096)
    ap.add_argument("--nets", default="vgg,alex")
    ap.add_argument("--reps", type=int, default=1)
    args = ap.parse_args()
    import torch
    import _native
    import device_pipeline as dp
    if args.workload:
        g = dp.workload_geometry(args.workload)
        H, W = g.canvas_h, g.canvas_w
    else:
        H, W = (int(v) for v in args.size.lower().split("x"))
    dev = torch.device("cuda", 0)
    ctx = _native.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    gen = torch.Generator(device=dev).manual_seed(1)
    a = torch.randint(0, 256, (H, W * 3), dtype=torch.uint8, device=dev, generator=gen)
    noise = torch.randint(-6, 7, (H, W * 3), dtype=torch.int16, device=dev, generator=gen)
    b = (a.to(torch.int16) + noise).clamp_(0, 255).to(torch.uint8)
    del noise
    out = {"image": f"{W}x{H}", "megapixels": H * W / 1e6, "tile": args.tile, "nets": {}}
    for net in args.nets.split(","):
        model = _native.LpipsModel(ctx, net, synthetic_weights(net))
        # warm-up on the first tile of the real image: allocates the activation buffers (4 x up to 4.7 GB at tile 4096)
        model.layer_sums(a.data_ptr(), W * 3, b.data_ptr(), W * 3, H, W, 3, args.tile, 0, 1)
        ctx.prof_enable(True)
        ctx.prof_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            sums = model.layer_sums(a.data_ptr(), W * 3, b.data_ptr(), W * 3, H, W, 3, args.tile)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.reps
        prof = {k: (ms / args.reps, n // args.reps) for k, (ms, n) in ctx.prof_get().items()}
        ctx.prof_enable(False)
        total, mfma = conv_flops(net, H, W)
        vals = [s / (lh * lw) for s, (lh, lw) in zip(sums, model.layer_sizes(H, W))]
        t_mfma = prof.get("lpips_conv_mfma", (0.0, 0))[0] / 1e3
        out["nets"][net] = {
            "seconds": round(dt, 4), "MP_per_s": round(H * W / 1e6 / dt, 2), "lpips": sum(vals),
            "useful_conv_TFLOP_both_images": round(2 * total / 1e12, 3),
            "useful_TFLOPs_per_s_whole_call": round(2 * total / 1e12 / dt, 2),
            "mfma_kernels": {"seconds": round(t_mfma, 4), "useful_TFLOP": round(2 * mfma / 1e12, 3),
                             "useful_TFLOPs_per_s": None if t_mfma <= 0 else round(2 * mfma / 1e12 / t_mfma, 2),
                             "frac_of_f32_matrix_peak": None if t_mfma <= 0 else round(2 * mfma / 1e12 / t_mfma / F32_MATRIX_PEAK_TFLOPS, 4),
                             "note": "useful FLOPs of the untiled forward / time of the MFMA convolution launches "
                                     "(tile halo recompute and partial 8x32 output tiles count as lost time)"},
            "kernel_ms": {k: round(v[0], 3) for k, v in prof.items()},
            "launches": {k: v[1] for k, v in prof.items()},
            "tiles": model.tile_count(H, W, args.tile),
        }
        model.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
