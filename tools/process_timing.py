#!/usr/bin/env python3
"""SuperResolutionPipeline.process() end to end at the 200 MP scale (device-resident path): wall time per stage, bytes that
crossed PCIe (one upload of the source, one download of the canvas) and the stage-5 writers against Pillow's.
usage: tools/process_timing.py [--width 8660 --height 5774] [--ext tif]   (x2 SR scale -> 17320 x 11548 canvas)"""
import argparse
import asyncio
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "super-resolution-system_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=8660)
    ap.add_argument("--height", type=int, default=5774)
    ap.add_argument("--ext", default="tif")
    ap.add_argument("--pillow", action="store_true", help="also time Pillow's writers on the same canvas")
    args = ap.parse_args()
    import main as sr_main
    import _native
    from PIL import Image
    Image.MAX_IMAGE_PIXELS = None
    rng = np.random.default_rng(20260313)
    yy, xx = np.mgrid[0:args.height, 0:args.width].astype(np.float32)
    img = np.stack([128 + 64 * np.sin(xx / 37.0 + 0.7 * c) + 48 * np.cos(yy / 23.0 + 1.3 * c) for c in range(3)], axis=-1)
    img = np.clip(img + rng.integers(-12, 13, img.shape), 0, 255).astype(np.uint8)
    del yy, xx
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        src = os.path.join(d, "in.png")
        _native.write_image(img, src)
        cfg = sr_main.PipelineConfig(block_size=2048, overlap_ratio=0.2, sr_scale=2)
        pipe = sr_main.SuperResolutionPipeline(cfg)
        pipe.tiling_module.l2_cache_dir = type(pipe.tiling_module.l2_cache_dir)(d)
        out = {"source": f"{args.width}x{args.height}", "runs": []}
        for rep in range(2):                                   # first run includes plan / context creation
            dst = os.path.join(d, f"out{rep}.{args.ext}")
            t0 = time.perf_counter()
            res = asyncio.run(pipe.process(src, dst, prompt=""))
            dt = time.perf_counter() - t0
            assert res.success, res.error_message
            out["runs"].append({"seconds": round(dt, 3), "stages_s": {k: round(v, 4) for k, v in pipe.stage_times.items()},
                                "transfers": pipe.transfers, "blocks": res.total_blocks, "bytes_written": os.path.getsize(dst),
                                "psnr": res.quality_report["full_reference"]["psnr"], "score": res.quality_score})
        canvas = np.asarray(Image.open(dst))
        out["canvas"] = f"{canvas.shape[1]}x{canvas.shape[0]} = {canvas.shape[0] * canvas.shape[1] / 1e6:.1f} MP"
        enc = {}
        for ext in ("tif", "png", "jpg"):
            p = os.path.join(d, "w." + ext)
            t0 = time.perf_counter()
            _native.write_image(canvas, p)
            enc[ext] = {"native_s": round(time.perf_counter() - t0, 3), "bytes": os.path.getsize(p), "threads": os.cpu_count()}
            if args.pillow:
                t0 = time.perf_counter()
                im = Image.fromarray(canvas)
                if ext == "tif":
                    im.save(p, format="TIFF", compression="tiff_lzw")
                elif ext == "png":
                    im.save(p, format="PNG", compress_level=3)
                else:
                    im.save(p, quality=95)
                enc[ext]["pillow_s"] = round(time.perf_counter() - t0, 3)
                enc[ext]["pillow_bytes"] = os.path.getsize(p)
        out["writers"] = enc
    print(json.dumps(out))


if __name__ == "__main__":
    main()
