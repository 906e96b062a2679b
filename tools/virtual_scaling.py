#!/usr/bin/env python3
"""Compute-side strong-scaling bound, measured on ONE GPU: every rank of a world of N is rehearsed in turn (its own
buffers, only the rows the exchange plan delivers, the staged blend, its strip's assessment) and timed with HIP
events.  max over ranks = the step time an N-GPU run would have with a free exchange; T(1) / max = the speed-up bound
the partition allows.  Not a measurement of an N-GPU run (no xGMI traffic here).   usage: tools/virtual_scaling.py [workload]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "super-resolution-system_amd")):
    sys.path.insert(0, p)
import numpy as np          # noqa: E402
import torch                # noqa: E402
import device_pipeline as dp  # noqa: E402
import bench                # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "200MP"
STAGED = len(sys.argv) > 2 and sys.argv[2] == "staged"      # default: the steady state of an image stream (monolithic)
geo = dp.workload_geometry(wl)
H, W, cn = geo.canvas_h, geo.canvas_w, geo.cn
dev = torch.device("cuda", 0)
src = bench.synthetic_source()
t_src = torch.from_numpy(src).to(dev)
image = torch.empty((H, W * cn), dtype=torch.uint8, device=dev)
mono = dp.DevicePipeline(geo, 0, 1, 0)
mono.ctx.resize_cubic_u8(t_src.data_ptr(), src.shape[1] * cn, src.shape[0], src.shape[1], cn, image.data_ptr(), W * cn, H, W)
reference = image.clone()


def timed(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


t1 = timed(lambda: mono.step(image, reference))


def stream10():
    mono.pipeline_begin(image)
    for i in range(10):
        mono.pipeline_step(reference, image if i < 9 else None)
    mono.pipeline_finish()


t1_stream = timed(stream10, reps=2) / 10            # one GPU, image stream with the second-stream assessment
full = {t: mono.local_tiles[t] for t in range(len(geo.rects))}
out = {"workload": wl, "blend": "staged" if STAGED else "monolithic", "T1_ms": round(t1, 4),
       "T1_stream_ms": round(t1_stream, 4), "worlds": {}}
for world in (2, 4, 8):
    per_rank, main_chain, qa_only = [], [], []
    for r in range(world):
        p = dp.DevicePipeline(geo, r, world, 0)
        p.rehearse_fill(full)

        def one():
            p.stage_tile(image)
            p.rehearse_step(reference, staged=STAGED)
        per_rank.append(round(timed(one), 4))

        def main_only():
            p.stage_tile(image)
            p.plan.blend(p._ptrs, p._strides, p.canvases[0].data_ptr(), p.canvases[0].stride(0))
        main_chain.append(round(timed(main_only), 4))
        qa_only.append(round(timed(lambda: p.stage_assess(reference)), 4))
        if r == world // 2:                       # per-kernel breakdown of a middle rank
            p.ctx.prof_enable(True)
            p.ctx.prof_reset()
            for _ in range(5):
                one()
            torch.cuda.synchronize()
            kern = {k: round(ms / 5, 4) for k, (ms, _) in p.ctx.prof_get().items()}
            p.ctx.prof_enable(False)
        p.close()
    stream_rank = [max(a, b) for a, b in zip(main_chain, qa_only)]      # assessment on the second stream, perfectly hidden
    out["worlds"][world] = {"per_rank_ms": per_rank, "max_ms": max(per_rank), "bound_speedup": round(t1 / max(per_rank), 2),
                            "main_chain_ms": main_chain, "assess_ms": qa_only,
                            "stream_bound_speedup": round(t1_stream / max(stream_rank), 2),
                            "kernels_mid_rank_ms": kern}
print(json.dumps(out))
