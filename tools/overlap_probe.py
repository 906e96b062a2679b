#!/usr/bin/env python3
"""Probe: does running the (VALU-bound) assessment of image i on a second stream beside the (bandwidth-bound) tile stage
and pyramids of image i+1 raise the single-GPU throughput?  Two contexts on two HIP streams, canvases and result words
double-buffered, events for the two dependencies.  usage (GPU box): python tools/overlap_probe.py"""
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
PRIO = int(os.environ.get("PROBE_PRIO", "0"))
s1 = torch.cuda.Stream(dev, priority=-1) if PRIO else torch.cuda.Stream(dev)
MASK = os.environ.get("PROBE_QA_MASK")               # e.g. 77777777: the assessment stream may use 3 of every 4 CUs
if MASK:
    import ctypes
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    raw = ctypes.c_void_p()
    words = (ctypes.c_uint32 * 8)(*([int(MASK, 16)] * 8))
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(raw), 8, words)
    assert rc == 0, rc
    s2 = torch.cuda.ExternalStream(raw.value)
else:
    s2 = torch.cuda.Stream(dev)
with torch.cuda.stream(s1):
    pipe = dp.DevicePipeline(geo, 0, 1, 0)          # blend context on s1
qa = _native.Context(0, stream=s2.cuda_stream)       # assessment context on s2
src = bench.synthetic_source()
t = torch.from_numpy(src).to(dev)
image = torch.empty((H, W * cn), dtype=torch.uint8, device=dev)
pipe.ctx.resize_cubic_u8(t.data_ptr(), src.shape[1] * cn, src.shape[0], src.shape[1], cn, image.data_ptr(), W * cn, H, W)
torch.cuda.synchronize()
reference = image.clone()
canv = [torch.zeros((H, W * cn), dtype=torch.uint8, device=dev) for _ in range(2)]
res = [torch.zeros(4, dtype=torch.float64, device=dev) for _ in range(2)]
ptrs, strides = pipe.sets[0]["ptrs"], pipe._strides


def run(n, overlap):
    done = [None, None]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        k = i & 1
        with torch.cuda.stream(s1):
            pipe.stage_tile(image)
            if done[k] is not None:
                s1.wait_event(done[k])                     # canvas k is free again
            pipe.plan.blend(ptrs, strides, canv[k].data_ptr(), canv[k].stride(0))
            e = torch.cuda.Event()
            e.record(s1)
        sq = s2 if overlap else s1
        ctx = qa if overlap else pipe.ctx
        with torch.cuda.stream(sq):
            sq.wait_event(e)
            ctx.assess_u8_async(reference.data_ptr(), reference.stride(0), canv[k].data_ptr(), canv[k].stride(0), H, W, cn,
                                res[k].data_ptr())
            d = torch.cuda.Event()
            d.record(sq)
            done[k] = d
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def run_b(n):
    """Variant B: the tile stage of image i+1 also moves to the second stream (behind the assessment of image i-1),
    into a second set of tile buffers; the main stream keeps pyramids + gather only."""
    u8 = dict(dtype=torch.uint8, device=dev)
    sets = [pipe.sets[0]["local"], {t: torch.empty_like(v) for t, v in pipe.sets[0]["local"].items()}]
    ptr_sets = [[sets[j][t].data_ptr() for t in range(len(geo.rects))] for j in (0, 1)]
    owned = list(range(len(geo.rects)))

    def extract(ctx, j):
        ctx.tile_extract(image.data_ptr(), H, W, cn, image.stride(0), [geo.rects[t] for t in owned],
                         [sets[j][t].data_ptr() for t in owned], [sets[j][t].stride(0) for t in owned])

    done = [None, None]          # assessment of canvas slot finished
    tiles_ready = [None, None]   # tile set j extracted
    blend_done = [None, None]    # blend that read tile set j / wrote canvas j finished
    with torch.cuda.stream(s2):
        extract(qa, 0)
        tiles_ready[0] = torch.cuda.Event(); tiles_ready[0].record(s2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        k = i & 1
        with torch.cuda.stream(s1):
            s1.wait_event(tiles_ready[k])
            if done[k] is not None:
                s1.wait_event(done[k])
            pipe.plan.blend(ptr_sets[k], strides, canv[k].data_ptr(), canv[k].stride(0))
            e = torch.cuda.Event(); e.record(s1)
            blend_done[k] = e
        with torch.cuda.stream(s2):
            if i + 1 < n:                                  # tile stage of the next image, other buffer set
                if blend_done[1 - k] is not None:
                    s2.wait_event(blend_done[1 - k])
                extract(qa, 1 - k)
                tiles_ready[1 - k] = torch.cuda.Event(); tiles_ready[1 - k].record(s2)
            s2.wait_event(e)
            qa.assess_u8_async(reference.data_ptr(), reference.stride(0), canv[k].data_ptr(), canv[k].stride(0), H, W, cn,
                               res[k].data_ptr())
            d = torch.cuda.Event(); d.record(s2)
            done[k] = d
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def run_c(n):
    """Variant C: two blend lanes (own stream, context, plan, arena and tile buffers each) alternate images, so the
    gather of image i can run beside the tile stage and pyramids of image i+1; assessment on the third stream."""
    s3 = torch.cuda.Stream(dev)
    with torch.cuda.stream(s3):
        pipe2 = dp.DevicePipeline(geo, 0, 1, 0)
    lanes = [(s1, pipe), (s3, pipe2)]
    done = [None, None]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        k = i & 1
        st, pp = lanes[k]
        with torch.cuda.stream(st):
            pp.stage_tile(image)
            if done[k] is not None:
                st.wait_event(done[k])
            pp.plan.blend(pp.sets[0]["ptrs"], pp._strides, canv[k].data_ptr(), canv[k].stride(0))
            e = torch.cuda.Event(); e.record(st)
        with torch.cuda.stream(s2):
            s2.wait_event(e)
            qa.assess_u8_async(reference.data_ptr(), reference.stride(0), canv[k].data_ptr(), canv[k].stride(0), H, W, cn,
                               res[k].data_ptr())
            d = torch.cuda.Event(); d.record(s2)
            done[k] = d
    torch.cuda.synchronize()
    dt = 1e3 * (time.perf_counter() - t0) / n
    return dt


run(3, False); run(3, True); run_b(3)
varb = run_b(20)
run_c(4)
varc = run_c(20)
seq = run(20, False)
ovl = run(20, True)
print(json.dumps({"sequential_ms_per_image": round(seq, 4), "two_stream_ms_per_image": round(ovl, 4),
                  "two_stream_tiles_on_second_ms_per_image": round(varb, 4),
                  "two_blend_lanes_plus_assessment_stream_ms_per_image": round(varc, 4),
                  "results_equal": bool(torch.equal(res[0], res[1]))}))
