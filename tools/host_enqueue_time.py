import sys, time, numpy as np, torch
sys.path.insert(0,'super-resolution-system_amd'); sys.path.insert(0,'.')
import device_pipeline as dp
geo = dp.workload_geometry("200MP")
H, W = geo.canvas_h, geo.canvas_w
pipe = dp.DevicePipeline(geo, 0, 1, 0)
img = torch.randint(0, 255, (H, W*3), dtype=torch.uint8, device="cuda")
ref = torch.randint(0, 255, (H, W*3), dtype=torch.uint8, device="cuda")
for _ in range(3): pipe.step(img, ref)
torch.cuda.synchronize()
for name, fn in [("step", lambda: pipe.step(img, ref)), ("tile", lambda: pipe.stage_tile(img)), ("blend", lambda: pipe.stage_blend()),
                 ("staged", lambda: (pipe.plan.pyramids(pipe._ptrs, pipe._strides, pipe._local_needed[:10], first=True), pipe.plan.pyramids(pipe._ptrs, pipe._strides, pipe._local_needed[10:], first=False), pipe.plan.gather(pipe._ptrs, pipe._strides, pipe.canvas.data_ptr(), pipe.canvas.stride(0)))),
                 ("assess", lambda: pipe.stage_assess(ref))]:
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(20): fn()
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print(f"{name:8s} host enqueue {1e3*(t1-t0)/20:.3f} ms/iter   total {1e3*(t2-t0)/20:.3f} ms/iter")
