"""Pure-read / pure-write / copy rates of the device (torch kernels: fill_, sum, copy_) -- what a store-heavy kernel can hope for."""
import json, torch
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
out = {}
for gb in (0.25, 1.0, 2.0):
    n = int(gb * 2**30) // 4
    x = torch.empty(n, dtype=torch.float32, device="cuda"); y = torch.empty_like(x)
    w = t(lambda: x.fill_(1.0)); r = t(lambda: x.sum()); c = t(lambda: y.copy_(x))
    out[f"{gb}GiB"] = {"write_TBps": round(n * 4 / w / 1e9, 3), "read_TBps": round(n * 4 / r / 1e9, 3), "copy_TBps_rw": round(2 * n * 4 / c / 1e9, 3),
                       "ms": [round(w, 4), round(r, 4), round(c, 4)]}
print(json.dumps(out))
