import sys, numpy as np
sys.path.insert(0, 'super-resolution-system_amd'); sys.path.insert(0, '.')
import _native
from oracle import oracle_c as oc
ctx = _native.default_context(0)
for (h, w, va, vb) in [(7, 7, 0, 0), (7, 7, 1, 1), (7, 7, 100, 100), (7, 7, 255, 255), (8, 7, 100, 100), (7, 8, 100, 100), (9, 9, 100, 50), (7,7,3,0)]:
    a, b = np.full((h, w), va, np.uint8), np.full((h, w), vb, np.uint8)
    da, db = ctx.upload(a), ctx.upload(b)
    r = ctx.assess_u8(da.ptr, w, db.ptr, w, h, w, 1, flags=2)
    print(h, w, va, vb, r["ssim_uniform"], oc.ssim(a, b, "uniform") * _native.ssim_count(h, w, "uniform"))
