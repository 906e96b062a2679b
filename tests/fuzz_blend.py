#!/usr/bin/env python3
"""Randomised parity sweep (GPU box): random tile arrangements, sizes, levels, weight types, channel counts, dtypes and
row windows through the HIP blend against the CPU oracle, bit for bit.  Also the fused assessment on random shapes.
Test infrastructure (it drives the oracle); tests/test_gpu_fuzz.py runs a short sweep.
usage: python tests/fuzz_blend.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "super-resolution-system_amd")):
    sys.path.insert(0, p)
import numpy as np            # noqa: E402
import _native                # noqa: E402
from oracle import oracle_c as oc   # noqa: E402  (checker only)



def run(cases: int = 200, seed: int = 1) -> int:
    rng = np.random.default_rng(seed)
    ctx = _native.default_context(0)
    bad = 0
    compared = rejected = strips = 0
    for it in range(cases):
        n = int(rng.integers(1, 7))
        cn = int(rng.choice([1, 3]))
        levels = int(rng.integers(1, 7))
        wt = str(rng.choice(["cosine", "linear", "sigmoid"]))
        f32 = bool(rng.integers(0, 4) == 0)
        tiles, pos = [], []
        for _ in range(n):
            h, w = int(rng.integers(9, 520)), int(rng.integers(9, 660))
            if rng.integers(0, 25) == 0:
                h, w = int(rng.integers(2, 9)), int(rng.integers(2, 12))       # tiny tiles: few levels
            shape = (h, w, 3) if cn == 3 else (h, w)
            base = rng.integers(0, 256, shape)
            t = base.astype(np.float32) * np.float32(0.93) + np.float32(1.7) if f32 else base.astype(np.uint8)
            tiles.append(t)
            pos.append((int(rng.integers(0, 400)), int(rng.integers(0, 520))))
        H = max(p[0] + t.shape[0] for p, t in zip(pos, tiles)) - int(rng.integers(0, 7))
        W = max(p[1] + t.shape[1] for p, t in zip(pos, tiles)) - int(rng.integers(0, 7))
        H, W = max(H, 2), max(W, 2)
        lap = bool(rng.integers(0, 5) != 0)
        try:
            if lap:
                want = oc.laplacian_fusion(tiles, pos, (H, W), levels, wt)
            else:
                want = oc.weighted_average_fusion(tiles, pos, (H, W), wt)
        except Exception as exc:                      # the oracle rejects what the reference rejects (e.g. 1-px tiles)
            want, oerr = None, exc
        try:
            got = ctx.fusion_np(tiles, pos, (H, W), levels, wt, laplacian=lap)
        except Exception as exc:
            got, gerr = None, exc
        if want is None or got is None:
            if (want is None) != (got is None):
                bad += 1
                print("MISMATCH in error behaviour", it, n, cn, levels, wt, f32, lap, [t.shape for t in tiles], pos, (H, W))
            rejected += 1
            continue
        compared += 1
        if not np.array_equal(got, want):
            bad += 1
            d = np.argwhere(got != want)
            print("MISMATCH", it, dict(n=n, cn=cn, levels=levels, wt=wt, f32=f32, lap=lap, shapes=[t.shape for t in tiles], pos=pos,
                                       canvas=(H, W)), "first diffs", d[:3].tolist(), "count", len(d))
        # the same canvas from two or three row-window plans (strip owners), u8 tiles
        if lap and not f32 and H >= 8 and rng.integers(0, 2) == 0:
            strips += 1
            rects = [(p[1], p[0], t.shape[1], t.shape[0]) for p, t in zip(pos, tiles)]
            bufs = [ctx.upload(np.ascontiguousarray(t)) for t in tiles]
            canvas = ctx.alloc(H * W * cn)
            ctx.memset(canvas.ptr, 0, H * W * cn)
            cuts = sorted(set([0, H] + [int(v) // 2 * 2 for v in rng.integers(1, H, int(rng.integers(1, 3)))]))
            for a_, b_ in zip(cuts[:-1], cuts[1:]):
                if b_ <= a_:
                    continue
                plan = _native.BlendPlan(ctx, rects, cn, H, W, levels, wt, a_, b_)
                plan.blend([bf.ptr for bf in bufs], [t.shape[1] * cn for t in tiles], canvas.ptr, W * cn)
                ctx.sync()
                plan.close()
            got2 = ctx.download(canvas.ptr, (H, W, 3) if cn == 3 else (H, W), np.uint8)
            if not np.array_equal(got2, want):
                bad += 1
                print("STRIP MISMATCH", it, dict(n=n, cn=cn, levels=levels, wt=wt, shapes=[t.shape for t in tiles], pos=pos,
                                                 canvas=(H, W), cuts=cuts))
            for bf in bufs:
                bf.free()
            canvas.free()
    # fused assessment on random shapes
    for it in range(max(cases // 4, 10)):
        h, w = int(rng.integers(7, 400)), int(rng.integers(7, 700))
        cn = int(rng.choice([1, 3]))
        shape = (h, w, 3) if cn == 3 else (h, w)
        a = rng.integers(0, 256, shape, dtype=np.uint8)
        b = np.clip(a.astype(np.int16) + rng.integers(-20, 21, shape), 0, 255).astype(np.uint8)
        da, db = ctx.upload(a), ctx.upload(b)
        r = ctx.assess_u8(da.ptr, w * cn, db.ptr, w * cn, h, w, cn)
        g0 = oc.rgb2gray_u8(a) if cn == 3 else a
        g1 = oc.rgb2gray_u8(b) if cn == 3 else b
        ok = r["sse"] == float(np.sum((a.astype(np.int64) - b.astype(np.int64)) ** 2))
        for mode in ("uniform", "gauss", "simple"):
            cnt = _native.ssim_count(h, w, mode)
            if cnt:
                ok = ok and abs(r[f"ssim_{mode}"] / cnt - oc.ssim(g0, g1, mode)) <= 1e-9
        if not ok:
            bad += 1
            print("ASSESS MISMATCH", it, shape)
        da.free(); db.free()
    print(f"fuzz: {cases} blend cases ({compared} compared, {rejected} rejected by both, {strips} also as strips) + "
          f"{max(cases // 4, 10)} assessment cases, {bad} mismatches")
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
