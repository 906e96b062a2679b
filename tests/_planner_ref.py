"""Independent NumPy restatement of the strip planner (csrc/sr_host.cpp sr_strip_bounds / sr_exchange_plan) -- test
infrastructure only: tests/test_planner.py checks the C planner against it."""
import bisect

import numpy as np

import _native

# Relative cost of the stages per pixel, from the r01_d per-kernel times at 200 MP on MI355X (ps per pixel): what a
# canvas row costs its strip owner.  Only the ratios matter.
COST_ASSESS = 10.5     # per canvas pixel (fused PSNR + 3 x SSIM; includes its tail effect on strips)
COST_GATHER = 4.0      # per tile pixel visited by the canvas gather
COST_PYRAMID = 6.6     # per tile pixel of the pyramid chains (extract, down, up) -- also paid for the halo rows


def strip_bounds(canvas_h: int, world: int, geo=None):
    """Strip boundaries.  Without a geometry: equal row counts.  With one: equal *work* -- a strip [a, b) costs the
    assessment and gather work of its own rows plus the pyramid work of rows a - halo .. b + halo (clipped to the
    canvas: the outer strips recompute a halo on one side only), where a row's tile work is the tile pixels covering
    it (rows inside tile overlaps count twice).  Found by bisection on the per-strip cost; boundaries are even (the
    kernels pair rows) and deterministic on every rank."""
    if world <= 1 or geo is None:
        return [canvas_h * r // world for r in range(world + 1)]
    cover = np.zeros(canvas_h, dtype=np.float64)
    for (_, y, w, h) in geo.rects:
        cover[max(y, 0):min(y + h, canvas_h)] += w
    cum_can = np.concatenate([[0.0], np.cumsum(COST_ASSESS * geo.canvas_w + COST_GATHER * cover)])
    cum_pyr = np.concatenate([[0.0], np.cumsum(COST_PYRAMID * cover)])

    halo = max(_native.pyramid_halo(geo.levels))

    def cost(a: int, b: int) -> float:
        lo, hi = max(a - halo, 0), min(b + halo, canvas_h)
        return float(cum_can[b] - cum_can[a] + cum_pyr[hi] - cum_pyr[lo])

    def place(target: float):
        bounds = [0]
        for _ in range(world - 1):
            a = bounds[-1]
            lo, hi = a + 2, canvas_h              # smallest even b with cost(a, b) >= target
            while lo < hi:
                mid = (lo + hi) // 2
                if cost(a, mid) >= target:
                    hi = mid
                else:
                    lo = mid + 1
            b = min(lo + (lo % 2), canvas_h)
            bounds.append(max(b, min(a + 2, canvas_h)))
        bounds.append(canvas_h)
        return bounds

    t_lo, t_hi = 0.0, cost(0, canvas_h)
    for _ in range(60):                           # bisection on the per-strip cost: the last strip absorbs the rest
        t = 0.5 * (t_lo + t_hi)
        bnd = place(t)
        if cost(bnd[-2], canvas_h) > t:
            t_lo = t
        else:
            t_hi = t
    bounds = place(t_hi)
    for r in range(1, world + 1):                 # monotone, inside the canvas
        bounds[r] = min(max(bounds[r], bounds[r - 1]), canvas_h)
    return bounds


def tile_owners(rects, bounds, policy: str = "balanced",
                need=None, cn=3):
    """Which rank holds each (SR output) tile.

    "balanced" (default): greedy, tile by tile, choose the owner that minimises the busiest rank-to-rank link
    after the assignment, then the bytes added, then the owner's tile count.  xGMI is point-to-point (7 links per
    GPU), so what bounds the exchange is the heaviest pair, not the total: with 2 ranks this puts every tile on
    the strip that needs most of it; with 8 it spreads a tile row over the strips that read it so each strip
    pulls from several peers in parallel.  Needs ``need`` (rows every rank reads of every tile).
    "roundrobin": tile t on rank t % world (independent SR workers).
    "locality": the rank whose strip holds the tile's centre row."""
    world = len(bounds) - 1
    if policy == "roundrobin" or (policy == "balanced" and need is None):
        return [t % world for t in range(len(rects))]
    if policy == "locality":
        out = []
        for (_, y, _, h) in rects:
            c = min(y + h // 2, bounds[-1] - 1)
            out.append(min(max(bisect.bisect_right(bounds, c) - 1, 0), world - 1))
        return out
    link = np.zeros((world, world), dtype=np.int64)          # bytes owner -> reader
    owned = [0] * world
    out = []
    for t, (_, _, w, _) in enumerate(rects):
        nbytes = [max(need[r][t][1] - need[r][t][0], 0) * w * cn for r in range(world)]
        best = None
        for o in range(world):
            add = [0 if r == o else nbytes[r] for r in range(world)]
            worst = max(int(max(link[o, r] + add[r] for r in range(world))), int(link.max()))
            key = (worst, sum(add), owned[o], o)
            if best is None or key < best[0]:
                best = (key, o, add)
        _, o, add = best
        for r in range(world):
            link[o, r] += add[r]
        owned[o] += 1
        out.append(o)
    return out



def make_exchange_plan(geo, world, halo, owner_policy="balanced"):
    bounds = strip_bounds(geo.canvas_h, world, geo)
    rows, need = [], []
    for r in range(world):
        a = max(bounds[r] - (halo if world > 1 else 0), 0)
        b = min(bounds[r + 1] + (halo if world > 1 else 0), geo.canvas_h)
        rows.append((a, b))
        need.append(_native.strip_tile_rows(geo.rects, geo.levels, geo.canvas_h, a, b))
    return bounds, rows, need, tile_owners(geo.rects, bounds, owner_policy, need, geo.cn)
