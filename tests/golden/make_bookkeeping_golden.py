"""Writes tests/golden/bookkeeping.json from the NumPy restatement (oracle/oracle_np.py) of the
reference's integer tile bookkeeping.  The survey's hand-checked known answers are
asserted here before anything is written, so the fixture is pinned by them.
    python tests/golden/make_bookkeeping_golden.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import oracle_np as onp

CASES = [(4096, 4096, 1024, 0.2), (1280, 720, 2048, 0.2), (1280, 720, 512, 0.2), (1920, 1080, 1024, 0.2),
         (17320, 11547, 4096, 0.2), (12245, 8163, 4096, 0.2), (1000, 700, 256, 0.1), (333, 1999, 200, 0.3)]
out = {"tiling": [], "target_size": []}
for (w, h, block, ratio) in CASES:
    ov = int(block * ratio)
    pos = onp.tile_positions(w, h, block, ov)
    ovl = [onp.tile_overlaps(x, y, tw, th, w, h, block, ov) for (x, y, tw, th) in pos]
    nbr = onp.neighbor_graph(pos, block, ov)
    out["tiling"].append({"w": w, "h": h, "block": block, "ratio": ratio, "overlap_px": ov,
                          "positions": pos, "overlaps": ovl,
                          "neighbors": [[-1 if n[k] is None else n[k] for k in ("top", "bottom", "left", "right")] for n in nbr]})
for size in [(1280, 720), (1080, 720), (1920, 1080), (720, 1280)]:
    for preset in ("100MP", "150MP", "200MP"):
        out["target_size"].append({"size": size, "preset": preset, "target": onp.target_size(size, preset)})

# known answers recorded by the survey
t = out["tiling"][0]
assert len(t["positions"]) == 25 and t["positions"][0] == (0, 0, 1024, 1024) and t["overlaps"][0] == (0, 204, 0, 204)
assert t["positions"][-1] == (3280, 3280, 816, 816) and t["overlaps"][-1] == (204, 4, 204, 4)
t = out["tiling"][1]
assert t["positions"] == [(0, 0, 1280, 720)] and t["overlaps"] == [(0, 1328, 0, 768)]
assert len(out["tiling"][2]["positions"]) == 6
t = out["tiling"][3]
assert len(t["positions"]) == 6 and t["positions"][-1] == (1640, 820, 280, 260) and t["overlaps"][-1] == (204, 560, 204, 540)
assert len(out["tiling"][4]["positions"]) == 24 and len(out["tiling"][5]["positions"]) == 12
assert onp.target_size((1280, 720), "200MP") == (17320, 9742) and onp.target_size((1080, 720), "200MP") == (17320, 11546)
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bookkeeping.json")
json.dump(out, open(dst, "w"))
print("wrote", dst)
