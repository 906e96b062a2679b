#!/usr/bin/env python3
"""Turns gpurun_out/profiles_<tag>/ (tools/collect_profiles.sh) into the committed summaries:
    profiles/<round>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary
    profiles/<round>_traffic.json       per-kernel HBM bytes per launch from FETCH_SIZE / WRITE_SIZE
    profiles/<round>_bench.json         the bench line of the same build
FETCH_SIZE on gfx950 under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM): the read side is
calibrated on k_tile_extract, a pure copy whose byte count is known exactly, and that factor is applied to all kernels.
usage: tools/summarize_profiles.py <tag> <round-prefix>      e.g.  r01b  r01_b
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, prefix = sys.argv[1], sys.argv[2]
src = f"gpurun_out/profiles_{tag}"
os.makedirs("profiles", exist_ok=True)
shutil.copy(f"{src}/kernel_stats.csv", f"profiles/{prefix}_kernel_stats.csv")
bench = json.load(open(f"{src}/bench.json"))
json.dump(bench, open(f"profiles/{prefix}_bench.json", "w"), indent=1)


def counter(name):
    """-> kernel -> counter total per bench step (sum over the launches of the run / number of steps run)."""
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{src}/pmc_{name}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    steps = max(len(agg.get("k_tile_extract", [])), 1)            # one extract launch per step
    return {k: sum(v) / steps for k, v in agg.items()}


fetch, write = counter("FETCH_SIZE"), counter("WRITE_SIZE")
tile_px, canvas_px = bench["config"]["tile_pixels"], bench["config"]["canvas_pixels"]
known_read = 3.0 * tile_px                      # k_tile_extract reads exactly the tile pixels once
cal = known_read / (fetch["k_tile_extract"] * 1024.0) if fetch.get("k_tile_extract") else 2.0
out = {"unit": "bytes per bench step (all launches of the kernel)", "fetch_calibration_factor": cal,
       "note": "read = FETCH_SIZE*1024*factor with the factor calibrated on k_tile_extract (a pure copy of known size), "
               "read_2x = FETCH_SIZE*1024*2 (the guide's correction for 16-B-per-lane streams; an upper bound for the "
               "12-byte and byte-aligned loads used here); write = WRITE_SIZE*1024; total uses the calibrated read, "
               "total_2x the guide's",
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("k_"):
        continue
    raw = fetch.get(k, 0.0) * 1024.0
    rd, wr = raw * cal, write.get(k, 0.0) * 1024.0
    out["kernels"][k] = {"read": rd, "read_2x": raw * 2.0, "write": wr, "total": rd + wr, "total_2x": raw * 2.0 + wr}
json.dump(out, open(f"profiles/{prefix}_traffic.json", "w"), indent=1)
# the bench line of this collection was printed before this summary existed: point its roofline at these counters
roof = bench.get("roofline") or {}
if roof.get("kernel"):
    sys.path.insert(0, ".")
    import bench as bench_mod
    names = bench_mod.ROCPROF_NAMES.get(roof["kernel"], [])
    names = names if isinstance(names, list) else [names]
    vals = [out["kernels"][n]["total"] for n in names if n in out["kernels"]]
    if vals:
        lps = bench["kernels"][roof["kernel"]]["launches_per_step"]
        roof["traffic"], roof["traffic_source"] = sum(vals) / max(lps, 1), f"{prefix}_traffic.json"
        json.dump(bench, open(f"profiles/{prefix}_bench.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
