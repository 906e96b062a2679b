#!/usr/bin/env python3
"""Per-step timeline of the image stream from a rocprofv3 --kernel-trace run of bench.py (tools/collect_profiles.sh):
which kernels of the two HIP streams overlap, how long each phase of a step takes.
usage: tools/stream_timeline.py <kernel_trace.csv> [first_step] [steps]   -> text on stdout"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
first, nsteps = (int(sys.argv[2]) if len(sys.argv) > 2 else 7), (int(sys.argv[3]) if len(sys.argv) > 3 else 3)
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Queue_Id"])
            for r in rows)
ext = [e for e in ev if e[2].startswith("k_tile_extract")]
a, b = ext[first][0], ext[first + nsteps][0]
queues = sorted({e[3] for e in ev if a <= e[0] < b})
print(f"steps {first}..{first + nsteps - 1} of the timed stream; times in microseconds from the first tile extract; queues {queues}")
print(f"{'start':>9} {'end':>9} {'dur':>8}  queue  kernel")
for s, e, n, q in ev:
    if a <= s < b:
        print(f"{(s - a) / 1e3:9.1f} {(e - a) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{queues.index(q)}     {n[:60]}")
step_us = (b - a) / 1e3 / nsteps
print(f"\nstep = {step_us:.1f} us")
for k in range(nsteps):
    s0 = ext[first + k][0]
    s1 = ext[first + k + 1][0]
    inwin = [e for e in ev if s0 <= e[0] < s1]
    gather = [e for e in inwin if e[2].startswith("k_final")]
    assess = [e for e in inwin if e[2].startswith("k_assess_march")]
    chain = [e for e in inwin if e[3] == ext[first][3] and not e[2].startswith("k_final")]
    if gather and assess and chain:
        print(f"step {first + k}: tile + pyramids beside the previous image's assessment {(max(e[1] for e in chain) - s0) / 1e3:7.1f} us "
              f"(assessment {(assess[0][1] - assess[0][0]) / 1e3:7.1f} us), gather alone {(gather[0][1] - gather[0][0]) / 1e3:7.1f} us")
