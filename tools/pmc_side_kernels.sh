#!/bin/bash
# Runs ON THE GPU BOX: VALU / LDS / occupancy counters of the round-3 side kernels (fused guided filters, column-march sampler,
# feather merge), one rocprofv3 --pmc pass per counter group, program directly after `--`.  Output: gpurun_out/pmc_side/summary.txt
export TMPDIR=/tmp
OUT=gpurun_out/pmc_side
rm -rf $OUT; mkdir -p $OUT
for PROG in tools/adjust_timing.py tools/sampler_timing.py tools/feather_merge_timing.py; do
  N=$(basename $PROG .py)
  for G in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU" "VALUBusy MemUnitStalled OccupancyPercent MemUnitBusy" "FETCH_SIZE SQ_WAVES SQ_INSTS_VMEM_RD" "WRITE_SIZE SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
    rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/$N -- python3 $PROG > /dev/null 2>&1
  done
done
python3 - <<'PY' > gpurun_out/pmc_side/summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_side/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/pmc_side/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
keep = ("k_cc_fused8", "k_gfx_coeff17", "k_gf_box17_out", "k_resize_gray_pair", "k_feather_merge", "k_cc_map", "k_hist_u8")
print("per-launch means (counters summed over the chip); duration under the counter passes, ms")
for k in sorted(agg):
    if not k.startswith(keep): continue
    d = dur.get(k, [0.0])
    print(f"{k}   launches {len(d)}  mean {sum(d)/len(d):.3f} ms")
    for c, v in sorted(agg[k].items()):
        print(f"    {c:24s} {sum(v)/len(v):18.1f}")
PY
cat gpurun_out/pmc_side/summary.txt
