#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): derived utilisation counters for every kernel of a short bench run, one
# rocprofv3 pass per counter group (PMC passes carry --kernel-trace only).   usage: tools/pmc_diag.sh <tag>
set -u
TAG=${1:-diag}
OUT=gpurun_out/pmc_$TAG
export TMPDIR=/tmp
rm -rf "$OUT"; mkdir -p "$OUT"
i=0
for GROUP in "VALUBusy MemUnitStalled" "OccupancyPercent VALUUtilization" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d "$OUT/g$i" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-pcie --sweep none \
      > /dev/null 2> "$OUT/g$i.err" || echo "group $i failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k in sorted(agg):
        fo.write(k + "\n")
        for c, v in sorted(agg[k].items()):
            fo.write(f"    {c:24s} mean {sum(v)/len(v):16.3f}  n={len(v)}\n")
print(open(out + "/summary.txt").read()[-6000:])
PY
