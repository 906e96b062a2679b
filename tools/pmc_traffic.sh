#!/bin/bash
# Runs ON THE GPU BOX: HBM-side bytes per kernel of a short bench run (FETCH_SIZE and WRITE_SIZE in separate passes: they do not
# fit one pass on gfx950).   usage: tools/pmc_traffic.sh <tag> [bench args]
set -u
TAG=${1:-traffic}; shift
OUT=gpurun_out/pmc_$TAG
export TMPDIR=/tmp
rm -rf "$OUT"; mkdir -p "$OUT"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/$C" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-pcie --sweep none "$@" \
      > /dev/null 2> "$OUT/$C.err" || echo "$C failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["Grid_Size"]) if "Grid_Size" in r else 0))
with open(out + "/summary.txt", "w") as fo:
    for k in sorted(agg):
        for c, v in sorted(agg[k].items()):
            vals = [x[0] for x in v]
            fo.write(f"{k[:60]:60s} {c:12s} max {max(vals)/1e6:12.3f} M(units)  mean {sum(vals)/len(vals)/1e6:12.3f}  n={len(v)}\n")
print(open(out + "/summary.txt").read())
PY
