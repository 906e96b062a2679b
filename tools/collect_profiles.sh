#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + HBM traffic counters for bench.py, ONE build.
# Counters go in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), never with other trace domains; the
# program itself follows `--` (python3, no wrapper).  Everything of one collection lives under a fresh run directory whose
# name is written to run_id.txt: gpurun merges gpurun_out/ into the caller's copy, where an older collection of the same
# tag may still sit -- tools/summarize_profiles.py reads only the run named there.
#   usage: tools/collect_profiles.sh <tag> [quick]     -> gpurun_out/profiles_<tag>/<run_id>/
set -u
TAG=${1:-run}
QUICK=${2:-}
RUN=$(date +%Y%m%d_%H%M%S)
BASE=gpurun_out/profiles_$TAG
OUT=$BASE/$RUN
export TMPDIR=/tmp
mkdir -p "$OUT"
PMC_STEPS=3; PMC_WARMUP=1
python3 bench.py --steps 20 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err" || echo "bench failed"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pcie --sweep none \
    > "$OUT/bench_under_trace.json" 2> "$OUT/trace.err" || echo "trace failed"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -- python3 bench.py --steps $PMC_STEPS --warmup $PMC_WARMUP --no-cpu-baseline --no-prof --no-pcie --sweep none \
      > /dev/null 2> "$OUT/pmc_$C.err" || echo "pmc $C failed"
done
if [ -z "$QUICK" ]; then
  for WL in 100MP 150MP 200MP-kd 4MP; do
    python3 bench.py --workload $WL --steps 20 --warmup 3 --sweep none --no-cpu-baseline > "$OUT/bench_$WL.json" 2> "$OUT/bench_$WL.err" || echo "bench $WL failed"
  done
  python3 bench.py --mode batch --workload 150MP --steps 20 --warmup 3 --sweep none --no-cpu-baseline > "$OUT/bench_batch150.json" 2> "$OUT/bench_batch150.err" || echo "bench batch failed"
fi
find "$OUT/trace" -name "*_kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
DIGEST=$(cat super-resolution-system_amd/libsrhip.digest 2>/dev/null || echo unknown)
echo "{\"run_id\": \"$RUN\", \"pmc_steps\": $PMC_STEPS, \"pmc_warmup\": $PMC_WARMUP, \"build_digest\": \"$DIGEST\"}" > "$OUT/meta.json"
echo "$RUN" > "$BASE/run_id.txt"
ls "$OUT" | head -40
