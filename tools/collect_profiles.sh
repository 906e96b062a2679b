#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + HBM traffic counters for bench.py.
# Counters go in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), never with other trace domains.
#   usage: tools/collect_profiles.sh <tag>      -> gpurun_out/profiles_<tag>/
set -u
TAG=${1:-run}
OUT=gpurun_out/profiles_$TAG
export TMPDIR=/tmp
rm -rf "$OUT"; mkdir -p "$OUT"
python3 bench.py --steps 20 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err" || echo "bench failed"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pcie --sweep none \
    > "$OUT/bench_under_trace.json" 2> "$OUT/trace.err" || echo "trace failed"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof --no-pcie --sweep none \
      > /dev/null 2> "$OUT/pmc_$C.err" || echo "pmc $C failed"
done
for WL in 100MP 150MP 200MP-kd 4MP; do
  python3 bench.py --workload $WL --steps 20 --warmup 3 --sweep none --no-cpu-baseline > "$OUT/bench_$WL.json" 2> "$OUT/bench_$WL.err" || echo "bench $WL failed"
done
python3 bench.py --mode batch --workload 150MP --steps 20 --warmup 3 --sweep none --no-cpu-baseline > "$OUT/bench_batch150.json" 2> "$OUT/bench_batch150.err" || echo "bench batch failed"
find "$OUT" -name "*_kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
ls -R "$OUT" | head -40
