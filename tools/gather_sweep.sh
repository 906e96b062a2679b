# One box, alternated runs: planner knobs of the marched gather (segments per list, rectangle size, side streams).
#   usage (GPU box): bash tools/gather_sweep.sh [reps]      -> gpurun_out/gather_sweep.txt
set -u
REPS=${1:-2}
OUT=gpurun_out/gather_sweep.txt
: > $OUT
run() {   # name, env assignments...
  local name=$1; shift
  env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pcie --sweep none > gpurun_out/gs_tmp.json 2> gpurun_out/gs_tmp.err || echo "$name failed" >> $OUT
  python - "$name" >> $OUT <<'P'
import json, sys
try:
    d = json.loads(open("gpurun_out/gs_tmp.json").read().strip().splitlines()[-1])
    fg = d["kernels"]["final_gather"]
    print(f"{sys.argv[1]:34s} step {d['ms_per_step']:.4f} median {d['step_ms']['median']:.4f} gather {fg['ms_per_step']:.4f} parts {fg.get('parts_ms')}")
except Exception as e:
    print(sys.argv[1], "PARSE ERROR", e)
P
}
for r in $(seq $REPS); do
  run base_wide         SR_DUMMY=1
  run narrow32          SR_RECT_WIDE=0
done
sort -s -k1,1 $OUT
