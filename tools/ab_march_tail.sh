set -u
for i in 1 2 3; do
  for v in 1 0; do
    SR_MARCH_TAIL=$v SR_MARCH_STATS=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pcie --sweep none > gpurun_out/s2_tail_${v}_$i.json 2> gpurun_out/s2_tail_${v}_$i.err || echo fail
    python - <<P
import json
d=json.loads(open("gpurun_out/s2_tail_${v}_$i.json").read().strip().splitlines()[-1])
print("tail=$v run $i", d["ms_per_step"], d["step_ms"]["median"], d["kernels"]["final_gather"]["ms_per_step"], d["kernels"]["final_gather"]["parts_ms"], d["parity"]["canvas_equal"] if "parity" in d else None)
P
  done
done
grep "\[march\]" gpurun_out/s2_tail_1_1.err | sort | uniq -c
grep "\[march\]" gpurun_out/s2_tail_0_1.err | sort | uniq -c
