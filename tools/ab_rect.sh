# A/B on one box: the remainder of the marched gather as rectangles (default) against the masked blocks + edge blocks (SR_RECT=0)
set -u
for i in 1 2 3; do
  for v in 1 0; do
    SR_RECT=$v SR_MARCH_STATS=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pcie --sweep none > gpurun_out/s2_rect_${v}_$i.json 2> gpurun_out/s2_rect_${v}_$i.err || echo fail
    python - <<P
import json
d=json.loads(open("gpurun_out/s2_rect_${v}_$i.json").read().strip().splitlines()[-1])
fg=d["kernels"]["final_gather"]
print("rect=$v run $i", d["ms_per_step"], d["step_ms"]["median"], fg["ms_per_step"], fg.get("parts_ms"), d.get("parity",{}).get("canvas_equal"))
P
  done
done
grep "\[march\]" gpurun_out/s2_rect_1_1.err | sort | uniq -c
