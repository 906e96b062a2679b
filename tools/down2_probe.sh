#!/bin/bash
OUT=gpurun_out/down2_probe.txt
: > $OUT
run() { timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pcie --sweep none 2>/dev/null | tail -1 |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$1', 'down_l0', round(k['down_l0']['ms_per_step'],4), 'down_l1p', round(k['down_l1p']['ms_per_step'],4), 'step', d['ms_per_step'])" >> $OUT; }
for SEG in 6 8 10 12; do SR_DOWN2_SEG=$SEG run "seg $SEG" || exit 1; done
SR_DOWN2_SEG=12 SR_DOWN2_PROBE_NOBORDER=1 run "seg 12 noborder" || exit 1
SR_DOWN2=0 run "down2 off" || exit 1
cat $OUT
