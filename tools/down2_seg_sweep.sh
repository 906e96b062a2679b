#!/bin/bash
# Runs ON THE GPU BOX: the fused level 0 -> 2 march (sr_down2.inc) with fixed segment lengths, one bench run each
# (kernel table only).   usage: tools/down2_seg_sweep.sh "<seg> <seg> ..."   -> gpurun_out/down2_seg_sweep.txt
OUT=gpurun_out/down2_seg_sweep.txt
: > $OUT
LIST=${1:-auto 12 16 21 24 32 42 64}
for SEG in $LIST; do
  if [ "$SEG" = auto ]; then unset SR_DOWN2_SEG; else export SR_DOWN2_SEG=$SEG; fi
  timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pcie --sweep none 2>/dev/null | tail -1 |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('seg $SEG', 'down_l0', round(k['down_l0']['ms_per_step'],4), 'down_l1p', round(k['down_l1p']['ms_per_step'],4), 'step', d['ms_per_step'])" >> $OUT || exit 1
done
cat $OUT
