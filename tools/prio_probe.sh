#!/bin/bash
# Runs ON THE GPU BOX: the image stream with the assessment's HIP stream at different priorities.
OUT=gpurun_out/prio_probe.txt
python3 -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')" > $OUT 2>&1
for P in none 1 0 -1 none 1; do
  if [ "$P" = none ]; then unset SR_QA_STREAM_PRIO; else export SR_QA_STREAM_PRIO=$P; fi
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-pcie --no-prof --sweep none 2>/dev/null | tail -n 1 |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('prio $P', 'step', d['ms_per_step'], 'median', d['step_ms']['median'])" >> $OUT || exit 1
done
cat $OUT
