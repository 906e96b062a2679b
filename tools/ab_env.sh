#!/bin/bash
# Runs ON THE GPU BOX: alternated bench runs of the shipped library under environment settings (A/B on one box).
#   usage: tools/ab_env.sh "NAME=VALUE ..." "NAME2=VALUE2 ..." [rounds]   ("-" = no setting)  -> gpurun_out/ab_env.txt
OUT=gpurun_out/ab_env.txt
: > $OUT
R=${3:-2}
for i in $(seq 1 $R); do
  for SET in "$1" "$2"; do
    if [ "$SET" = "-" ]; then E=""; else E="$SET"; fi
    env $E timeout -k 10 200 python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-pcie --sweep none 2>/dev/null | tail -n 1 | python3 tools/bench_line.py "[$SET]" >> $OUT || exit 1
  done
done
cat $OUT
