#!/bin/bash
# Runs ON THE GPU BOX: times every library under variants/ (tools/build_variants.sh) with one short bench run each, on the same
# box.   usage: tools/run_variants.sh [extra env assignments ...]   -> gpurun_out/variants.txt
OUT=gpurun_out/variants.txt
: > $OUT
PK=super-resolution-system_amd
cp $PK/libsrhip.so /tmp/libsrhip.keep; cp $PK/libsrhip.digest /tmp/libsrhip.digest.keep
for d in variants/*/; do
  name=$(basename $d)
  cp $d/libsrhip.so $PK/libsrhip.so; cp $d/libsrhip.digest $PK/libsrhip.digest
  env SR_HIPCC_EXTRA="$(cat $d/flags)" "$@" timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pcie --sweep none 2>/dev/null | tail -1 |
    python3 tools/bench_line.py $name >> $OUT || { echo "$name failed" >> $OUT; break; }
done
cp /tmp/libsrhip.keep $PK/libsrhip.so; cp /tmp/libsrhip.digest.keep $PK/libsrhip.digest
cat $OUT
