#!/bin/bash
# Builds libsrhip.so once per SR_HIPCC_EXTRA variant HERE (no GPU needed) into variants/<name>/, so that one gpurun call can time
# them all on the same box:   tools/build_variants.sh name1 "flags1" name2 "flags2" ...     (the last build left in place is the
# default one: the script ends by rebuilding without extra flags)
cd "$(dirname "$0")/.." || exit 1
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  SR_HIPCC_EXTRA="$flags" python3 super-resolution-system_amd/_build.py > /dev/null 2>&1 || { echo "build $name failed"; exit 1; }
  mkdir -p variants/$name
  cp super-resolution-system_amd/libsrhip.so super-resolution-system_amd/libsrhip.digest variants/$name/
  echo "$flags" > variants/$name/flags
  echo "built $name ($flags)"
done
python3 super-resolution-system_amd/_build.py > /dev/null 2>&1
