#!/bin/bash
# On the GPU box: A/B of library variants (tools/build_variants.py -> feature-extractor_amd/lib/variants/<name>.so) on a stream of device blocks
# (tools/device_blocks.py), interleaved so that clock drift cancels.  The variant is selected by path (FX_LIBRARY_OVERRIDE): the shipped library
# is never touched.   tools/ab_blocks.sh "x1 x2" "8192 1024 480 64" ["1024 1024 480 64" ...]      ("shipped" names the shipped library)
set -u
L=$(pwd)/feature-extractor_amd/lib
names=$1; shift
for round in 1 2 3; do
  for v in $names; do
    for shape in "$@"; do
      echo -n "$v: "
      if [ $v = shipped ]; then timeout -k 10 120 python3 tools/device_blocks.py $shape 2>&1 | tail -1
      else FX_LIBRARY_OVERRIDE=$L/variants/$v.so timeout -k 10 120 python3 tools/device_blocks.py $shape 2>&1 | tail -1; fi
    done
  done
done
