#!/bin/bash
# On the GPU box: rebuild with each set of extra hipcc flags and print tools/pmc_quick.py's line for the pair kernel at 4096 and 2048 points.
# Usage: tools/pair_variants.sh "<flags 1>" "<flags 2>" ...   ("" = the shipped build); SHAPES="4096:1024:64 2048:4096:64" by default
SHAPES=${SHAPES:-"4096:1024:64 2048:4096:64"}
for v in "$@"; do
  FX_EXTRA_HIPCC_FLAGS="$v" python3 feature-extractor_amd/build.py > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  for s in $SHAPES; do
    IFS=: read n c t <<< "$s"
    FX_WAVES_PER_FRAME=2 python3 tools/pmc_quick.py $n $c $t "[${v:-shipped}]" 2>&1 | tail -1
  done
done
