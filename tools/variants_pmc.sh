#!/bin/bash
# On the GPU box: rebuild per flag set, print tools/pmc_quick.py lines for the shapes in SHAPES ("N:C:T ...").
SHAPES=${SHAPES:-"1024:1024:512"}
for v in "$@"; do
  FX_EXTRA_HIPCC_FLAGS="$v" python3 feature-extractor_amd/build.py > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  for s in $SHAPES; do
    IFS=: read n c t <<< "$s"
    python3 tools/pmc_quick.py $n $c $t "[${v:-shipped}]" 2>&1 | tail -1
  done
done
