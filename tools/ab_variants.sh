#!/bin/bash
# On the GPU box: A/B of library variants built beforehand with tools/build_variants.py (feature-extractor_amd/lib/variants/<name>.so),
# interleaved so that clock drift cancels.  The variant is selected by path (FX_LIBRARY_OVERRIDE, feature-extractor_amd/capi.py): the SHIPPED
# library is never overwritten, so nothing run afterwards can measure or test an experiment by accident.
#   here   : python tools/build_variants.py small a=-DFOO=1 b=-DFOO=2
#   GPU box: tools/ab_variants.sh "a b" "1024 1024 512" ["2048 4096 64" ...]       (shapes: N C T for tools/window_timing.py)
set -u
L=$(pwd)/feature-extractor_amd/lib
names=$1; shift
for round in 1 2; do
  for v in $names; do
    [ -f $L/variants/$v.so ] || continue
    for shape in "$@"; do
      echo -n "$v: "; FX_LIBRARY_OVERRIDE=$L/variants/$v.so timeout -k 10 120 python3 tools/window_timing.py $shape 2>&1 | tail -1
    done
  done
done
