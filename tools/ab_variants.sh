#!/bin/bash
# On the GPU box: A/B of library variants built beforehand with tools/build_variants.py (feature-extractor_amd/lib/variants/<name>.so),
# interleaved so that clock drift cancels.  Whatever happens -- a time-out, Ctrl-C, a failing variant -- the SHIPPED library is back
# in place when this script ends (trap), so that nothing run afterwards measures or tests an experiment by accident.
#   here   : python tools/build_variants.py small a=-DFOO=1 b=-DFOO=2
#   GPU box: tools/ab_variants.sh "a b" "1024 1024 512" ["2048 4096 64" ...]       (shapes: N C T for tools/window_timing.py)
set -u
L=feature-extractor_amd/lib
names=$1; shift
cp $L/libfx_hip.so $L/variants/_shipped.so
trap 'cp $L/variants/_shipped.so $L/libfx_hip.so' EXIT
for round in 1 2; do
  for v in $names; do
    cp $L/variants/$v.so $L/libfx_hip.so || continue
    for shape in "$@"; do
      echo -n "$v: "; timeout -k 10 120 python3 tools/window_timing.py $shape 2>&1 | tail -1
    done
  done
done
