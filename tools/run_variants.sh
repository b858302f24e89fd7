#!/bin/bash
# GPU box: tools/pmc_quick.py for each prebuilt variant library (tools/build_variants.py): run_variants.sh "N C T" name [name ...]
L=feature-extractor_amd/lib
shape=$1; shift
cp $L/libfx_hip.so $L/variants/_shipped.so
for v in "$@"; do
  cp $L/variants/$v.so $L/libfx_hip.so || continue
  timeout -k 10 120 python3 tools/pmc_quick.py $shape "[$v]" 2>&1 | tail -1 | cut -c1-175
done
cp $L/variants/_shipped.so $L/libfx_hip.so
