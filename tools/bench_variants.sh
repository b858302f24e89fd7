#!/bin/bash
# GPU box: bench.py's line (value, ms per step, frame kernel ms) for each prebuilt variant library: bench_variants.sh name [name ...]
L=feature-extractor_amd/lib
cp $L/libfx_hip.so $L/variants/_shipped.so
for v in "$@"; do
  cp $L/variants/$v.so $L/libfx_hip.so || continue
  timeout -k 10 300 python3 bench.py ${BENCH_ARGS:---no-cpu-baseline} 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$v]', d['value'], d['ms_per_step'], d.get('roofline',{}).get('kernel_ms'), d.get('roofline',{}).get('frac'))"
done
cp $L/variants/_shipped.so $L/libfx_hip.so
