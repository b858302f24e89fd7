#!/bin/bash
# On the GPU box: rebuild the library with each set of extra flags and print the frame-kernel time.
# Usage: tools/variants.sh "<flags for variant 1>" "<flags for variant 2>" ...   ("" = the shipped build)
for v in "$@"; do
  FX_EXTRA_HIPCC_FLAGS="$v" python3 feature-extractor_amd/build.py > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-40s kernel ms %.3f  frames/s %.4g' % ('$v' or '(shipped)', r['avg_launch_ms'], d['value']))"
done
