#!/bin/bash
# On the GPU box: 2048-pt one-wavefront kernel under build variants x workgroup shapes.  Usage: tools/variants_2048.sh "<flags>:<ch>:<k>" ...
for spec in "$@"; do
  IFS=: read v ch k <<< "$spec"
  FX_EXTRA_HIPCC_FLAGS="$v" python3 feature-extractor_amd/build.py > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  FX_WAVES_PER_FRAME=1 FX_CHANNELS_PER_WG=$ch FX_WAVES=$k python3 tools/pmc_quick.py 2048 4096 64 "[${v:-shipped} ${ch}x${k}]" 2>&1 | tail -1
done
