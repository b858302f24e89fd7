#!/bin/bash
# On the GPU box: one-hop round trip (tools/stream_latency.cpp) per build variant, same box, alternating.  Usage: tools/hop_latency_ab.sh "<flags>" ...
g++ -O2 -std=c++14 -I include tools/stream_latency.cpp -o /tmp/stream_latency -L feature-extractor_amd/lib -lfx_hip -Wl,-rpath,$PWD/feature-extractor_amd/lib || exit 1
for rep in 1 2; do
for v in "$@"; do
  FX_EXTRA_HIPCC_FLAGS="$v" python3 feature-extractor_amd/build.py > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  for n in 4096 2048 1024; do
    a=$(/tmp/stream_latency $n 1 1 4000 | grep "round trip" | sed 's/ per call.*//')
    if [ $n != 1024 ]; then b=$(FX_WAVES_PER_FRAME=2 /tmp/stream_latency $n 1 1 4000 | grep "round trip" | sed 's/ per call.*//'); else b=""; fi
    echo "[${v:-shipped}] $n-pt: three waves $a | pairs $b"
  done
done
done
