#!/bin/bash
# On the GPU box: device timeline of three ways of feeding the same cadence (8192 channels x 1024 points): 480-sample blocks (block-fed kernels),
# 512-sample blocks (whole hops in place through fx_push_samples) and the bench's fx_push_hops loop; tools/trace_gaps.py reads the gaps.
set -u
export TMPDIR=/tmp
R=$(pwd)
mkdir -p gpurun_out/gaps
for cfg in "8192 1024 480 64" "8192 1024 512 64"; do
  tag=$(echo $cfg | tr ' ' '_')
  rm -rf gpurun_out/gaps/prof_$tag
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gaps/prof_$tag -- python3 $R/tools/device_blocks.py $cfg > $R/gpurun_out/gaps/$tag.log 2>&1)
  tail -1 gpurun_out/gaps/$tag.log | cut -c1-110
  python3 tools/trace_gaps.py gpurun_out/gaps/prof_$tag
  rm -rf gpurun_out/gaps/prof_$tag
done
