set -u
export TMPDIR=/tmp
R=$(pwd)
mkdir -p gpurun_out/r06d
for cfg in "8192 1024 480 64" "8192 1024 512 64" "8192 1024 480 64 reblock" "1024 1024 480 64" "1024 1024 512 64"; do
  tag=$(echo $cfg | tr ' ' '_')
  rm -rf gpurun_out/r06d/prof_$tag
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06d/prof_$tag -- python3 $R/tools/device_blocks.py $cfg > $R/gpurun_out/r06d/$tag.log 2>&1)
  tail -1 gpurun_out/r06d/$tag.log
  f=$(find gpurun_out/r06d/prof_$tag -name "*kernel_stats.csv" | head -1)
  cp $f gpurun_out/r06d/${tag}_kernel_stats.csv
  head -8 $f | cut -c1-200
  rm -rf gpurun_out/r06d/prof_$tag
done
