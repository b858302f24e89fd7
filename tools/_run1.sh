set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/t_blocks2.log 2>&1 || { tail -30 gpurun_out/t_blocks2.log; exit 1; }
tail -2 gpurun_out/t_blocks2.log
for r in 1 2; do
for shape in "8192 1024 2000 32" "8192 1024 4097 16" "4096 2048 4000 16" "1024 4096 10000 16" "8192 1024 480 64" "8192 1024 1024 64"; do
  for mode in "" reblock; do echo -n "[$mode] "; timeout -k 10 120 python3 tools/device_blocks.py $shape $mode 2>&1 | tail -1; done
done; done | tee gpurun_out/blocks_general.txt
timeout -k 10 300 python3 tools/stress_parity.py 150 90909 > gpurun_out/stress_90909.txt 2>&1; tail -3 gpurun_out/stress_90909.txt
