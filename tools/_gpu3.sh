tools/profile.sh r02 > gpurun_out/prof_r02.log 2>&1
tools/profile_window.sh r02_2048 --window 2048 --channels-per-gpu 4096 --frames 32 > /dev/null 2>&1
tools/profile_window.sh r02_4096 --window 4096 --channels-per-gpu 1024 --frames 42 > /dev/null 2>&1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02_noise/trace -- python3 bench.py --no-cpu-baseline --no-extra --signal noise --steps 20 > gpurun_out/prof_r02_noise/trace.log 2>&1
python3 tools/pmc_summary.py gpurun_out/prof_r02_noise > gpurun_out/prof_r02_noise/summary.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; echo rc=$?
tail -c 3000 gpurun_out/r02_bench.json
g++ -O2 -std=c++14 -I include tools/stream_latency.cpp -o /tmp/stream_latency -L feature-extractor_amd/lib -lfx_hip -Wl,-rpath,$PWD/feature-extractor_amd/lib && /tmp/stream_latency 4096 1 1 4000 > gpurun_out/r02_stream_latency.txt 2>&1; cat gpurun_out/r02_stream_latency.txt
timeout 300 python bench.py --stream --fp16 --window 4096 --channels-per-gpu 1 --frames 1 --steps 2000 --warmup 100 2>/dev/null | tail -1 > gpurun_out/r02_stream_py.json
timeout 300 python bench.py --stream --window 1024 --channels-per-gpu 1024 --frames 64 --steps 30 --warmup 3 2>/dev/null | tail -1 >> gpurun_out/r02_stream_py.json
timeout 300 python bench.py --stream --fp16 --window 1024 --channels-per-gpu 1024 --frames 64 --steps 30 --warmup 3 2>/dev/null | tail -1 >> gpurun_out/r02_stream_py.json
cat gpurun_out/r02_stream_py.json | cut -c1-200
