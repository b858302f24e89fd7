export PYTHONFAULTHANDLER=1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^  File \"/usr/local/lib/python3.10/dist-packages/\(pluggy\|_pytest\)" | tail -30
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err; echo bench rc=$?
tail -c 8000 gpurun_out/r02a_bench.json; tail -5 gpurun_out/r02a_bench.err
timeout 600 tools/profile_window.sh r02a_2048 --window 2048 --channels-per-gpu 4096 --frames 32 | tail -40
timeout 600 tools/profile_window.sh r02a_4096 --window 4096 --channels-per-gpu 1024 --frames 32 | tail -40
