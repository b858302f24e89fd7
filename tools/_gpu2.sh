timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4
export BENCH_ARGS="--window 2048 --channels-per-gpu 4096 --frames 48"
tools/variants.sh "" "-DFX_EXP_2048_LDS_TW"
FX_CHANNELS_PER_WG=2 tools/variants.sh ""
python3 feature-extractor_amd/build.py > /dev/null 2>&1
timeout 300 python tools/stress_parity.py 60 99 2>&1 | tail -3
