#!/bin/bash
# First contact with a multi-GPU node: comm_ranks 2 / 8 (C++, RCCL through the C ABI), bench.py --gpus 1, 2, 4, 8 and ONE JSON with
# every rank's frames/s, kernel times, RCCL rank count and gather times on the side stream.  See tools/first_multigpu_run.py.
#   tools/first_multigpu_run.sh [--out gpurun_out/first_multigpu.json] [--steps 20] [--warmup 5] [--max-gpus 8]
set -eu
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
# python3 is started as a child (not exec'ed) and this shell makes no GPU call of its own
python3 tools/first_multigpu_run.py "$@"
