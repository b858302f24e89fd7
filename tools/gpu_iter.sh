#!/bin/bash
# one optimisation iteration on the GPU box: parity first, then the bench line
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('frames/s %.4g  ms/step %.3f  kernel ms %.3f  GB/s %.1f  frac %.4f' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['achieved'], r['frac']))"
