timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4
b() { timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra $2 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-70s kernel ms %.3f  frames/s %.4g  hbm %.4f' % ('$1 $2', r['avg_launch_ms'], d['value'], r['frac']))"; }
b "" ""
b "" "--window 2048 --channels-per-gpu 4096 --frames 48"
b "" "--window 4096 --channels-per-gpu 1024 --frames 42"
timeout 300 python tools/stress_parity.py 100 4711 2>&1 | tail -3
