tools/profile_window.sh r02_2048 --window 2048 --channels-per-gpu 4096 --frames 32 > /dev/null 2>&1
grep -A17 "PMC per launch.*fx_frame_kernel" gpurun_out/prof_r02_2048/summary.txt | head -18; grep "trace void fxk::fx_frame" gpurun_out/prof_r02_2048/summary.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; echo rc=$?
python3 -c "
import json
d=json.loads(open('gpurun_out/r02_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline'].get('valu_issue_frac'), d['other_windows'], d['data_dependence']['noise']['relative_to_synth'], d['spectral_only']['value'])"
hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates > gpurun_out/r02_valu_rates.txt 2>&1; head -40 gpurun_out/r02_valu_rates.txt
