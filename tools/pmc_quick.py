"""On the GPU box: one rocprofv3 --pmc pass (8 SQ counters) + kernel time of the frame / pair kernel at a shape, printed per frame.
Usage: python3 tools/pmc_quick.py N C T [label]      (tuning through the FX_* environment, e.g. FX_WAVES_PER_FRAME=2)"""
import importlib, os, subprocess, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
N, C, T = (int(v) for v in sys.argv[1:4])
label = sys.argv[4] if len(sys.argv) > 4 else ""
fx = importlib.import_module("feature-extractor_amd")
inp = "/tmp/fx_pq_%d_%d_%d.npy" % (N, C, T)
if not os.path.exists(inp):
    np.save(inp, fx.synth.frames(C, T, N))
exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
counters = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"]
child = [sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", "--window", str(N), "--channels-per-gpu", str(C), "--frames", str(T),
         "--input-file", inp, "--steps", "6", "--warmup", "2"]
out = {}
for name, args in (("pmc", ["--pmc"] + counters), ("trace", ["--kernel-trace", "--stats"])):
    d = tempfile.mkdtemp(prefix="fx_pq_", dir="/tmp")
    env = dict(os.environ); env["TMPDIR"] = "/tmp"
    p = subprocess.run([exe] + args + ["--output-format", "csv", "-d", d, "--"] + child, env=env, cwd="/tmp", capture_output=True, text=True)
    if p.returncode:
        print("rocprofv3 failed:", p.stderr[-400:]); sys.exit(1)
    if name == "pmc":
        for kern in ("fx_pair_kernel<%d" % N, "fx_frame_kernel<%d" % N):
            vals, n = bench._parse_counter_csv(d, kern)
            if vals:
                out = vals; out["kernel"] = kern
                break
    else:
        import csv, glob
        for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if out.get("kernel", "?") in row["Name"]:
                    out["avg_ns"] = float(row["AverageNs"])
    shutil.rmtree(d, ignore_errors=True)
fr = C * T
wc = out["SQ_WAVE_CYCLES"]
print("%s %s N=%d C=%d T=%d: %.3f ms  %.4g frames/s | per frame: VALU %.0f SALU %.0f LDS %.0f | per wave-cycle: VALU-active %.3f wait_any %.3f wait_inst %.3f | waves %d, wave-cycles/frame %.0f"
      % (label, out["kernel"], N, C, T, out["avg_ns"] / 1e6, fr / (out["avg_ns"] / 1e9), out["SQ_INSTS_VALU"] / fr, out["SQ_INSTS_SALU"] / fr, out["SQ_INSTS_LDS"] / fr,
         out["SQ_ACTIVE_INST_VALU"] / wc, out["SQ_WAIT_ANY"] / wc, out["SQ_WAIT_INST_ANY"] / wc, out["SQ_WAVES"], 4 * wc / fr), flush=True)
