"""On the GPU box: rocprofv3 --pmc passes (8 SQ counters; with --traffic also FETCH_SIZE and WRITE_SIZE, each in its own pass, as
the MI355X guide prescribes) + kernel time of the frame / pair kernel at a shape, printed per frame.
Usage: python3 tools/pmc_quick.py [--traffic] [--hops] [--analysers spectral|harmonic|both] N C T [label]
(--hops: the input is hops through fx_push_hops -- half a window per frame, consecutive frames of a channel overlap -- instead of whole frames)
(tuning through the FX_* environment, e.g. FX_WAVES_PER_FRAME=2).  The library is built HERE, in the parent, before any profiler
starts: the profiled child only loads it."""
import importlib, os, subprocess, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
argv = sys.argv[1:]
traffic = "--traffic" in argv
hops = "--hops" in argv
argv = [a for a in argv if a not in ("--traffic", "--hops")]
analysers = "both"
if "--analysers" in argv:
    k = argv.index("--analysers"); analysers = argv[k + 1]; del argv[k:k + 2]
N, C, T = (int(v) for v in argv[:3])
label = argv[3] if len(argv) > 3 else ""
fx = importlib.import_module("feature-extractor_amd")
importlib.import_module("feature-extractor_amd.build").build()          # never under the profiler
inp = "/tmp/fx_pq_%d_%d_%d%s.npy" % (N, C, T, "_hops" if hops else "")
if not os.path.exists(inp):
    np.save(inp, fx.synth.hops(C, T, N) if hops else fx.synth.frames(C, T, N))
exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
counters = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"]
child = [sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", "--window", str(N), "--channels-per-gpu", str(C), "--frames", str(T),
         "--input-file", inp, "--steps", "6", "--warmup", "2", "--analysers", analysers]
out = {}
passes = [("pmc", ["--pmc"] + counters), ("trace", ["--kernel-trace", "--stats"])]
if traffic:
    passes += [("fetch", ["--pmc", "FETCH_SIZE"]), ("write", ["--pmc", "WRITE_SIZE"])]
for name, args in passes:
    d = tempfile.mkdtemp(prefix="fx_pq_", dir="/tmp")
    env = dict(os.environ); env["TMPDIR"] = "/tmp"
    p = subprocess.run([exe] + args + ["--output-format", "csv", "-d", d, "--"] + child, env=env, cwd="/tmp", capture_output=True, text=True)
    if p.returncode:
        print("rocprofv3 failed:", p.stderr[-400:]); sys.exit(1)
    if name != "trace":
        for kern in ((out["kernel"],) if "kernel" in out else ("fx_pair_kernel<%d" % N, "fx_frame_kernel<%d" % N)):
            vals, n = bench._parse_counter_csv(d, kern)
            if vals:
                out.update(vals); out["kernel"] = kern
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
line = ("%s %s %s%s N=%d C=%d T=%d: %.3f ms  %.4g frames/s | per frame: VALU %.0f SALU %.0f LDS %.0f | per wave-cycle: VALU-active %.3f wait_any %.3f wait_inst %.3f | waves %d, wave-cycles/frame %.0f"
        % (label, out["kernel"], analysers, " hops" if hops else "", N, C, T, out["avg_ns"] / 1e6, fr / (out["avg_ns"] / 1e9), out["SQ_INSTS_VALU"] / fr, out["SQ_INSTS_SALU"] / fr, out["SQ_INSTS_LDS"] / fr,
           out["SQ_ACTIVE_INST_VALU"] / wc, out["SQ_WAIT_ANY"] / wc, out["SQ_WAIT_INST_ANY"] / wc, out["SQ_WAVES"], 4 * wc / fr))
if traffic:
    hbm = (2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024.0          # gfx950: FETCH_SIZE counts half the bytes of wide streaming reads
    alg = ((2 * N if hops else 4 * N) + 48) * fr              # hops: every sample is new once
    line += " | HBM %.4g B per launch = %.3f x algorithmic (read %.4g, written %.4g)" % (hbm, hbm / alg, 2048.0 * out["FETCH_SIZE"], 1024.0 * out["WRITE_SIZE"])
print(line, flush=True)
