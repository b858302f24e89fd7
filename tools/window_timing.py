"""On the GPU box: frames/s and frame-kernel time of fx_process_frames at a shape (best of three passes of 20 steps).
Usage: python3 tools/window_timing.py N C T [analysers] [low_latency]   e.g.  512 1024 512   |   1024 1024 512 spectral"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
fx = importlib.import_module("feature-extractor_amd")
N, C, T = (int(v) for v in sys.argv[1:4])
analysers = sys.argv[4] if len(sys.argv) > 4 else "both"
low = len(sys.argv) > 5 and sys.argv[5] == "low_latency"
frames = torch.from_numpy(fx.synth.frames(C, T, N)).cuda()
an = fx.BatchAnalyser(C, N, analysers=analysers, low_latency=low)
best = max(bench.time_steps(an, frames, None, None, 20, warmup=5) for _ in range(3))
print("N=%d C=%d T=%d %s%s: %.4g frames/s, frame kernel %.3f ms" % (N, C, T, analysers, " low_latency" if low else "", best[0], best[1]), flush=True)
