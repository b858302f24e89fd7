"""On the GPU box: frames/s of the one-wavefront-per-frame kernel against the pair kernel at the bench's 2048 / 4096 shapes."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
fx = importlib.import_module("feature-extractor_amd")
shapes = [(2048, 4096, 64), (4096, 1024, 64)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]]
for (N, C, T) in shapes:
    fr = torch.from_numpy(fx.synth.frames(C, T, N)).cuda()
    for wpf in (1, 2):
        an = fx.BatchAnalyser(C, N)
        an.set_tuning(waves_per_frame=wpf)
        best = max(bench.time_steps(an, fr, None, None, 20, warmup=5) for _ in range(3))
        print("N=%d C=%d T=%d waves_per_frame=%d: %.4g frames/s, kernel %.3f ms" % (N, C, T, wpf, best[0], best[1]), flush=True)
        an.close()
