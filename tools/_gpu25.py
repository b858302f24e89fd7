import importlib, os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench
fx = importlib.import_module("feature-extractor_amd")
for (n, c, t) in ((4096, 1024, 42), (4096, 4096, 42), (4096, 256, 168), (2048, 4096, 32), (2048, 16384, 32), (2048, 1024, 128)):
    fr = torch.from_numpy(fx.synth.frames(c, t, n)).cuda()
    an = fx.BatchAnalyser(c, n)
    best = 1e9
    for r in range(3):
        fps, fms = bench.time_steps(an, fr, None, None, 8, warmup=3)
        best = min(best, fms)
    print(n, c, t, "kernel ms %.3f  frames/s %.4g" % (best, c * t / best * 1e3), flush=True)
    an.close(); del fr
