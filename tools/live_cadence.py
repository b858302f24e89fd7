"""On the GPU box: microseconds per call of ONE hop per channel (device-resident hops, fx_push_hops back to back), hop kernel against
the batch kernels at several workgroup shapes.  Usage: python3 tools/live_cadence.py [--default-only] [N [C ...]]
(--default-only: just the path a default context takes -- what `rocprofv3 --kernel-trace --stats` of this script should show)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
fx = importlib.import_module("feature-extractor_amd")
default_only = "--default-only" in sys.argv
sys.argv = [a for a in sys.argv if a != "--default-only"]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for C in ([int(a) for a in sys.argv[2:]] or (1, 64, 256, 512, 1024, 2048, 4096, 8192)):
    hops = torch.from_numpy(fx.synth.hops(C, 8, N)).cuda()
    views = [hops[:, k:k + 1].contiguous() for k in range(8)]
    r = torch.empty((C, 1, 12), dtype=torch.float32, device="cuda")
    s = torch.empty_like(r)
    row = []
    for name, knobs in ((("default", dict()),) if default_only else (("hop", dict(one_hop_kernel=1)), ("batch ch1", dict(one_hop_kernel=0, channels_per_workgroup=1)), ("batch ch4", dict(one_hop_kernel=0, channels_per_workgroup=4)),
                        ("batch ch8", dict(one_hop_kernel=0, channels_per_workgroup=8)))):
        an = fx.BatchAnalyser(C, N)
        an.set_tuning(**knobs)
        n = 300
        torch.cuda.synchronize()
        with torch.cuda.stream(an.torch_stream()):      # on the library's stream: no cross-stream waits per call (25 us from Python)
            for k in range(n + 30):
                if k == 30:
                    an.sync(); t0 = time.perf_counter()
                an.push_hops(views[k % 8], out_raw=r, out_smoothed=s)
            an.sync()
            us = (time.perf_counter() - t0) / n * 1e6
        an.close()
        row.append("%s %.1f us" % (name, us))
    print("N=%d C=%5d: %s" % (N, C, " | ".join(row)), flush=True)
