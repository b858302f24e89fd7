"""On the GPU box: device time of ONE frame per channel through the batch kernels (frame / pair kernel + fused tail), by kernel."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
fx = importlib.import_module("feature-extractor_amd")
for N in (2048, 4096):
    for C in (1, 8):
        hops = fx.synth.hops(C, 40, N, first_channel=24)
        for wpf in (1, 2):
            an = fx.BatchAnalyser(C, N)
            an.set_tuning(waves_per_frame=wpf, one_hop_kernel=0, call_timing=1)       # (one-frame calls record no timing events by default)
            ts = []
            for t in range(40):
                an.push_hops(hops[:, t:t + 1])
                ts.append(an.last_kernel_ms()[0] * 1e3)
            an.close()
            print("N=%d C=%d waves_per_frame=%d: frame kernel median %.1f us (min %.1f)" % (N, C, wpf, float(np.median(ts[5:])), min(ts[5:])), flush=True)
