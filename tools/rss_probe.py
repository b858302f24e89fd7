"""Does resident memory grow with the number of calls?  Each path of the library, tens of thousands of calls, /proc/self/statm before and after
(after a warm-up of the same kind of calls).  A diagnostic for long-running hosts (tools/live_soak.cpp runs for minutes)."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fx = importlib.import_module("feature-extractor_amd")
import torch  # noqa: E402


def rss_kb():
    return int(open("/proc/self/statm").read().split()[1]) * (os.sysconf("SC_PAGE_SIZE") // 1024)


def probe(name, make, step, calls=20000, warm=2000):
    ctx = make()
    for _ in range(warm):
        step(ctx)
    r0 = rss_kb(); t0 = time.perf_counter()
    for _ in range(calls):
        step(ctx)
    r1 = rss_kb()
    print("%-58s %6d calls: %+8d KB  (%.2f KB per call, %.1f us per call)" % (name, calls, r1 - r0, (r1 - r0) / calls, (time.perf_counter() - t0) / calls * 1e6), flush=True)


def main():
    C, N = 256, 1024
    H = N // 2
    hop_h = np.random.default_rng(0).standard_normal((C, 1, H)).astype(np.float32) * 0.1
    blk_h = np.ascontiguousarray(hop_h[:, 0, :480])
    hop_d = torch.from_numpy(hop_h).cuda()
    blk_d = torch.from_numpy(blk_h).cuda()

    def ring():
        an = fx.BatchAnalyser(C, N)
        return an, fx.HopStream(an, 2, slots=3, dtype=np.float32)

    def ring_step(ctx):
        an, st = ctx
        if st.in_flight() == 3:
            st.collect_samples()
        st.push_samples(blk_h)

    def ring1():
        an = fx.BatchAnalyser(C, N)
        return an, fx.HopStream(an, 1, slots=3, dtype=np.float32)

    def ring1_step(ctx):
        an, st = ctx
        if st.in_flight() == 3:
            st.collect()
        st.push(hop_h)

    probe("nothing (the interpreter and numpy alone)", lambda: None, lambda ctx: hop_h.sum())
    probe("fx_push_hops, device hop, no outputs", lambda: fx.BatchAnalyser(C, N), lambda an: an.push_hops(hop_d, want_raw=False, want_smoothed=False))
    probe("fx_push_hops, host hop, host outputs", lambda: fx.BatchAnalyser(C, N), lambda an: an.push_hops(hop_h))
    probe("fx_push_samples, device block of 480", lambda: fx.BatchAnalyser(C, N), lambda an: an.push_samples(blk_d, want_raw=False, want_smoothed=False))
    probe("fx_push_samples, host block of 480", lambda: fx.BatchAnalyser(C, N), lambda an: an.push_samples(blk_h))
    probe("ring, one hop per batch (fx_hop_kernel, flag polled)", ring1, ring1_step)
    probe("ring, fx_stream_push_samples of 480 (three queues, events)", ring, ring_step)
    probe("fx_get_smoothed to the host", lambda: fx.BatchAnalyser(C, N), lambda an: an.get_features())
    probe("fx_get_osc_datagrams to the host", lambda: fx.BatchAnalyser(C, N), lambda an: an.osc_datagrams())


if __name__ == "__main__":
    main()
