"""Host hops -> pinned ring -> GPU: frames/s of fx_stream_push over fill threads x ring slots x fill mode (non-temporal stores / memcpy),
three passes each, beside the pinned hipMemcpyAsync rate of the same bytes (best of five).  One shape per run:
    python tools/ingest_sweep.py [window channels hops format]        e.g. 1024 1024 64 s16"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fx = importlib.import_module("feature-extractor_amd")
import torch  # noqa: E402


def memcpy_rate(nbytes, runs=5, copies=4):
    host = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    devb = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    best = 0.0
    devb.copy_(host, non_blocking=True); torch.cuda.synchronize()
    for _ in range(runs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(copies):
            devb.copy_(host, non_blocking=True)
        e1.record(); torch.cuda.synchronize()
        best = max(best, nbytes * copies / (e0.elapsed_time(e1) / 1e3) / 1e9)
    return best


def main():
    N, C, T = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1024, 1024, 64)
    fmt = sys.argv[4] if len(sys.argv) > 4 else "s16"
    src32 = np.ascontiguousarray(np.tile(fx.synth.hops(128, T, N), (C // 128, 1, 1)))
    src = {"f32": src32, "f16": src32.astype(np.float16), "s16": np.round(src32 * 32767.0).astype(np.int16)}[fmt]
    peak = memcpy_rate(src.nbytes)
    print("%d ch x %d hops x %d-pt %s: %.1f MB per batch, pinned H2D %.1f GB/s" % (C, T, N, fmt, src.nbytes / 1e6, peak), flush=True)
    for streaming in (1, 0):
        for slots in (3, 4, 6):
            for threads in (4, 6, 8, 12):
                an = fx.BatchAnalyser(C, N)
                an.set_tuning(stream_fill_streaming=streaming)
                st = fx.HopStream(an, T, slots=slots, dtype=src.dtype)
                rates = []
                for _ in range(3):
                    steps, warm = 24, 4
                    for k in range(steps + warm):
                        if k == warm:
                            while st.in_flight():
                                st.collect(want_raw=False)
                            t0 = time.perf_counter()
                        if st.in_flight() == slots:
                            st.collect(want_raw=False)
                        st.push(src, fill_threads=threads)
                    while st.in_flight():
                        st.collect(want_raw=False)
                    rates.append(src.nbytes * steps / (time.perf_counter() - t0) / 1e9)
                st.close(); an.close()
                print("  %-9s slots %d threads %2d: %5.1f %5.1f %5.1f GB/s  = %.2f .. %.2f of the link" % ("streaming" if streaming else "memcpy", slots, threads,
                      rates[0], rates[1], rates[2], min(rates) / peak, max(rates) / peak), flush=True)


if __name__ == "__main__":
    main()
