"""fx_reblock_kernel alone (the byte mover behind fx_push_samples): device time and GB/s of blocks that complete no hop, so that a call is
the re-blocking launch and nothing else.  HBM-bound: bytes moved = block read + carry written (+ carry read when something is pending).
    python tools/reblock_rate.py"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fx = importlib.import_module("feature-extractor_amd")
import torch  # noqa: E402


def main():
    for C, N, fmt, n in ((16384, 4096, torch.float32, 2047), (65536, 1024, torch.float32, 511), (16384, 4096, torch.int16, 2047), (16384, 4096, torch.float32, 1000)):
        an = fx.BatchAnalyser(C, N)
        x = (torch.rand((C, n), device="cuda") * 100).to(fmt)
        lib = an.torch_stream()
        best = None
        busy = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")       # a ~300 us fill in front of the timed launch: the launch is queued
        with torch.cuda.stream(lib):                                        # behind it, so the events bracket the kernel, not the host's call path
            for _ in range(6):
                an.reset_state()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                busy.zero_()
                e0.record(lib)
                an.push_samples(x, want_raw=False, want_smoothed=False)
                e1.record(lib)
                an.sync()
                ms = e0.elapsed_time(e1)
                best = ms if best is None or ms < best else best
                assert an.pending_samples() == n
            # second block on top of a pending one (the carry is read as well): 2 x n must stay below a hop
            if 2 * n < N // 2:
                moved2 = 2 * 2 * x.numel() * x.element_size()
                an.reset_state(); an.push_samples(x, want_raw=False, want_smoothed=False)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                busy.zero_(); e0.record(lib); an.push_samples(x, want_raw=False, want_smoothed=False); e1.record(lib); an.sync()
                print("   (second block onto %d pending samples: %.1f us = %.2f TB/s)" % (n, e0.elapsed_time(e1) * 1e3, moved2 / (e0.elapsed_time(e1) / 1e3) / 1e12))
        moved = 2 * x.numel() * x.element_size()
        # the runtime's own device-to-device copy of the same bytes (aligned, contiguous): what "a copy" reaches on this box
        y = torch.empty_like(x)
        cp = None
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            busy.zero_(); e0.record(); y.copy_(x); e1.record(); torch.cuda.synchronize()
            cp = e0.elapsed_time(e1) if cp is None or e0.elapsed_time(e1) < cp else cp
        print("   (torch's device-to-device copy of the same bytes: %.1f us = %.2f TB/s)" % (cp * 1e3, moved / (cp / 1e3) / 1e12))
        print("%6d channels x %4d samples %-8s %7.1f MB moved (read + written)  %.1f us  = %.2f TB/s = %.1f %% of 8 TB/s"
              % (C, n, str(fmt).replace("torch.", ""), moved / 1e6, best * 1e3, moved / (best / 1e3) / 1e12, 100.0 * moved / (best / 1e3) / 8e12), flush=True)
        an.close()


if __name__ == "__main__":
    main()
