"""A stream of device blocks through fx_push_samples, for rocprofv3 (kernel trace or --pmc FETCH_SIZE / WRITE_SIZE passes):
    python tools/device_blocks.py [channels window block blocks][reblock]        default 8192 1024 480 64
and, with block >= 100000, ONE long block per call (the re-blocking kernel at HBM-bound sizes): e.g.  16384 4096 262243 1
Prints the algorithmic bytes of the re-blocking launches (2 x sample bytes moved) so that the counters can be held against them."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fx = importlib.import_module("feature-extractor_amd")
import torch  # noqa: E402


def main():
    C, N, n, blocks = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (8192, 1024, 480, 64)
    an = fx.BatchAnalyser(C, N)
    if len(sys.argv) > 5 and sys.argv[5] == "reblock":
        an.set_test_hooks(16)                   # FX_HOOK_NO_BLOCK_FEED: every call through fx_reblock_kernel, as before round 6
    elif len(sys.argv) > 5 and sys.argv[5].startswith("hooks="):
        an.set_test_hooks(int(sys.argv[5][6:]))  # e.g. hooks=32: two-hop calls through the batch kernels' two-frame form
    g = torch.Generator(device="cuda").manual_seed(1)
    pieces = [(torch.rand((C, n), generator=g, device="cuda") - 0.5) for _ in range(min(blocks, 8))]
    moved = 0
    with torch.cuda.stream(an.torch_stream()):
        for rep in range(3):
            an.reset_state()
            t0 = time.perf_counter()
            for b in range(blocks):
                pending = an.pending_samples()
                an.push_samples(pieces[b % len(pieces)], want_raw=False)
                if rep == 2:
                    moved += 2 * (pending + n) * C * 4
            an.sync()
            dt = time.perf_counter() - t0
    frames = (n * blocks) // (N // 2)
    print("%d channels x %d-pt, %d blocks of %d samples: %.1f us per call, %.4g frames/s; re-blocking launches of the last pass moved %.4g B (algorithmic: read + written)"
          % (C, N, blocks, n, dt / blocks * 1e6, C * frames / dt, moved), flush=True)


if __name__ == "__main__":
    main()
