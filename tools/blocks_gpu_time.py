"""Is a stream of one-hop calls limited by the GPU or by the host that issues it?  The same cadence three ways -- fx_push_hops, fx_push_samples
with 512-sample blocks (whole hops, analysed in place) and with 480-sample blocks (block-fed kernels) -- timed twice: as the host sees it
(wall clock around the loop + sync) and as the device sees it when the host is OUT of the way (the calls are queued behind a spin kernel that
holds the stream for a few milliseconds; HIP events around them on the library's stream).
    python tools/blocks_gpu_time.py [channels window]      default 8192 1024"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fx = importlib.import_module("feature-extractor_amd")
import torch  # noqa: E402


def main():
    C, N = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (8192, 1024)
    H = N // 2
    calls = 64
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand((C, calls * H), generator=g, device="cuda") - 0.5
    hops = [x[:, k * H:(k + 1) * H].reshape(C, 1, H).contiguous() for k in range(calls)]
    b512 = [x[:, k * H:(k + 1) * H].contiguous() for k in range(calls)]
    b480 = [x[:, k * 480:(k + 1) * 480].contiguous() for k in range(calls)]
    an = fx.BatchAnalyser(C, N)
    raw = torch.empty((C, 1, 12), device="cuda"); sm = torch.empty_like(raw)
    ways = {"fx_push_hops, one hop per call": lambda k: an.push_hops(hops[k], out_raw=raw, out_smoothed=sm),
            "fx_push_samples, 512-sample blocks (in place)": lambda k: an.push_samples(b512[k]),
            "fx_push_samples, 480-sample blocks (block-fed)": lambda k: an.push_samples(b480[k])}
    lib = an.torch_stream()
    with torch.cuda.stream(lib):
        for name, call in ways.items():
            best_host = best_dev = None
            for rep in range(4):
                an.reset_state(); an.sync()
                t0 = time.perf_counter()
                for k in range(calls):
                    call(k)
                an.sync()
                host = (time.perf_counter() - t0) / calls * 1e6
                an.reset_state(); an.sync()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda._sleep(int(8e6))                 # ~ 4 ms at 2.1 GHz: the host queues every call meanwhile
                e0.record()
                for k in range(calls):
                    call(k)
                e1.record()
                an.sync()
                dev = e0.elapsed_time(e1) / calls * 1e3
                best_host = host if best_host is None or host < best_host else best_host
                best_dev = dev if best_dev is None or dev < best_dev else best_dev
            print("%-52s host-timed %6.1f us per call   device alone %6.1f us per call" % (name, best_host, best_dev), flush=True)


if __name__ == "__main__":
    main()
