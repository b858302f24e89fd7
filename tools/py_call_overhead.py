"""On the GPU box: where the time of one fx_push_hops call from Python goes (C channels x ONE hop, device-resident input).
Usage: python3 tools/py_call_overhead.py [N] [C]"""
import ctypes, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
fx = importlib.import_module("feature-extractor_amd")
capi = fx.capi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1
hops = torch.from_numpy(fx.synth.hops(C, 1, N)).cuda()
r = torch.empty((C, 1, 12), dtype=torch.float32, device="cuda"); s = torch.empty_like(r)
an = fx.BatchAnalyser(C, N)
def timed(f, n=2000):
    for _ in range(100): f()
    an.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    an.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
full = timed(lambda: an.push_hops(hops, out_raw=r, out_smoothed=s))
L, h = an._lib, an._h
px, pr, ps = ctypes.c_void_p(hops.data_ptr()), ctypes.c_void_p(r.data_ptr()), ctypes.c_void_p(s.data_ptr())
bare = timed(lambda: L.fx_push_hops(h, px, 1, capi.SAMPLE_F32, capi.MEM_DEVICE, pr, ps))
cur, lib = torch.cuda.current_stream(), an._torch_stream(hops.device)
def waits():
    lib.wait_stream(cur); cur.wait_stream(lib)
w = timed(waits)
print("N=%d C=%d: push_hops %.1f us | the C call alone %.1f us | the two stream waits alone %.1f us" % (N, C, full, bare, w))
