"""On the GPU box, with a library built with FX_EXTRA_HIPCC_FLAGS=-DFX_PAIR_STAMPS: where the two wavefronts of a pair spend
the last frame of channel 0 -- shader-clock stamps at entry and exit of every wait() of the pair kernel.
Usage: python3 tools/pair_stamps.py N [C] [T]"""
import ctypes, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
fx = importlib.import_module("feature-extractor_amd")
N = int(sys.argv[1]); C = int(sys.argv[2]) if len(sys.argv) > 2 else 1; T = int(sys.argv[3]) if len(sys.argv) > 3 else 2
an = fx.BatchAnalyser(C, N)
an.set_tuning(waves_per_frame=2, one_hop_kernel=0, debug_flags=2)
hops = fx.synth.hops(C, T, N, first_channel=24)
for _ in range(3):
    an.push_hops(hops)
buf = (ctypes.c_ulonglong * 128)()
fx.capi.check(an._lib.fx_debug_read_stamps(an._h, buf))
st = np.array(buf[:], dtype=np.uint64).reshape(2, 64).astype(np.int64)
t0 = min(st[0][0], st[1][0])
n = int(max((st[0] > 0).sum(), (st[1] > 0).sum()))
print("N=%d C=%d T=%d: frame of %d / %d cycles (wave 0 / wave 1); per wait(): cycles since frame start at entry, cycles waited" % (N, C, T, st[0][n - 1] - st[0][0], st[1][n - 1] - st[1][0]))
tot = [0, 0]
for i in range(1, n - 1, 2):
    row = []
    for w in (0, 1):
        row.append("w%d enter %6d waited %5d" % (w, st[w][i] - t0, st[w][i + 1] - st[w][i]))
        tot[w] += st[w][i + 1] - st[w][i]
    print("  sync %2d: %s" % ((i + 1) // 2, "   ".join(row)))
print("waited in total: wave 0 %d, wave 1 %d cycles" % (tot[0], tot[1]))
