"""Debug helper: run the HIP path and the CPU oracle on the same inputs and print differences."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fx = importlib.import_module("feature-extractor_amd")
from oracle import fx_oracle as fo


def compare(name, got, want, rtol=1e-5):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    both_nan = np.isnan(got) & np.isnan(want)
    same_inf = np.isinf(got) & np.isinf(want) & (np.sign(got) == np.sign(want))
    with np.errstate(invalid="ignore", divide="ignore"):
        err = np.abs(got - want) / np.maximum(np.abs(want), 1e-30)
    err = np.where(both_nan | same_inf | (got == want), 0.0, err)
    err = np.where(np.isnan(err), np.inf, err)
    worst = err.reshape(-1, 12).max(axis=0)
    bad = (err > rtol)
    print("%s: max rel err per feature:" % name)
    for i, n in enumerate(fx.FEATURE_NAMES):
        nb = int(bad.reshape(-1, 12)[:, i].sum())
        print("   %-9s %.3e   mismatches %d / %d" % (n, worst[i], nb, err.reshape(-1, 12).shape[0]))
    return bad


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    C = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    mode = sys.argv[4] if len(sys.argv) > 4 else "hops"
    x = fx.synth.hops(C, T, N) if mode == "hops" else fx.synth.frames(C, T, N)
    an = fx.BatchAnalyser(C, N)
    an.set_tuning(call_timing=1)          # (one-frame calls record no timing events by default: fx_last_kernel_ms would refuse)
    t0 = time.time()
    if mode == "hops":
        raw, sm = an.push_hops(x)
        oraw, osm = fo.push_hops(x, N)
    else:
        raw, sm = an.process_frames(x)
        oraw, osm = fo.process_frames(x, N)
    print("N=%d C=%d T=%d mode=%s  kernel ms %s" % (N, C, T, mode, an.last_kernel_ms()))
    np.set_printoptions(precision=6, suppress=True, linewidth=220)
    bad = compare("raw", raw, oraw)
    compare("smoothed", sm, osm)
    idx = np.argwhere(bad)
    for c, t, f in idx[:10]:
        print("  mismatch c=%d t=%d %s: gpu %r oracle %r" % (c, t, fx.FEATURE_NAMES[f], raw[c, t, f], oraw[c, t, f]))
    print("gpu   raw[0,-1]:", raw[0, -1])
    print("oracle raw[0,-1]:", oraw[0, -1])


if __name__ == "__main__":
    main()
