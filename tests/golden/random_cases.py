"""The forty random cases behind tests/golden/random/cases.npz: parameters and hops, regenerated from seeds.

draw_case(k) is deterministic (numpy's PCG64 streams and the signal recipe of tools/stress_parity.py); the fixture stores
a CRC of every case's hops so that a platform whose libm or numpy produced other samples is noticed instead of compared."""
import os
import sys

import numpy as np

CASE_SEED = 20261004
NUM_CASES = 40


def draw_case(k):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
    from stress_signals import make_signal
    rng = np.random.default_rng([CASE_SEED, k])
    N = int(rng.choice([256, 512, 1024, 1024, 2048, 2048, 4096]))
    C = int(rng.integers(1, 5))
    T = int(rng.integers(4, 29))
    if k % 10 == 9 and N <= 1024:         # a few long calls: the frame kernel cuts them in time
        T, C = int(rng.integers(130, 200)), 2
    p = {"N": N, "C": C, "T": T, "order": int(rng.integers(0, 3)), "onset_type": int(rng.integers(0, 3)),
         "onset_window": int(rng.integers(1, 22)), "sensitivity": float(np.float32(rng.uniform(0, 2))),
         "gain": float(rng.choice([1.0, 1.0, 0.5, 3.0])), "sample_rate": float(rng.choice([48000.0, 48000.0, 44100.0]))}
    return p, make_signal(rng, C, T, N)
