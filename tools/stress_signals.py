"""Random test signals of tools/stress_parity.py (numpy only: also used by tests/golden/random_cases.py)."""
import numpy as np


def make_signal(rng, C, T, N):
    n = T * N // 2
    t = np.arange(n)
    x = np.zeros((C, n))
    for c in range(C):
        kind = rng.integers(0, 8)
        level = 10.0 ** rng.uniform(-5, 1.5)
        if kind == 0:      # harmonic tone
            f = rng.uniform(40, 6000)
            ph = 2 * np.pi * f * t / 48000
            x[c] = sum(rng.uniform(0, 1) / (h + 1) * np.sin((h + 1) * ph + rng.uniform(0, 6)) for h in range(rng.integers(1, 8)))
        elif kind == 1:    # noise
            x[c] = rng.normal(0, 1, n)
        elif kind == 2:    # tone + noise
            f = rng.uniform(40, 6000)
            x[c] = np.sin(2 * np.pi * f * t / 48000) + rng.uniform(0, 0.3) * rng.normal(0, 1, n)
        elif kind == 3:    # sparse impulses
            x[c, rng.integers(0, n, max(1, n // 3000))] = rng.normal(0, 1, max(1, n // 3000))
        elif kind == 4:    # gated bursts with exact silence
            gate = (rng.random(T) > 0.5).repeat(N // 2)
            x[c] = gate * np.sin(2 * np.pi * rng.uniform(80, 2000) * t / 48000)
        elif kind == 5:    # chirp
            f = np.linspace(rng.uniform(50, 500), rng.uniform(500, 12000), n)
            x[c] = np.sin(2 * np.pi * np.cumsum(f) / 48000)
        elif kind == 6:    # DC + tiny noise
            x[c] = rng.uniform(-1, 1) + 1e-3 * rng.normal(0, 1, n)
        else:              # silence with one loud hop
            h = rng.integers(0, T)
            x[c, h * N // 2:(h + 1) * N // 2] = rng.normal(0, 1, N // 2)
        x[c] *= level
    return x.astype(np.float32).reshape(C, T, N // 2)
