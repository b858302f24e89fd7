"""Randomised GPU-vs-oracle parity (tools/stress_parity.py) with fixed seeds and a bounded case count:
random signal classes (tones, noise, impulses, gated bursts, chirps, DC), levels over 6 decades, every
window size, ragged batch shapes, split calls, all order modes and onset settings."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("seed", [3, 17])
def test_random_cases_match_oracle(gpu_fx, seed):
    import stress_parity
    cases, frames, bad, worst = stress_parity.run(seconds=60.0, seed=seed, max_cases=400, save_failures=False, verbose=True)
    assert cases >= 50
    assert bad == 0, "%d of %d random cases mismatch (worst finite rel err %g)" % (bad, cases, worst)
