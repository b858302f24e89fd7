"""Writes tests/golden/random/cases.npz (build container only: needs /root/reference).

Forty random cases in the style of tools/stress_parity.py -- signal mix, level over six decades, window size, channels,
hops, order mode, onset type / window / sensitivity, gain, sample rate -- analysed by the REFERENCE'S OWN hot-path headers
(tools/refdiff: compiled unmodified against the JUCE stand-in, log10(float) correctly rounded).  Only the seeds and the
expected outputs are stored: the hops are regenerated from the seed by the test and checked against a CRC, so the
fixture stays small.  The oracle must reproduce every value bit for bit (tests/test_oracle.py), the GPU path within
1e-5 / onset exact (tests/test_gpu_parity.py) on a box where /root/reference does not exist.

Same caveat as make_golden.py: PARITY STAYS UNPINNED at the JUCE boundary.

Run from the repo root:  python tests/golden/make_random_cases.py
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tools", "refdiff"))

import refdiff  # noqa: E402
from random_cases import CASE_SEED, NUM_CASES, draw_case  # noqa: E402  (tests/golden/random_cases.py: shared with the tests)
from oracle import fx_oracle as fo  # noqa: E402


def main():
    params, raws, sms, crcs = [], [], [], []
    for k in range(NUM_CASES):
        p, hops = draw_case(k)
        raw, sm = refdiff.run(hops, p["N"], order=p["order"], onset_type=p["onset_type"], onset_window=p["onset_window"],
                              onset_sensitivity=p["sensitivity"], gain=p["gain"], sample_rate=p["sample_rate"], mode="cr")
        oraw, osm = fo.batch_hops(hops, p["N"], sample_rate=p["sample_rate"], order=p["order"], gain=p["gain"], onset_type=p["onset_type"],
                                  onset_sensitivity=p["sensitivity"], onset_window=p["onset_window"])
        for a, b in ((raw, oraw), (sm, osm)):        # the oracle agrees bit for bit (NaN == NaN), or nothing is written
            assert ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all(), (k, p)
        raws.append(raw.ravel()); sms.append(sm.ravel()); crcs.append(zlib.crc32(hops.tobytes()))
        print(k, p, "->", raw.shape)
    os.makedirs(os.path.join(HERE, "random"), exist_ok=True)
    np.savez_compressed(os.path.join(HERE, "random", "cases.npz"), raw=np.concatenate(raws), smoothed=np.concatenate(sms),
                        hop_crc32=np.array(crcs, np.uint32), case_seed=CASE_SEED, num_cases=NUM_CASES,
                        source="reference headers (/root/reference/Source/{AudioDataCollector,RealTimeAudioAnalysis,PitchAnalyser,"
                               "SpectralCharacteristics,HarmonicCharacteristics,RealTimeAnalyser}.h, unmodified) compiled against "
                               "tools/refdiff/juce_standin.h, log10(float) correctly rounded; per-case settings from "
                               "tests/golden/random_cases.py draw_case(k)")


if __name__ == "__main__":
    main()
