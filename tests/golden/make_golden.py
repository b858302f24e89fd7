"""Writes tests/golden/*.npz (build container only: needs /root/reference).

`raw` and `smoothed` are produced by the REFERENCE'S OWN hot-path headers, compiled unmodified against
tools/refdiff/juce_standin.h and single-stepped by tools/refdiff/refdiff_driver.cpp (log10(float) correctly
rounded, the convention DESIGN.md section 5 documents); the `source` field of every file says so, together with
the order mode.  The oracle must reproduce them bit for bit (tests/test_oracle.py) and the GPU path within
1e-5 / onset exact (tests/test_gpu_parity.py) on a box where /root/reference does not exist.

PARITY STAYS UNPINNED at the JUCE boundary: the reference cannot be *built* here (JUCE 4.2.3 is absent) and
ships no vectors of its own; JUCE's FFT / getRMSLevel / applyGainRamp / getMagnitude are restated in the
stand-in from the same published algorithm as in the oracle.  The `tap_*` arrays are oracle intermediates.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "tools", "refdiff"))

import refdiff  # noqa: E402
import signals  # noqa: E402
from oracle import fx_oracle as fo  # noqa: E402

ORDER_NAMES = {0: "spectral analyser then harmonic analyser on one shared AudioFeatures",
               1: "harmonic analyser then spectral analyser on one shared AudioFeatures",
               2: "one AudioFeatures per analyser (isolated)"}

CASES = [
    # name, signal, N, C, T, order
    ("tone_1024", "tone", 1024, 3, 24, 0),
    ("tone_2048", "tone", 2048, 2, 16, 0),
    ("tone_4096", "tone", 4096, 2, 10, 0),
    ("bursts_1024_harmfirst", "bursts", 1024, 3, 24, 1),
    ("bursts_2048_isolated", "bursts", 2048, 2, 16, 2),
    ("levels_1024", "levels", 1024, 8, 8, 0),
    ("flat_edge_1024", "flat_edge", 1024, 12, 6, 0),
    ("impulse_1024", "impulse", 1024, 2, 16, 0),
    ("impulse_on_boundary_512", "impulse_on_boundary", 512, 3, 12, 0),
    ("silence_2048", "silence", 2048, 1, 8, 0),
    ("loud_noise_2048", "loud_noise", 2048, 2, 8, 0),
    ("sine_512", "sine", 512, 2, 12, 0),
    ("dc_256", "dc", 256, 1, 12, 0),
    # the lag search runs past the first few hundred lags (fundamentals of 9 .. 70 Hz, DC under a tone, drift)
    ("low_tones_1024", "low_tones", 1024, 6, 10, 0),
    ("low_tones_4096", "low_tones", 4096, 3, 8, 0),
]


def main():
    only = sys.argv[1:]                      # optional: names of the cases to (re)write
    for name, sig, N, C, T, order in CASES:
        if only and name not in only:
            continue
        hops = signals.ALL[sig](C, T, N)
        raw, sm = refdiff.run(hops, N, order=order, mode="cr")
        oraw, osm = fo.push_hops(hops, N, order=order)
        for a, b in ((raw, oraw), (sm, osm)):        # the oracle agrees bit for bit (NaN == NaN), or nothing is written
            assert ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all(), name
        # taps of the first channel's last frame, for debugging a failing port
        frame = np.concatenate([hops[0, T - 2], hops[0, T - 1]])
        spec = fo.forward_real(fo.bartlett(frame))
        # pitch path of the same frame (ref RealTimeAnalyser.h:152-160): low-pass, window, spectrum, then
        # PitchAnalyser's cumulative normalised difference and lag
        low = fo.lowpass(frame)
        pitch_spec = fo.forward_real(fo.bartlett(low))
        f0, lag, cnd = fo.estimate_pitch(pitch_spec, 24000.0)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), hops=hops, raw=raw, smoothed=sm,
                            window_size=N, order=order, sample_rate=48000.0,
                            source="reference headers (/root/reference/Source/{AudioDataCollector,RealTimeAudioAnalysis,PitchAnalyser,"
                                   "SpectralCharacteristics,HarmonicCharacteristics,RealTimeAnalyser}.h, unmodified) compiled against "
                                   "tools/refdiff/juce_standin.h, log10(float) correctly rounded; order mode %d: %s; onset: Amplitude, "
                                   "window 5, sensitivity 0.7 (the reference's defaults)" % (order, ORDER_NAMES[order]),
                            tap_frame=frame, tap_spectrum=spec, tap_lowpass=low, tap_cnd=cnd,
                            tap_lag=np.float32(lag), tap_f0=np.float64(f0))
        print(name, hops.shape, "->", raw.shape)


if __name__ == "__main__":
    main()
