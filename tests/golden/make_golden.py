"""Writes tests/golden/*.npz.

PARITY UNPINNED: the reference cannot be built or run in this image (it needs JUCE 4.2.3, which is
absent), and it ships no golden vectors of its own, so these fixtures are outputs of
the CPU oracle (oracle/fx_oracle.c), NOT of the reference.  They pin the oracle against drift and
give the GPU tests committed data to compare with on a box where /root/reference does not exist.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import signals  # noqa: E402
from oracle import fx_oracle as fo  # noqa: E402

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
]


def main():
    for name, sig, N, C, T, order in CASES:
        hops = signals.ALL[sig](C, T, N)
        raw, sm = fo.push_hops(hops, N, order=order)
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
                            tap_frame=frame, tap_spectrum=spec, tap_lowpass=low, tap_cnd=cnd,
                            tap_lag=np.float32(lag), tap_f0=np.float64(f0))
        print(name, hops.shape, "->", raw.shape)


if __name__ == "__main__":
    main()
