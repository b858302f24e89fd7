"""Build container only: expected outputs of the reference's LEGACY offline analyser (AudioAnalysis.h, compiled unmodified
against tools/refdiff/juce_standin.h) on seeded inputs -> tests/golden/offline/cases.npz.  Inputs are regenerated from the
seeds by offline_inputs() (tests/test_offline.py imports it), so the fixture holds expected outputs only.

    python tests/golden/make_offline_cases.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "refdiff"))


def offline_inputs():
    """Seeded inputs for the functions: {name: [case, ...]}."""
    rng = np.random.default_rng(20260403)
    cases = {}
    # zero crossings: noise, a sine with exact zeros, silence, DC, a ragged length
    n = np.arange(6000)
    audio = np.stack([rng.normal(0, 0.3, 6000), np.sin(2 * np.pi * n / 8.0), np.zeros(6000), np.full(6000, 0.25),
                      np.where(n % 3 == 0, 0.0, rng.normal(0, 1, 6000)), -np.abs(rng.normal(0, 1, 6000))]).astype(np.float32)
    cases["zc"] = [(audio, 7), (audio[:, :4096], 64), (audio[:, :1000], 1), (audio[:, :5], 2)]
    # log attack time: (envelope, input samples, downsamples, sample rate)
    env = np.abs(rng.normal(0, 1, 300)).astype(np.float32)
    env2 = env.copy(); env2[0] = 9.0                           # maximum first: log10 (0)
    env3 = env.copy(); env3[77] = env3[200] = 5.0              # the first of two equal maxima
    cases["lat"] = [(env, 300 * 441, 300, 44100), (env2, 300 * 441, 300, 44100), (env3, 48000, 300, 48000), (env[:10], 12345, 10, 22050)]
    # FFT-LBP: (cur, prev) magnitude frames
    cur = np.abs(rng.normal(0, 0.2, (5, 1025))).astype(np.float32)
    prev = (cur + rng.normal(0, 0.08, cur.shape)).astype(np.float32)
    prev[3] = cur[3]                                           # nothing over the threshold
    cases["lbp"] = [(cur, prev), (cur[:, :513], prev[:, :513] * 0.0)]
    # histogram F0: T frames x C channels x B magnitudes; harmonic combs (so that intervals repeat), noise, near-silence, a frame
    # that makes previousF0's octave rule fire
    def comb(B, spacing, level, decay=0.97):
        m = np.abs(rng.normal(0, 0.02, B))
        k = np.arange(spacing, B, spacing)
        m[k] += level * decay ** np.arange(k.size)
        return m
    frames = []
    for t, (sp0, sp1) in enumerate([(20, 33), (20, 33), (61, 33), (20, 11), (20, 33), (7, 5)]):
        frames.append(np.stack([comb(1025, sp0, 8.0), comb(1025, sp1, 3.0), np.abs(rng.normal(0, 1.0, 1025)),
                                np.full(1025, 1e-7), comb(1025, sp0 * 3 if t % 2 else sp0, 5.0, 0.9)]))
    cases["hc"] = [(np.stack(frames).astype(np.float32), 24000.0), (np.stack(frames)[:, :, :513].astype(np.float32), 22050.0)]
    # (round 4) the legacy full-spectrum characteristics: T frames x C channels x B magnitudes through one analyser per channel -- a comb, noise,
    # a frame under the 0.001 gate in the middle (previousBinMagnitudes must survive it), loud magnitudes (the product overflows to inf),
    # tiny ones (it underflows to 0), and a channel with a negative magnitude (the function takes whatever the buffer holds)
    rng2 = np.random.default_rng(20260404)
    sc = []
    for t in range(5):
        loud = np.abs(rng2.normal(0, 1, 513)) * 1e3 + 10.0
        tiny = np.abs(rng2.normal(0, 1, 513)) * 1e-3 + 1e-5
        quiet = np.full(513, 1e-7) if t == 2 else np.abs(rng2.normal(0, 0.5, 513))
        neg = np.abs(rng2.normal(0, 1, 513)); neg[100] = -0.5
        sc.append(np.stack([comb(513, 16 + t, 6.0), np.abs(rng2.normal(0, 1.0, 513)), quiet, loud, tiny, neg]))
    cases["sc"] = [(np.stack(sc).astype(np.float32), 24000.0), (np.stack(sc)[:3, :2, :65].astype(np.float32), 11025.0)]
    # the legacy slope: decaying, rising, flat (energy variance 0: NaN), silent (under 0.0001: 0), one with a negative peak
    k = np.arange(1025)
    falling, rising = np.exp(-k / 200.0), k / 1024.0
    negpeak = np.abs(rng2.normal(0, 0.1, 1025)); negpeak[7] = -3.0
    cases["slope"] = [np.stack([falling, rising, np.full(1025, 0.3), np.full(1025, 1e-5), negpeak, np.abs(rng2.normal(0, 1, 1025))]).astype(np.float32),
                      np.abs(rng2.normal(0, 1, (3, 17))).astype(np.float32)]
    # auto-correlation: complex items; a tie for the largest product (the first wins), the maximum at item 0, ordinary noise
    ac = rng2.normal(0, 1, (4, 512, 2)).astype(np.float32)
    ac[0, 40] = [3.0, 4.0]; ac[0, 300] = [5.0, 0.0]            # both products are 25
    ac[1, 0] = [9.0, 9.0]
    cases["ac"] = [(ac, 24000.0), (ac[:2, :16].copy(), 8000.0)]
    return cases


def main():
    import refdiff
    assert refdiff.legacy_available(), "needs /root/reference"
    cases = offline_inputs()
    out = {"source": "AudioAnalyser (ref AudioAnalysis.h) compiled unmodified against tools/refdiff/juce_standin.h; tools/refdiff/refdiff_legacy.cpp"}
    for k, (audio, nd) in enumerate(cases["zc"]):
        out["zc_%d" % k] = refdiff.legacy_zero_crosses(audio, nd)
    for k, (env, ns, nd, sr) in enumerate(cases["lat"]):
        out["lat_%d" % k] = np.float32(refdiff.legacy_log_attack_time(env, ns, nd, sr))
    for k, (cur, prev) in enumerate(cases["lbp"]):
        bits, hi, act = refdiff.legacy_fft_lbp(cur, prev)
        out["lbp_%d_bits" % k], out["lbp_%d_hi" % k], out["lbp_%d_act" % k] = bits, hi, act
    for k, (mags, nyq) in enumerate(cases["hc"]):
        o, pf = refdiff.legacy_harmonic_characteristics(mags, nyq)
        out["hc_%d_out" % k], out["hc_%d_prev" % k] = o, pf
    for k, (mags, nyq) in enumerate(cases["sc"]):
        o, pb = refdiff.legacy_spectral_characteristics(mags, nyq)
        out["sc_%d_out" % k], out["sc_%d_prev" % k] = o, pb
    for k, mags in enumerate(cases["slope"]):
        out["slope_%d" % k] = refdiff.legacy_spectral_slope(mags)
    for k, (data, nyq) in enumerate(cases["ac"]):
        prod, freq = refdiff.legacy_auto_correlation(data, nyq)
        out["ac_%d_prod" % k], out["ac_%d_freq" % k] = prod, freq
    path = os.path.join(ROOT, "tests", "golden", "offline", "cases.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
