"""The reference's legacy offline analyser (struct AudioAnalyser, ref AudioAnalysis.h): the oracle
(oracle/fx_offline.c) against vectors produced by the reference's own header (tests/golden/offline/cases.npz, made by
tests/golden/make_offline_cases.py from the unmodified AudioAnalysis.h) and -- with a GPU -- the HIP kernels behind
fx_offline_* against both, bit for bit: counts and bit patterns are integers, the float outputs are single roundings of
fp64 expressions evaluated in the reference's own order."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_offline_cases import offline_inputs  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "offline", "cases.npz")


def same(a, b):
    """bit for bit; a NaN equals a NaN whatever its sign and payload (pow of a negative number: glibc and the device differ there)"""
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    if a.tobytes() == b.tobytes():
        return True
    if a.dtype.kind != "f":
        return False
    bits = {4: np.uint32, 8: np.uint64}[a.dtype.itemsize]
    return bool(np.all((a.view(bits) == b.view(bits)) | (np.isnan(a) & np.isnan(b))))


def test_oracle_equals_the_reference_headers_vectors(oracle):
    g, cases = np.load(GOLDEN), offline_inputs()
    assert "AudioAnalysis.h" in str(g["source"])
    for k, (audio, nd) in enumerate(cases["zc"]):
        assert same(oracle.offline_zero_crosses(audio, nd), g["zc_%d" % k]), k
    for k, (env, ns, nd, sr) in enumerate(cases["lat"]):
        assert same(oracle.offline_log_attack_time(env, ns, nd, sr), g["lat_%d" % k]), k
    assert np.isneginf(g["lat_1"])                                  # the maximum in the first step: log10 (0), as the reference returns it
    for k, (cur, prev) in enumerate(cases["lbp"]):
        bits, hi, act = oracle.offline_fft_lbp(cur, prev)
        assert same(bits, g["lbp_%d_bits" % k]) and same(hi, g["lbp_%d_hi" % k]) and same(act, g["lbp_%d_act" % k]), k
    for k, (mags, nyq) in enumerate(cases["hc"]):
        pf = np.zeros(mags.shape[1])
        for t in range(mags.shape[0]):
            out = oracle.offline_harmonic_characteristics(mags[t], nyq, pf)
            assert same(out, g["hc_%d_out" % k][t]) and same(pf, g["hc_%d_prev" % k][t]), (k, t)
    # the cases exercise what they are meant to: the octave rule keeps the previous F0 (frame 2, channel 0), silence returns zeros
    o = g["hc_0_out"]
    assert o[2, 0, 0] == o[1, 0, 0] and o[3, 0, 0] == o[1, 0, 0] and not o[:, 3].any() and (o[:, :3, 0] > 0).all()
    # round 4: the legacy full-spectrum characteristics, slope and auto-correlation
    for k, (mags, nyq) in enumerate(cases["sc"]):
        pb = np.zeros(mags.shape[1:], np.float64)
        for t in range(mags.shape[0]):
            assert same(oracle.offline_spectral_characteristics(mags[t], nyq, pb), g["sc_%d_out" % k][t]), (k, t)
        assert same(pb, g["sc_%d_prev" % k]), k
    o = g["sc_0_out"]
    assert not o[2, 2].any() and o[3, 2].any()                      # the frame under the gate returns zeros ...
    assert np.isinf(o[:, 3, 2]).all() and not o[:, 4, 2].any()      # the product overflows to inf / underflows to 0, as in the reference
    for k, mags in enumerate(cases["slope"]):
        assert same(oracle.offline_spectral_slope(mags), g["slope_%d" % k]), k
    s = g["slope_0"]
    assert np.isnan(s[2]) and s[3] == 0.0 and s[1] > s[0] > 0        # flat: energy variance 0 -> NaN; under 0.0001: 0 (as the reference returns them)
    for k, (data, nyq) in enumerate(cases["ac"]):
        prod = oracle.offline_conjugate_multiplication(data)
        assert same(prod, g["ac_%d_prod" % k]), k
        peaks, freqs = oracle.offline_auto_correlation(prod, nyq)
        assert same(freqs, g["ac_%d_freq" % k]), k
    assert g["ac_0_freq"][0] == 40 * (24000.0 / 512) + 24000.0 / 1024 and g["ac_0_freq"][1] == 24000.0 / 1024      # the first of a tie; item 0


@pytest.mark.gpu
def test_offline_kernels_equal_the_reference_headers_vectors(gpu_fx, oracle):
    g, cases = np.load(GOLDEN), offline_inputs()
    for k, (audio, nd) in enumerate(cases["zc"]):
        an = gpu_fx.offline.AudioAnalyser(audio.shape[0])
        assert same(an.analyse_normalised_zero_crosses(audio, nd), g["zc_%d" % k]), k
    an = gpu_fx.offline.AudioAnalyser(1)
    for k, (env, ns, nd, sr) in enumerate(cases["lat"]):
        assert same(an.set_log_attack_time(env, ns, nd, sr), g["lat_%d" % k]), k
    for k, (cur, prev) in enumerate(cases["lbp"]):
        bits, hi, act = gpu_fx.offline.AudioAnalyser(cur.shape[0]).calculate_fft_lbp(cur, prev)
        assert same(bits, g["lbp_%d_bits" % k]) and same(hi, g["lbp_%d_hi" % k]) and same(act, g["lbp_%d_act" % k]), k
    for k, (mags, nyq) in enumerate(cases["hc"]):
        an = gpu_fx.offline.AudioAnalyser(mags.shape[1], nyq)
        for t in range(mags.shape[0]):
            out = an.calculate_harmonic_characteristics(mags[t])
            assert same(out, g["hc_%d_out" % k][t]), (k, t, out, g["hc_%d_out" % k][t])
            assert same(an.previous_f0, g["hc_%d_prev" % k][t]), (k, t)
        an.reset()
        assert not an.previous_f0.any()


@pytest.mark.gpu
def test_offline_round4_kernels_equal_the_reference_headers_vectors(gpu_fx, oracle):
    """The legacy full-spectrum characteristics (AudioAnalysis.h:463-515), slope (:566-609) and auto-correlation peak (:623-665) on the GPU
    against what the reference's own header produced, bit for bit, and against the oracle on shapes beyond the fixture (bit for bit
    where no libm function is involved; flatness and spread go through pow(), which the device and glibc round independently)."""
    g, cases = np.load(GOLDEN), offline_inputs()
    for k, (mags, nyq) in enumerate(cases["sc"]):
        an = gpu_fx.offline.AudioAnalyser(mags.shape[1], nyq)
        for t in range(mags.shape[0]):
            got = an.calculate_spectral_characteristics(mags[t])
            assert same(got, g["sc_%d_out" % k][t]), (k, t, got, g["sc_%d_out" % k][t])
        assert same(an.previous_bin_magnitudes, g["sc_%d_prev" % k]), k
        with pytest.raises(gpu_fx.FxError):
            an.calculate_spectral_characteristics(mags[0][:, :-1])             # previousBinMagnitudes has the analyser's size
        an.reset()
        assert same(an.calculate_spectral_characteristics(mags[0]), g["sc_%d_out" % k][0])
    for k, mags in enumerate(cases["slope"]):
        assert same(gpu_fx.offline.AudioAnalyser(mags.shape[0]).calculate_normalised_spectral_slope(mags), g["slope_%d" % k]), k
    for k, (data, nyq) in enumerate(cases["ac"]):
        prod, peaks, freqs = gpu_fx.offline.AudioAnalyser(data.shape[0], nyq).analyse_auto_correlation(data)
        assert same(prod, g["ac_%d_prod" % k]) and same(freqs, g["ac_%d_freq" % k]), k
    rng = np.random.default_rng(11)
    for C, B, nyq in ((9, 4097, 24000.0), (3, 1, 100.0), (40, 1025, 22050.0)):
        an = gpu_fx.offline.AudioAnalyser(C, nyq)
        pb = np.zeros((C, B))
        for t in range(3):
            mags = np.abs(rng.normal(0, 1.0, (C, B))).astype(np.float32) * (10.0 ** rng.integers(-3, 3))
            got, want = an.calculate_spectral_characteristics(mags), oracle.offline_spectral_characteristics(mags, nyq, pb)
            assert same(got[:, [0, 3]], want[:, [0, 3]]), (C, B, t)                                # centroid, flux: sums only
            np.testing.assert_allclose(got[:, [1, 2]], want[:, [1, 2]], rtol=2e-7, atol=0)            # spread, flatness: pow()
            assert same(an.previous_bin_magnitudes, pb)
            assert same(an.calculate_normalised_spectral_slope(mags), oracle.offline_spectral_slope(mags)), (C, B, t)
        data = rng.normal(0, 1, (C, max(B // 2, 1), 2)).astype(np.float32)
        prod, peaks, freqs = an.analyse_auto_correlation(data)
        want_prod = oracle.offline_conjugate_multiplication(data)
        wp, wf = oracle.offline_auto_correlation(want_prod, nyq)
        assert same(prod, want_prod) and same(peaks, wp) and same(freqs, wf), (C, B)
        # NaNs: getMaxIndex (ref AudioAnalysis.h:651-665) skips them -- `data[i] > currentMax` is false -- unless one sits at item 0, which
        # then stays the maximum; a NaN early in a thread's stride must not hide the true maximum behind it
        if data.shape[1] > 600:
            data[:, 3, 0] = np.nan                   # item 3; the maximum is put 256 items further on (the same thread's stride)
            data[:, 3 + 256, 0] = 50.0
            data[0, 0, 0] = np.nan                   # channel 0: NaN at item 0
            prod, peaks, freqs = an.analyse_auto_correlation(data)
            want_prod = oracle.offline_conjugate_multiplication(data)
            wp, wf = oracle.offline_auto_correlation(want_prod, nyq)
            assert same(peaks, wp) and same(freqs, wf), (C, B, peaks, wp)
            assert peaks[0] == 0 and (peaks[1:] == 3 + 256).all()


@pytest.mark.gpu
def test_offline_serial_sums_at_every_batch_boundary_and_ties_across_waves(gpu_fx, oracle):
    """Round 6: each serial sum runs on a lane of its own, eight loads ahead of the additions, and the maxima are butterflies.  Bin counts on
    both sides of every batch boundary (so that the first batch, the pipelined middle and the tail are each empty once), signed magnitudes
    (the sums are not monotone), and equal maxima placed in different lanes and different waves (the first one must win, +0.0 == -0.0)."""
    rng = np.random.default_rng(61)
    for B in (2, 7, 8, 9, 15, 16, 17, 23, 24, 25, 31, 32, 33, 255, 256, 257):
        C, nyq = 5, 24000.0
        an = gpu_fx.offline.AudioAnalyser(C, nyq)
        pb = np.zeros((C, B))
        for t in range(3):
            mags = rng.normal(0, 1.0, (C, B)).astype(np.float32) * np.float32(10.0 ** rng.integers(-6, 6))
            if t < 2:
                mags = np.abs(mags)                 # (signed magnitudes once: the centroid can leave [0, nyquist] then, pow() of a negative base is the reference's business)
            got, want = an.calculate_spectral_characteristics(mags), oracle.offline_spectral_characteristics(mags, nyq, pb)
            assert same(got[:, [0, 3]], want[:, [0, 3]]), (B, t)
            if t < 2:
                np.testing.assert_allclose(got[:, [1, 2]], want[:, [1, 2]], rtol=2e-7, atol=0)
            assert same(an.previous_bin_magnitudes, pb)
            assert same(an.calculate_normalised_spectral_slope(mags), oracle.offline_spectral_slope(mags)), (B, t)
            if B >= 4:
                fresh = gpu_fx.offline.AudioAnalyser(C, nyq)
                assert same(fresh.calculate_harmonic_characteristics(np.abs(mags)), oracle.offline_harmonic_characteristics(np.abs(mags), nyq, np.zeros(C))), (B, t)
    # auto-correlation: the same largest real part at items that fall to different lanes (i, i + 1), different waves (i, i + 64) and a later
    # stride of the same thread (i, i + 256): the first wins; all items zero with mixed signs: item 0
    for n in (3, 64, 65, 300, 1024):
        C = 6
        data = rng.normal(0, 0.1, (C, n, 2)).astype(np.float32)
        data[:, :, 1] = 0.0                         # real items: the product's real part is the square
        for c, (i, j) in enumerate(((1, 2), (1, 65), (2, 258), (70, 71), (0, n - 1))):
            if j < n:
                data[c, i, 0] = 3.0; data[c, j, 0] = -3.0           # equal squares
        data[5] = 0.0
        data[5, ::2, 1] = -0.0
        an = gpu_fx.offline.AudioAnalyser(C, 1000.0)
        prod, peaks, freqs = an.analyse_auto_correlation(data)
        want_prod = oracle.offline_conjugate_multiplication(data)
        wp, wf = oracle.offline_auto_correlation(want_prod, 1000.0)
        assert same(prod, want_prod) and same(peaks, wp) and same(freqs, wf), (n, peaks, wp)
    # FFT-LBP counts / highest bin through the butterflies
    for C, B in ((3, 63), (3, 64), (3, 65), (2, 300)):
        cur = np.abs(rng.normal(0, 0.3, (C, B))).astype(np.float32)
        prev = np.abs(rng.normal(0, 0.3, (C, B))).astype(np.float32)
        assert all(same(a, b) for a, b in zip(gpu_fx.offline.AudioAnalyser(C).calculate_fft_lbp(cur, prev), oracle.offline_fft_lbp(cur, prev))), (C, B)


@pytest.mark.gpu
def test_offline_kernels_equal_the_oracle_on_random_inputs(gpu_fx, oracle):
    """Sizes and contents beyond the fixture: many channels, windows up to 8192 points (4097 magnitudes), spectra with thousands
    of peaks (a histogram of ~10^6 pairs), ties in the weighted counts, device-resident buffers."""
    import torch
    rng = np.random.default_rng(7)
    for C, S, nd in ((3, 48000, 100), (40, 4097, 4096), (2, 100000, 3)):
        audio = rng.normal(0, 1, (C, S)).astype(np.float32)
        audio[:, ::7] = 0.0
        assert same(gpu_fx.offline.AudioAnalyser(C).analyse_normalised_zero_crosses(audio, nd), oracle.offline_zero_crosses(audio, nd)), (C, S, nd)
    for n in (1, 5, 1000):
        env = np.abs(rng.normal(0, 1, n)).astype(np.float32)
        assert same(gpu_fx.offline.AudioAnalyser(1).set_log_attack_time(env, 44100 * 3, max(1, n), 44100), oracle.offline_log_attack_time(env, 44100 * 3, max(1, n), 44100))
    for C, B in ((7, 4097), (1, 5), (33, 1025)):
        cur = np.abs(rng.normal(0, 0.3, (C, B))).astype(np.float32)
        prev = np.abs(rng.normal(0, 0.3, (C, B))).astype(np.float32)
        got, want = gpu_fx.offline.AudioAnalyser(C).calculate_fft_lbp(cur, prev), oracle.offline_fft_lbp(cur, prev)
        assert all(same(a, b) for a, b in zip(got, want)), (C, B)
    for C, B, nyq in ((12, 4097, 24000.0), (5, 2049, 22050.0), (3, 513, 8000.0), (2, 4, 100.0)):
        an = gpu_fx.offline.AudioAnalyser(C, nyq)
        pf = np.zeros(C)
        for t in range(5):
            mags = np.abs(rng.normal(0, 1.0, (C, B))).astype(np.float32)
            if B > 100:
                mags[:, (17 + 3 * t)::(17 + 3 * t)] += 12.0                # a harmonic comb whose spacing moves from frame to frame
                mags[0, :] = np.round(mags[0, :] * 4) / 4                  # quantised magnitudes: ties between candidates
            if t == 3:
                mags[1] = 1e-8                                             # below the 0.001 gate: zeros, previousF0 untouched
            got = an.calculate_harmonic_characteristics(mags)
            want = oracle.offline_harmonic_characteristics(mags, nyq, pf)
            assert same(got, want), (C, B, t, got, want)
            assert same(an.previous_f0, pf), (C, B, t)
    # device-resident buffers through the C ABI
    C, B = 4, 1025
    mags = np.abs(rng.normal(0, 1.0, (C, B))).astype(np.float32)
    mags[:, 40::40] += 9.0
    an = gpu_fx.offline.AudioAnalyser(C, 24000.0)
    d_in, d_out = torch.from_numpy(mags).cuda(), torch.zeros((C, 3), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    import ctypes
    gpu_fx.capi.check(an._lib.fx_offline_harmonic_characteristics(an._h, ctypes.c_void_p(d_in.data_ptr()), B, ctypes.c_void_p(d_out.data_ptr()), gpu_fx.capi.MEM_DEVICE))
    gpu_fx.capi.check(an._lib.fx_offline_sync(an._h))
    assert same(d_out.cpu().numpy(), oracle.offline_harmonic_characteristics(mags, 24000.0, np.zeros(C)))
