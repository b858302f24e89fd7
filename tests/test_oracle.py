"""CPU tests of the oracle (oracle/fx_oracle.c): mathematics, the second numpy restatement,
behaviours SURVEY.md 8(c) observed from the reference headers, and the committed fixtures.
The reference has no tests or golden vectors of its own, so none can be checked here
("parity unpinned", see oracle/fx_oracle.h)."""
import glob
import os

import numpy as np
import pytest

import signals
from oracle import fx_numpy as fn
from oracle import fx_oracle as fo

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.mark.parametrize("n", [256, 512, 1024, 2048, 4096])
def test_fft_matches_fp64_dft(n):
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    X = np.fft.fft(x.astype(np.complex128))
    assert np.abs(fo.fft_complex(x) - X).max() <= 3e-7 * np.abs(X).max()
    xi = np.fft.ifft(x.astype(np.complex128)) * n          # JUCE perform() is un-normalised both ways
    assert np.abs(fo.fft_complex(x, inverse=True) - xi).max() <= 3e-7 * np.abs(xi).max()


@pytest.mark.parametrize("n", [256, 512, 1024, 2048])
def test_fft_bit_identical_to_second_restatement(n):
    rng = np.random.default_rng(7 * n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    for inv in (False, True):
        a = fo.fft_complex(x, inv)
        r, i = fn.fft(x.real.copy(), x.imag.copy(), inv)
        assert np.array_equal(a.real, r) and np.array_equal(a.imag, i)


def test_forward_real_layout():
    x = np.random.default_rng(0).standard_normal(1024).astype(np.float32)
    spec = fo.forward_real(x)
    assert spec.shape == (2048,)
    X = np.fft.fft(x.astype(np.float64))
    assert np.abs(spec[0::2] - X.real).max() < 1e-4 and np.abs(spec[1::2] - X.imag).max() < 1e-4


def test_bartlett_is_the_asymmetric_triangle():
    w = fo.bartlett(np.ones(1024, np.float32))
    i = np.arange(1024)
    expect = np.where(i < 512, 2 * i / 1024, 2 - 2 * i / 1024).astype(np.float32)
    assert np.array_equal(w, expect)
    assert w[0] == 0 and w[512] == 1 and w[1023] == np.float32(2 / 1024)


def test_lowpass_constants_and_recurrence():
    a, b = fo.lpf_constants()
    assert a == float(np.float32(np.pi) / np.float32(2))
    assert abs(b - np.exp(-np.pi / 2)) < 1e-7
    x = np.random.default_rng(3).standard_normal(300).astype(np.float32)
    assert np.array_equal(fo.lowpass(x), fn.lowpass(x))


def test_silence_gives_f0_4p8_and_zeros():
    # silence -> f0 feature 4.8 (lag 2 -> 24 kHz / 5000), everything else 0
    raw, sm = fo.Channel(2048).push_hops(signals.silence(1, 8, 2048)[0])
    expect = np.zeros(12, np.float32)
    expect[fo.F0] = np.float32(24000.0 / 5000.0)
    assert np.array_equal(raw[-1], expect)
    assert np.allclose(sm[-1], expect, rtol=1e-6)      # float sum of eight 4.8f, divided by 8


def test_loud_noise_overflows_flatness_and_exceeds_unit_range():
    # sigma = 1 noise -> flatness = inf, slope > 1, centroid > 1
    raw, _ = fo.Channel(2048).push_hops(signals.loud_noise(1, 8, 2048)[0])
    assert np.isinf(raw[-1, fo.FLATNESS])
    assert raw[:, fo.CENTROID].max() > 0.97       # log-compressed centroid is not bounded by 1 (up to 1.279)
    assert raw[:, fo.SLOPE].max() > 1.0


def test_bench_like_tone_underflows_flatness_to_zero():
    # flatness = 0 observed at N = 2048 for a 0.4-amplitude tone (product underflow)
    import importlib
    fx = importlib.import_module("feature-extractor_amd")
    raw, _ = fo.Channel(2048).push_hops(fx.synth.hops(1, 6, 2048)[0])
    assert raw[-1, fo.FLATNESS] == 0.0


def test_impulse_triggers_an_amplitude_onset_one_hop_later():
    # single impulse -> onset = 1 one hop after the impulse (Amplitude type)
    hops = signals.impulse(1, 16, 1024, at_hop=6)
    raw, _ = fo.Channel(1024).push_hops(hops[0])
    assert raw[:, fo.ONSET].sum() >= 1
    assert raw[7, fo.ONSET] == 1.0 and raw[6, fo.ONSET] == 0.0


def test_dc_input_has_a_slightly_negative_slope():
    """a DC 0.5 input run through the real headers gave a slope just below zero, all energy in the
    low fifth of the spectrum and rms = log10(0.5 * 9 + 1)."""
    hops = np.full((1, 12, 1024), 0.5, np.float32)
    raw, _ = fo.push_hops(hops, 2048)
    last = raw[0, -1]
    assert -1e-3 < last[fo.SLOPE] < 0.0
    assert last[fo.LER] == 1.0
    assert last[fo.RMS] == np.float32(np.log10(np.float32(0.5) * np.float32(9.0) + np.float32(1.0)))
    assert last[fo.ONSET] == 0.0 and last[fo.FLUX] == 0.0


def test_oer_slot_is_a_copy_of_her():
    raw, sm = fo.Channel(1024).push_hops(signals.tone_vibrato_noise(1, 12, 1024)[0])
    assert np.array_equal(raw[:, fo.OER], raw[:, fo.HER])
    assert np.array_equal(sm[:, fo.OER], sm[:, fo.HER])


def test_hops_and_preassembled_frames_agree():
    N, T = 1024, 10
    hops = signals.tone_vibrato_noise(1, T, N)[0]
    stream = np.concatenate([np.zeros(N // 2, np.float32), hops.reshape(-1)])
    frames = np.stack([stream[t * N // 2: t * N // 2 + N] for t in range(T)])
    a = fo.Channel(N).push_hops(hops)
    b = fo.Channel(N).process_frames(frames)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_gain_is_applied_to_hops():
    N = 1024
    hops = signals.tone_vibrato_noise(1, 6, N)[0]
    ch = fo.Channel(N)
    ch.set_gain(0.5)
    a = ch.push_hops(hops)
    b = fo.Channel(N).push_hops((hops * np.float32(0.5)).astype(np.float32))
    assert np.array_equal(a[0], b[0])


def test_order_modes_differ_only_in_smoothed_rms_and_onset():
    N = 1024
    hops = signals.bursts(1, 24, N)[0]
    r0, s0 = fo.Channel(N, order=fo.ORDER_SPECTRAL_THEN_HARMONIC).push_hops(hops)
    r1, s1 = fo.Channel(N, order=fo.ORDER_HARMONIC_THEN_SPECTRAL).push_hops(hops)
    r2, s2 = fo.Channel(N, order=fo.ORDER_ISOLATED).push_hops(hops)
    for r in (r1, r2):
        assert np.array_equal(r[:, 1:], r0[:, 1:], equal_nan=True)     # raw values except onset
    cols = [c for c in range(12) if c not in (fo.ONSET, fo.RMS)]
    assert np.array_equal(s0[:, cols], s1[:, cols], equal_nan=True)
    assert np.array_equal(s0[:, fo.RMS], s1[:, fo.RMS])                    # both orders end a hop with both inserts
    assert not np.array_equal(s0[:, fo.RMS], s2[:, fo.RMS])               # 5-hop vs 10-hop mean


def test_onset_window_reset_blocks_detection_until_refilled():
    N = 1024
    hops = signals.bursts(1, 30, N, seed=11)[0]
    ch = fo.Channel(N)
    ch.push_hops(hops[:20])
    ch.set_onset_window(7)
    raw, _ = ch.push_hops(hops[20:])
    assert not raw[:6, fo.ONSET].any()           # histories not full yet (ref SpectralCharacteristics.h:256-258)


@pytest.mark.parametrize("n", [256, 1024])
def test_c_oracle_agrees_with_numpy_restatement(n):
    hops = signals.bursts(1, 8, n, seed=n)[0]
    stream = np.concatenate([np.zeros(n // 2, np.float32), hops.reshape(-1)])
    frames = np.stack([stream[t * n // 2: t * n // 2 + n] for t in range(8)])
    want = fn.raw_features(frames)
    got, _ = fo.Channel(n).process_frames(frames)
    got = got.copy()
    got[:, 0] = 0
    signals.assert_features_close(got, want, rtol=1e-6, names=fo.FEATURE_NAMES, what="C oracle vs numpy restatement")


def test_levels_walk_flatness_through_zero_finite_and_inf():
    raw, _ = fo.push_hops(signals.levels(8, 8, 1024), 1024)
    flat = raw[:, -1, fo.FLATNESS]
    assert (flat == 0).any() and np.isinf(flat).any() and ((flat > 0) & np.isfinite(flat)).any()


def test_osc_message_bytes():
    v = np.arange(12, dtype=np.float32) / 8
    msg = fo.osc_message("/Audio/A0", v)
    assert len(msg) == 76
    assert msg[:12] == b"/Audio/A0\0\0\0" and msg[12:28] == b",ffffffffffff\0\0\0"
    wire = np.frombuffer(msg[28:], ">f4")
    order = [0, 1, 2, 3, 8, 4, 5, 6, 7, 9, 10, 11]     # onset,rms,f0,centroid,slope,spread,flatness,ler,flux,her,oer,inharm
    assert np.array_equal(wire, v[order])


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_committed_fixtures(path):
    g = np.load(path)
    raw, sm = fo.push_hops(g["hops"], int(g["window_size"]), float(g["sample_rate"]), order=int(g["order"]))
    assert np.array_equal(raw, g["raw"], equal_nan=True)
    assert np.array_equal(sm, g["smoothed"], equal_nan=True)
    # taps of channel 0's last frame
    frame = g["tap_frame"]
    assert np.array_equal(fo.forward_real(fo.bartlett(frame)), g["tap_spectrum"])
    low = fo.lowpass(frame)
    assert np.array_equal(low, g["tap_lowpass"])
    f0, lag, cnd = fo.estimate_pitch(fo.forward_real(fo.bartlett(low)), float(g["sample_rate"]) / 2)
    assert np.array_equal(cnd, g["tap_cnd"], equal_nan=True) and lag == float(g["tap_lag"])
    assert f0 == float(g["tap_f0"]) or (np.isnan(f0) and np.isnan(float(g["tap_f0"])))
    # the analyser's own F0 slot is that estimate / 5000 (ref RealTimeAnalyser.h:166) when the harmonic analyser ran last
    T = g["hops"].shape[1]
    if T >= 2:
        assert raw[0, T - 1, fo.F0] == np.float32(f0 / 5000.0)


def test_threaded_hop_batch_equals_per_channel_calls():
    hops = (np.random.default_rng(5).standard_normal((7, 9, 256)) * 0.2).astype(np.float32)
    kw = dict(order=1, gain=0.5, onset_type=fo.ONSET_AMPLITUDE, onset_sensitivity=0.7, onset_window=4, analysers=3)
    a = fo.push_hops(hops, 512, **kw)
    b = fo.batch_hops(hops, 512, threads=3, **kw)
    assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1], equal_nan=True)


def _random_reference_cases():
    import sys
    import zlib
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from random_cases import NUM_CASES, draw_case
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "random", "cases.npz"))
    assert int(fx["num_cases"]) == NUM_CASES
    at = 0
    for k in range(NUM_CASES):
        p, hops = draw_case(k)
        n = p["C"] * p["T"] * 12
        if zlib.crc32(hops.tobytes()) != int(fx["hop_crc32"][k]):
            pytest.fail("case %d: this platform regenerates other samples than the fixture was made from (numpy / libm)" % k)
        yield k, p, hops, fx["raw"][at:at + n].reshape(p["C"], p["T"], 12), fx["smoothed"][at:at + n].reshape(p["C"], p["T"], 12)
        at += n


def test_oracle_reproduces_random_reference_cases():
    """Forty random cases (signal mix, level, window size, order mode, onset settings, gain, sample rate) whose expected
    outputs come from the reference's own headers (tests/golden/make_random_cases.py): bit for bit."""
    for k, p, hops, raw, sm in _random_reference_cases():
        oraw, osm = fo.batch_hops(hops, p["N"], sample_rate=p["sample_rate"], order=p["order"], gain=p["gain"], onset_type=p["onset_type"],
                                  onset_sensitivity=p["sensitivity"], onset_window=p["onset_window"])
        assert np.array_equal(oraw, raw, equal_nan=True) and np.array_equal(osm, sm, equal_nan=True), (k, p)
