"""The reference's own hot-path headers (compiled unmodified against tools/refdiff/juce_standin.h) against the CPU
oracle.  Build container only: skipped where /root/reference does not exist (the GPU box)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "refdiff"))

import refdiff  # noqa: E402
import signals  # noqa: E402

pytestmark = pytest.mark.skipif(not refdiff.available(), reason="/root/reference is not present on this machine")

CASES = [(sig, N) for sig in sorted(signals.ALL) for N in (256, 1024, 2048)] + [("tone", 4096), ("bursts", 4096), ("levels", 512)]


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _same_bits(a, b):
    """bit-identical, except that a NaN is a NaN whatever its sign / payload (0.0f / 0 has the sign of its compiler)"""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return (_bits(a) == _bits(b)) | (np.isnan(a) & np.isnan(b))


@pytest.mark.parametrize("sig,N", CASES)
def test_oracle_is_bit_identical_to_the_reference_headers(oracle, sig, N):
    """correctly rounded log10(float) on both sides: every raw and smoothed value, every frame"""
    C, T = 3, 14
    hops = signals.ALL[sig](C, T, N)
    raw, sm = refdiff.run(hops, N, mode="cr")
    oraw, osm = oracle.push_hops(hops, N)
    assert _same_bits(raw, oraw).all(), "raw differs at %s" % (np.argwhere(~_same_bits(raw, oraw))[:5],)
    assert _same_bits(sm, osm).all(), "smoothed differs at %s" % (np.argwhere(~_same_bits(sm, osm))[:5],)


@pytest.mark.parametrize("order", [0, 1, 2])
@pytest.mark.parametrize("otype,window,sens", [(0, 3, 0.3), (1, 5, 0.7), (2, 9, 0.2)])
def test_order_modes_and_onset_settings(oracle, order, otype, window, sens):
    N, C, T = 1024, 3, 40
    hops = signals.bursts(C, T, N, seed=50 + order)
    raw, sm = refdiff.run(hops, N, order=order, onset_type=otype, onset_window=window, onset_sensitivity=sens, gain=0.75, mode="cr")
    oraw, osm = oracle.push_hops(hops, N, order=order, onset_type=otype, onset_window=window, onset_sensitivity=sens, gain=0.75)
    assert _same_bits(raw, oraw).all() and _same_bits(sm, osm).all()


# (a block longer than (4097 - N/2) / 2 samples cannot be given to the reference single-threaded: with the writer less than a block behind the
# reader its getAnalysisBuffer spins on indexesOverlap until another thread moves the indices, AudioDataCollector.h:77,96-105)
@pytest.mark.parametrize("N,block", [(1024, 480), (1024, 441), (2048, 1), (4096, 1000), (512, 63), (1024, 1700)])
def test_the_reference_collector_fed_device_blocks_is_a_fifo_of_hops(oracle, N, block):
    """the reference's OWN AudioDataCollector (ref AudioDataCollector.h:36-94) fed blocks of any length through audioDeviceIOCallback, its
    analysers stepped whenever half a window is waiting: bit for bit the oracle fed the same stream as whole hops -- the equivalence
    fx_push_samples rests on"""
    C, T = 2, 10 if block > 1 else 4
    hops = signals.bursts(C, T, N, seed=N + block)
    raw, sm = refdiff.run_blocks(hops.reshape(C, -1), N, block, order=1)
    oraw, osm = oracle.push_hops(hops, N, order=1)
    assert _same_bits(raw, oraw).all() and _same_bits(sm, osm).all()


def test_committed_block_fixture_is_what_the_reference_collector_produces():
    """tests/golden/blocks/cases.npz regenerated from the headers (gain changes and clearBuffer between blocks included)"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from block_cases import CASES, stream_of
    g = np.load(os.path.join(ROOT, "tests", "golden", "blocks", "cases.npz"))
    for k, (name, N, C, hops, extra, block, order, events) in enumerate(CASES):
        raw, sm = refdiff.run_blocks(stream_of(name, N, C, hops, extra, seed=100 + k), N, block, order=order, events=events)
        assert _same_bits(raw, g[name + "_raw"]).all() and _same_bits(sm, g[name + "_smoothed"]).all(), name


def test_osc_message_is_the_reference_senders(fx, oracle):
    """The reference's OWN OSCFeatureAnalysisOutput (ref OSCFeatureAnalysisOutput.h:23-145, compiled unmodified; the stand-in's OSCSender records what it is
    handed): the twelve values and their order, the "ip[:port]" parsing with its default port and the 60 Hz timer -- against fx_pack_osc12 / fx_osc_encode,
    the oracle's message and the sink's target parsing."""
    import ctypes
    from importlib import import_module
    sharded = import_module("feature-extractor_amd.sharded")
    host, port, hz, addr, args = refdiff.osc_probe("192.168.1.20:7001")
    assert (host, port, hz, addr) == ("192.168.1.20", 7001, 60, "/Audio/A7") and len(args) == 12
    lib = fx.load_library()
    slots = (np.arange(12) + 100).astype(np.float32)
    wire = np.empty(12, np.float32)
    lib.fx_pack_osc12(slots.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), wire.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    assert wire.tolist() == args                                                 # onset, rms, f0, centroid, slope, spread, flatness, ler, flux, her, oer, inharm
    msg = oracle.osc_message(addr, slots)
    assert [float(v) for v in np.frombuffer(msg[-48:], ">f4")] == args
    assert refdiff.osc_probe("127.0.0.1")[:2] == ("127.0.0.1", 9000)             # the default port
    assert sharded.parse_osc_target("127.0.0.1") == ("127.0.0.1", 9000) and sharded.parse_osc_target("192.168.1.20:7001") == ("192.168.1.20", 7001)


def test_sample_rate_other_than_48k(oracle):
    N, C, T = 2048, 2, 12
    hops = signals.tone_vibrato_noise(C, T, N, seed=9)
    raw, sm = refdiff.run(hops, N, sample_rate=44100.0, mode="cr")
    oraw, osm = oracle.push_hops(hops, N, sample_rate=44100.0)
    assert _same_bits(raw, oraw).all() and _same_bits(sm, osm).all()


def test_platform_log10f_changes_no_decision(oracle):
    """With this platform's log10f in the reference headers (what a real build would call) the oracle's values stay
    within 1e-5 and no onset decision flips; the number of values that are not bit-identical is reported."""
    different = total = 0
    for sig in sorted(signals.ALL):
        for N in (1024, 2048):
            hops = signals.ALL[sig](3, 20, N)
            raw, sm = refdiff.run(hops, N, mode="libm")
            oraw, osm = oracle.push_hops(hops, N)
            signals.assert_features_close(raw, oraw, 1e-5, oracle.FEATURE_NAMES, "%s N=%d raw (libm log10f)" % (sig, N))
            signals.assert_features_close(sm, osm, 1e-5, oracle.FEATURE_NAMES, "%s N=%d smoothed (libm log10f)" % (sig, N))
            different += int((~_same_bits(raw, oraw)).sum() + (~_same_bits(sm, osm)).sum())
            total += raw.size + sm.size
    print("libm log10f: %d of %d values not bit-identical to the oracle" % (different, total))


def test_committed_fixtures_come_from_the_reference_headers():
    """tests/golden/*.npz were written by tests/golden/make_golden.py from this harness (cr mode): regenerate and compare"""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")))
    assert paths
    for p in paths:
        g = np.load(p)
        assert str(g["source"]).startswith("reference headers"), p
        raw, sm = refdiff.run(g["hops"], int(g["window_size"]), order=int(g["order"]), sample_rate=float(g["sample_rate"]), mode="cr")
        assert _same_bits(raw, g["raw"]).all() and _same_bits(sm, g["smoothed"]).all(), p


def test_legacy_offline_analyser_oracle_equals_the_reference_header():
    """oracle/fx_offline.c against the reference's LEGACY AudioAnalysis.h, compiled unmodified (tools/refdiff/refdiff_legacy.cpp):
    bit for bit on the seeded cases the committed fixture holds -- and the fixture itself is what that header produces now."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_offline_cases import offline_inputs
    from oracle import fx_oracle as fo
    if not refdiff.legacy_available():
        pytest.skip("reference sources not present")
    g, cases = np.load(os.path.join(ROOT, "tests", "golden", "offline", "cases.npz")), offline_inputs()
    audio, nd = cases["zc"][0]
    ref = refdiff.legacy_zero_crosses(audio, nd)
    assert ref.tobytes() == g["zc_0"].tobytes() == fo.offline_zero_crosses(audio, nd).tobytes()
    env, ns, ndn, sr = cases["lat"][2]
    assert np.float32(refdiff.legacy_log_attack_time(env, ns, ndn, sr)).tobytes() == g["lat_2"].tobytes() == fo.offline_log_attack_time(env, ns, ndn, sr).tobytes()
    cur, prev = cases["lbp"][0]
    rb, rh, ra = refdiff.legacy_fft_lbp(cur, prev)
    ob, oh, oa = fo.offline_fft_lbp(cur, prev)
    assert np.array_equal(rb, ob) and rh.tobytes() == oh.tobytes() and ra.tobytes() == oa.tobytes()
    mags, nyq = cases["hc"][0]
    ro, rp = refdiff.legacy_harmonic_characteristics(mags, nyq)
    assert ro.tobytes() == g["hc_0_out"].tobytes() and rp.tobytes() == g["hc_0_prev"].tobytes()
    # fresh random spectra beyond the fixture
    rng = np.random.default_rng(99)
    m2 = np.abs(rng.normal(0, 1.0, (4, 3, 513))).astype(np.float32)
    m2[:, :, 23::23] += 10.0
    ro, rp = refdiff.legacy_harmonic_characteristics(m2, 11025.0)
    pf = np.zeros(3)
    for t in range(4):
        assert fo.offline_harmonic_characteristics(m2[t], 11025.0, pf).tobytes() == ro[t].tobytes() and pf.tobytes() == rp[t].tobytes()
    # round 4: the legacy full-spectrum characteristics, slope and auto-correlation -- the fixture is what the header produces now, and
    # the oracle equals the header on fresh random frames (same glibc on both sides: pow() included)
    mags, nyq = cases["sc"][0]
    ro, rp = refdiff.legacy_spectral_characteristics(mags, nyq)
    assert ro.tobytes() == g["sc_0_out"].tobytes() and rp.tobytes() == g["sc_0_prev"].tobytes()
    assert refdiff.legacy_spectral_slope(cases["slope"][0]).tobytes() == g["slope_0"].tobytes()
    prod, freq = refdiff.legacy_auto_correlation(*cases["ac"][0])
    assert prod.tobytes() == g["ac_0_prod"].tobytes() and freq.tobytes() == g["ac_0_freq"].tobytes()
    m3 = np.abs(rng.normal(0, 1.0, (5, 4, 257))).astype(np.float32) * np.float32(3.0)
    m3[2, 1] = 1e-8
    ro, rp = refdiff.legacy_spectral_characteristics(m3, 8000.0)
    pb = np.zeros((4, 257))
    for t in range(5):
        assert fo.offline_spectral_characteristics(m3[t], 8000.0, pb).tobytes() == ro[t].tobytes(), t
    assert pb.tobytes() == rp.tobytes()
    assert fo.offline_spectral_slope(m3[0]).tobytes() == refdiff.legacy_spectral_slope(m3[0]).tobytes()
    d3 = rng.normal(0, 2.0, (3, 100, 2)).astype(np.float32)
    prod, freq = refdiff.legacy_auto_correlation(d3, 8000.0)
    assert fo.offline_conjugate_multiplication(d3).tobytes() == prod.tobytes() and fo.offline_auto_correlation(prod, 8000.0)[1].tobytes() == freq.tobytes()
