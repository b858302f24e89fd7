"""fx_pair_kernel (windows of 2048 / 4096 points, one frame across two wavefronts) against the oracle and against the
one-wavefront-per-frame kernel: discrete decisions (onset, pitch lag) bit-exact, everything else within the parity bar."""
import numpy as np
import pytest

import signals

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def close(got, want, what):
    from oracle import fx_oracle as fo
    return signals.assert_features_close(got, want, RTOL, fo.FEATURE_NAMES, what)


def analysers(gpu_fx, C, N, **kw):
    pair, single = gpu_fx.BatchAnalyser(C, N, **kw), gpu_fx.BatchAnalyser(C, N, **kw)
    pair.set_tuning(waves_per_frame=2)
    single.set_tuning(waves_per_frame=1)
    return pair, single


def same_decisions(got, want, what):
    """onset and f0 (the integer lag) are discrete: they must agree exactly between the two kernels; NaN / inf patterns too"""
    for k in (0, 1):
        assert np.array_equal(got[k][:, :, 0], want[k][:, :, 0]), what + ": onset"
        assert np.array_equal(got[k][:, :, 2], want[k][:, :, 2], equal_nan=True), what + ": f0"
        assert np.array_equal(np.isnan(got[k]), np.isnan(want[k])) and np.array_equal(np.isinf(got[k]), np.isinf(want[k])), what


@pytest.mark.parametrize("N", [2048, 4096])
@pytest.mark.parametrize("sig", sorted(signals.ALL))
def test_pair_kernel_matches_oracle_and_single_wave_kernel(gpu_fx, oracle, sig, N):
    C, T = 6, 14
    hops = signals.ALL[sig](C, T, N)
    pair, single = analysers(gpu_fx, C, N)
    got, ref = pair.push_hops(hops), single.push_hops(hops)
    oraw, osm = oracle.push_hops(hops, N)
    close(got[0], oraw, "%s N=%d pair raw" % (sig, N))
    close(got[1], osm, "%s N=%d pair smoothed" % (sig, N))
    same_decisions(got, ref, "%s N=%d" % (sig, N))


@pytest.mark.parametrize("N,C,T", [(2048, 300, 21), (4096, 70, 19), (2048, 5, 3), (4096, 3, 2)])
def test_pair_kernel_many_channels_ragged_calls_and_state(gpu_fx, oracle, N, C, T):
    """More channels than a round of workgroups, calls shorter than the pairs of a workgroup, state carried from call to call
    (flux state, window tail, histories), one-frame calls through the batch path (k = 1 pair)."""
    hops = np.concatenate([signals.bursts(C, T - T // 2, N, seed=N + C), signals.low_tones(C, T // 2, N)], axis=1)
    pair, single = analysers(gpu_fx, C, N)
    pair.set_tuning(one_hop_kernel=0)
    single.set_tuning(one_hop_kernel=0)
    cuts = [0, 1, 2, T // 2, T] if T > 4 else [0, 1, T]
    got = [pair.push_hops(hops[:, a:b]) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
    ref = single.push_hops(hops)
    got = tuple(np.concatenate([g[k] for g in got], axis=1) for k in (0, 1))
    same_decisions(got, ref, "ragged N=%d" % N)
    sel = np.arange(0, C, max(1, C // 10))
    oraw, osm = oracle.push_hops(hops[sel], N)
    close(got[0][sel], oraw, "ragged pair raw")
    close(got[1][sel], osm, "ragged pair smoothed")
    assert np.array_equal(pair.get_features(), got[1][:, -1], equal_nan=True)


@pytest.mark.parametrize("N,C,T", [(2048, 3, 130), (4096, 2, 129)])
def test_pair_kernel_cut_in_time_equals_uncut_bitwise(gpu_fx, oracle, N, C, T):
    """Work units cut in time hand the flux state from workgroup to workgroup (as in fx_frame_kernel): the pair kernel's
    cut launch must equal its own uncut launch bit for bit, and the oracle's values."""
    hops = signals.bursts(C, T, N, seed=12)
    whole = gpu_fx.BatchAnalyser(C, N)
    whole.set_tuning(waves_per_frame=2, frames_per_unit=0)
    want = whole.push_hops(hops)
    oraw, osm = oracle.push_hops(hops, N)
    close(want[0], oraw, "pair uncut raw")
    close(want[1], osm, "pair uncut smoothed")
    for unit in (16, 100):
        an = gpu_fx.BatchAnalyser(C, N)
        an.set_tuning(waves_per_frame=2, frames_per_unit=unit)
        got = an.push_hops(hops)
        assert np.array_equal(got[0], want[0], equal_nan=True) and np.array_equal(got[1], want[1], equal_nan=True), unit


@pytest.mark.parametrize("N", [2048, 4096])
def test_pair_kernel_fp16_frames_orders_and_settings(gpu_fx, oracle, N):
    C, T = 4, 24
    h16 = gpu_fx.synth.hops(C, T, N).astype(np.float16)
    pair, _ = analysers(gpu_fx, C, N, order=1)
    pair.set_gain(0.75)
    pair.set_onset_detection_type(2)
    got = pair.push_hops(h16)
    oraw, osm = oracle.push_hops(h16.astype(np.float32), N, order=1, gain=0.75, onset_type=2)
    close(got[0], oraw, "pair fp16 raw")
    close(got[1], osm, "pair fp16 smoothed")
    frames = gpu_fx.synth.frames(C, T, N, first_channel=5)
    pair2, _ = analysers(gpu_fx, C, N)
    g2 = pair2.process_frames(frames)
    o2 = oracle.process_frames(frames, N)
    close(g2[0], o2[0], "pair frames raw")
    close(g2[1], o2[1], "pair frames smoothed")


@pytest.mark.parametrize("N,C", [(2048, 5), (4096, 1), (4096, 9)])
def test_one_hop_calls_on_pairs_equal_the_pair_kernel_bitwise(gpu_fx, oracle, N, C):
    """One hop per call with every analyser on a pair of wavefronts (fx_hop_pair_kernel: six wavefronts per channel, one launch)
    -- through the pinned ring and through fx_push_hops -- runs the sections fx_pair_kernel runs: bit for bit the batch
    path's values (pair kernel + fused tail), with setter calls in between, and the oracle's within the parity bar."""
    nb = 30
    hops = np.concatenate([signals.bursts(C, nb // 2, N, seed=15), signals.low_tones(C, nb - nb // 2, N)], axis=1)

    def settings(an, b):
        if b == 6:
            an.set_gain(0.5)
            an.set_onset_window_length(4)
        if b == 13:
            an.set_onset_detection_type(2)
            an.set_onset_detection_sensitivity(0.2)
        if b == 21:
            an.sample_rate_changed(44100.0)

    ref = gpu_fx.BatchAnalyser(C, N)
    ref.set_tuning(waves_per_frame=2, one_hop_kernel=0)           # fx_pair_kernel + fx_tail_fused_kernel
    want = []
    for b in range(nb):
        settings(ref, b)
        want.append(ref.push_hops(hops[:, b:b + 1]))
    want = tuple(np.concatenate([x[k] for x in want], 1) for k in (0, 1))

    direct = gpu_fx.BatchAnalyser(C, N)
    direct.set_tuning(waves_per_frame=2)                          # one-frame calls: fx_hop_pair_kernel
    got = []
    for b in range(nb):
        settings(direct, b)
        got.append(direct.push_hops(hops[:, b:b + 1]))
    for k in (0, 1):
        assert np.array_equal(np.concatenate([g[k] for g in got], 1), want[k], equal_nan=True), k

    an = gpu_fx.BatchAnalyser(C, N)
    an.set_tuning(waves_per_frame=2)
    st = gpu_fx.HopStream(an, 1, slots=3)
    got = []
    for b in range(nb):
        settings(an, b)
        if st.in_flight() == 2:
            got.append(st.collect())
        st.push(hops[:, b:b + 1])
    while st.in_flight():
        got.append(st.collect())
    st.close()
    for k in (0, 1):
        assert np.array_equal(np.concatenate([g[k] for g in got], 1), want[k], equal_nan=True), k
    assert np.array_equal(an.get_features(), ref.get_features(), equal_nan=True)

    chans = [oracle.Channel(N) for _ in range(C)]
    oraw, osm = [], []
    for b in range(nb):
        if b == 6:
            [(ch.set_gain(0.5), ch.set_onset_window(4)) for ch in chans]
        if b == 13:
            [(ch.set_onset_type(2), ch.set_onset_sensitivity(0.2)) for ch in chans]
        if b == 21:
            [ch.set_sample_rate(44100.0) for ch in chans]
        r = [ch.push_hops(hops[c, b:b + 1]) for c, ch in enumerate(chans)]
        oraw.append(np.stack([x[0] for x in r])); osm.append(np.stack([x[1] for x in r]))
    close(want[0], np.concatenate(oraw, 1), "hop pairs raw")
    close(want[1], np.concatenate(osm, 1), "hop pairs smoothed")
