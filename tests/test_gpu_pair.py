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
