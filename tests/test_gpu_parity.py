"""GPU parity tests: the HIP path, called through the C ABI (include/fx.h), against the CPU oracle
on the same inputs.  Bar: onset bit-exact; every other slot within 1e-5 relative (BASELINE.json
north_star); NaN/inf must match exactly.  All tests here need a real MI355X."""
import glob
import os

import numpy as np
import pytest

import signals

pytestmark = pytest.mark.gpu

RTOL = 1e-5
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
SIZES = [256, 512, 1024, 2048, 4096]


def close(got, want, what):
    from oracle import fx_oracle as fo
    return signals.assert_features_close(got, want, RTOL, fo.FEATURE_NAMES, what)


@pytest.mark.parametrize("N", SIZES)
@pytest.mark.parametrize("sig", sorted(signals.ALL))
def test_hops_match_oracle(gpu_fx, oracle, sig, N):
    C, T = 6, 14
    hops = signals.ALL[sig](C, T, N)
    an = gpu_fx.BatchAnalyser(C, N)
    raw, sm = an.push_hops(hops)
    oraw, osm = oracle.push_hops(hops, N)
    close(raw, oraw, "%s N=%d raw" % (sig, N))
    close(sm, osm, "%s N=%d smoothed" % (sig, N))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_committed_fixtures(gpu_fx, path):
    g = np.load(path)
    hops = g["hops"]
    an = gpu_fx.BatchAnalyser(hops.shape[0], int(g["window_size"]), float(g["sample_rate"]), order=int(g["order"]))
    raw, sm = an.push_hops(hops)
    close(raw, g["raw"], "golden raw")
    close(sm, g["smoothed"], "golden smoothed")


def test_random_reference_cases(gpu_fx):
    """The HIP path against outputs of the reference's own headers on forty random cases (tests/golden/random/cases.npz,
    made in the build container by tests/golden/make_random_cases.py; hops regenerated from seeds and CRC-checked): no
    oracle in between.  Some of the cases are long calls, which the frame kernel cuts in time."""
    from test_oracle import _random_reference_cases
    for k, p, hops, raw, sm in _random_reference_cases():
        an = gpu_fx.BatchAnalyser(p["C"], p["N"], p["sample_rate"], order=p["order"])
        an.set_onset_detection_type(p["onset_type"])
        an.set_onset_window_length(p["onset_window"])
        an.set_onset_detection_sensitivity(p["sensitivity"])
        an.set_gain(p["gain"])
        graw, gsm = an.push_hops(hops)
        close(graw, raw, "random case %d %r raw" % (k, p))
        close(gsm, sm, "random case %d %r smoothed" % (k, p))
        an.close()


@pytest.mark.parametrize("N", [1024, 2048])
def test_preassembled_frames_match_oracle(gpu_fx, oracle, N):
    frames = gpu_fx.synth.frames(5, 12, N, first_channel=3)
    raw, sm = gpu_fx.BatchAnalyser(5, N).process_frames(frames)
    oraw, osm = oracle.process_frames(frames, N)
    close(raw, oraw, "frames raw")
    close(sm, osm, "frames smoothed")


@pytest.mark.parametrize("order", [0, 1, 2])
def test_order_modes(gpu_fx, oracle, order):
    hops = signals.bursts(4, 30, 1024, seed=21)
    raw, sm = gpu_fx.BatchAnalyser(4, 1024, order=order).push_hops(hops)
    oraw, osm = oracle.push_hops(hops, 1024, order=order)
    close(raw, oraw, "order %d raw" % order)
    close(sm, osm, "order %d smoothed" % order)


@pytest.mark.parametrize("otype", [0, 1, 2])
@pytest.mark.parametrize("window", [3, 5, 9, 21])
def test_onset_settings(gpu_fx, oracle, otype, window):
    hops = signals.bursts(6, 48, 1024, seed=100 + window)
    an = gpu_fx.BatchAnalyser(6, 1024)
    an.set_onset_detection_type(otype)
    an.set_onset_window_length(window)
    an.set_onset_detection_sensitivity(0.3)
    raw, sm = an.push_hops(hops)
    oraw, osm = oracle.push_hops(hops, 1024, onset_type=otype, onset_window=window, onset_sensitivity=0.3)
    assert oraw[:, :, 0].sum() > 0 or otype == 2          # the case really contains onsets
    close(raw, oraw, "onset raw")
    close(sm, osm, "onset smoothed")


def test_settings_changed_mid_stream(gpu_fx, oracle):
    N, C = 1024, 3
    hops = signals.bursts(C, 40, N, seed=33)
    an = gpu_fx.BatchAnalyser(C, N)
    chans = [oracle.Channel(N) for _ in range(C)]
    got, want = [], []

    def both(lo, hi):
        got.append(an.push_hops(hops[:, lo:hi]))
        want.append([ch.push_hops(hops[c, lo:hi]) for c, ch in enumerate(chans)])

    both(0, 11)
    an.set_onset_window_length(4)
    an.set_gain(0.5)
    [ch.set_onset_window(4) for ch in chans]
    [ch.set_gain(0.5) for ch in chans]
    both(11, 23)
    an.set_onset_detection_type(0)
    an.sample_rate_changed(44100.0)
    [ch.set_onset_type(0) for ch in chans]
    [ch.set_sample_rate(44100.0) for ch in chans]
    both(23, 40)
    for (raw, sm), w in zip(got, want):
        close(raw, np.stack([x[0] for x in w]), "mid-stream raw")
        close(sm, np.stack([x[1] for x in w]), "mid-stream smoothed")


@pytest.mark.parametrize("N", [1024, 4096])
def test_split_calls_equal_one_call_bitwise(gpu_fx, N):
    C, T = 4, 21
    hops = signals.bursts(C, T, N, seed=5)
    one = gpu_fx.BatchAnalyser(C, N).push_hops(hops)
    an = gpu_fx.BatchAnalyser(C, N)
    parts = [an.push_hops(hops[:, a:b]) for a, b in ((0, 1), (1, 2), (2, 9), (9, 10), (10, 21))]
    for k in (0, 1):
        assert np.array_equal(np.concatenate([p[k] for p in parts], axis=1), one[k], equal_nan=True)
    assert np.array_equal(an.get_features(), one[1][:, -1], equal_nan=True)


@pytest.mark.parametrize("N,C,T", [(1024, 3, 300), (1024, 2, 700), (2048, 2, 130), (256, 5, 200), (4096, 2, 129)])
def test_long_calls_cut_in_time_match_oracle_and_the_uncut_launch(gpu_fx, oracle, monkeypatch, N, C, T):
    """Calls of >= 128 frames per channel are cut in time (FrameParams::num_chunks): several workgroups per channel, each
    taking 64 consecutive frames, the flux state handed from one to the next through global memory behind a ticket
    order.  Bursts with digital silence in between put the reference's skip rule (no update of the previous magnitudes,
    SpectralCharacteristics.h:121-123) across chunk boundaries; the last chunk is ragged."""
    hops = signals.bursts(C, T, N, seed=12)
    raw, sm = gpu_fx.BatchAnalyser(C, N).push_hops(hops)
    oraw, osm = oracle.push_hops(hops, N)
    close(raw, oraw, "chunked raw")
    close(sm, osm, "chunked smoothed")
    frames = gpu_fx.synth.frames(C, T, N, first_channel=7)
    fraw, fsm = gpu_fx.BatchAnalyser(C, N).process_frames(frames)
    for per_chunk in ("0", "16", "100"):
        monkeypatch.setenv("FX_FRAMES_PER_CHUNK", per_chunk)
        r0, s0 = gpu_fx.BatchAnalyser(C, N).push_hops(hops)
        assert np.array_equal(r0, raw, equal_nan=True) and np.array_equal(s0, sm, equal_nan=True), per_chunk
        r1, s1 = gpu_fx.BatchAnalyser(C, N).process_frames(frames)
        assert np.array_equal(r1, fraw, equal_nan=True) and np.array_equal(s1, fsm, equal_nan=True), per_chunk


def test_cut_launch_with_many_channels_equals_the_uncut_one_bitwise(gpu_fx, monkeypatch):
    """Enough work units (2400 channels x 3 chunks of 256-pt frames) that chunks of one channel run on different CUs
    while others wait for their predecessor: every value must equal the one-workgroup-per-channel launch."""
    C, T, N = 2400, 192, 256
    hops = np.tile(signals.bursts(48, T, N, seed=3), (C // 48, 1, 1))
    hops = (hops * np.linspace(0.2, 1.0, C, dtype=np.float32)[:, None, None]).astype(np.float32)
    an = gpu_fx.BatchAnalyser(C, N)
    raw, sm = an.push_hops(hops)
    raw2, sm2 = an.push_hops(hops)                       # a second call continues from the first's state
    monkeypatch.setenv("FX_FRAMES_PER_CHUNK", "0")
    an0 = gpu_fx.BatchAnalyser(C, N)
    want = an0.push_hops(hops), an0.push_hops(hops)
    assert np.array_equal(raw, want[0][0], equal_nan=True) and np.array_equal(sm, want[0][1], equal_nan=True)
    assert np.array_equal(raw2, want[1][0], equal_nan=True) and np.array_equal(sm2, want[1][1], equal_nan=True)


@pytest.mark.parametrize("N,shape", [(1024, (3, 2)), (1024, (2, 8)), (2048, (3, 4)), (2048, (2, 6)), (512, (4, 2)),
                                     (4096, (2, 4)), (4096, (8, 1)), (4096, (3, 2))])      # 4096: the hand-over counters of up to 8 channels share the twiddle image's gap
def test_workgroup_shapes_give_the_same_bits(gpu_fx, monkeypatch, N, shape):
    """Channels per workgroup x waves per channel is a scheduling choice (several channels can share one workgroup's
    twiddle table in LDS): every shape must give the results of the default one bit for bit, including a last
    workgroup that is only partly filled (7 channels in groups of 2, 3 or 4) and calls shorter than the wave count."""
    C, T = 7, 13
    hops = signals.bursts(C, T, N, seed=N + shape[0])
    want = gpu_fx.BatchAnalyser(C, N).push_hops(hops)
    monkeypatch.setenv("FX_CHANNELS_PER_WG", str(shape[0]))
    monkeypatch.setenv("FX_WAVES", str(shape[1]))
    an = gpu_fx.BatchAnalyser(C, N)
    got = [an.push_hops(hops[:, a:b]) for a, b in ((0, 1), (1, 4), (4, 13))]
    for k in (0, 1):
        assert np.array_equal(np.concatenate([g[k] for g in got], 1), want[k], equal_nan=True)


def test_streaming_one_hop_per_call_matches_oracle(gpu_fx, oracle):
    N, C, T = 2048, 3, 26
    hops = signals.tone_vibrato_noise(C, T, N, seed=8)
    an = gpu_fx.BatchAnalyser(C, N)
    outs = [an.push_hops(hops[:, t:t + 1]) for t in range(T)]
    oraw, osm = oracle.push_hops(hops, N)
    close(np.concatenate([o[0] for o in outs], axis=1), oraw, "streaming raw")
    close(np.concatenate([o[1] for o in outs], axis=1), osm, "streaming smoothed")


def test_reset_state_restores_a_fresh_analyser(gpu_fx):
    hops = signals.tone_vibrato_noise(2, 12, 1024)
    an = gpu_fx.BatchAnalyser(2, 1024)
    a = an.push_hops(hops)
    an.push_hops(signals.loud_noise(2, 5, 1024))
    an.reset_state()
    b = an.push_hops(hops)
    assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1], equal_nan=True)


def test_fp16_samples(gpu_fx, oracle):
    N, C, T = 4096, 2, 10                    # BASELINE config 5 shape: 4096-pt, fp16 samples
    h16 = gpu_fx.synth.hops(C, T, N).astype(np.float16)
    raw16, sm16 = gpu_fx.BatchAnalyser(C, N).push_hops(h16)
    raw32, sm32 = gpu_fx.BatchAnalyser(C, N).push_hops(h16.astype(np.float32))
    assert np.array_equal(raw16, raw32, equal_nan=True) and np.array_equal(sm16, sm32, equal_nan=True)
    oraw, osm = oracle.push_hops(h16.astype(np.float32), N)
    close(raw16, oraw, "fp16 raw")
    close(sm16, osm, "fp16 smoothed")


def test_device_buffers_equal_host_buffers(gpu_fx):
    import torch
    N, C, T = 1024, 8, 9
    hops = gpu_fx.synth.hops(C, T, N)
    host = gpu_fx.BatchAnalyser(C, N).push_hops(hops)
    an = gpu_fx.BatchAnalyser(C, N)
    d = torch.from_numpy(hops).cuda()
    raw, sm = an.push_hops(d)
    an.sync()
    assert raw.is_cuda and np.array_equal(raw.cpu().numpy(), host[0], equal_nan=True)
    assert np.array_equal(sm.cpu().numpy(), host[1], equal_nan=True)
    fr = torch.from_numpy(gpu_fx.synth.frames(C, T, N)).cuda()
    an2 = gpu_fx.BatchAnalyser(C, N)
    raw2, sm2 = an2.process_frames(fr)
    an2.sync()
    assert np.array_equal(raw2.cpu().numpy(), host[0], equal_nan=True)      # frames of the same stream


def test_error_paths(gpu_fx):
    an = gpu_fx.BatchAnalyser(2, 1024)
    with pytest.raises(gpu_fx.FxError):
        an.set_onset_window_length(0)
    with pytest.raises(gpu_fx.FxError):
        an.set_onset_detection_type(7)
    with pytest.raises(gpu_fx.FxError):
        gpu_fx.BatchAnalyser(2, 1024, device=99)
    raw, sm = an.push_hops(np.zeros((2, 0, 512), np.float32))       # empty input is a no-op
    assert raw.shape == (2, 0, 12)


def test_full_size_config_properties(gpu_fx, oracle):
    """BASELINE configs[1] shape (1024 channels x 1024-pt frames): properties that do not need the
    oracle at full size, plus an oracle spot check on a random subset of channels."""
    N, C, T = 1024, 1024, 32
    frames = gpu_fx.synth.frames(C, T, N)
    an = gpu_fx.BatchAnalyser(C, N)
    raw, sm = an.process_frames(frames)
    # repeatable bit for bit
    an.reset_state()
    raw_b, sm_b = an.process_frames(frames)
    assert np.array_equal(raw, raw_b, equal_nan=True) and np.array_equal(sm, sm_b, equal_nan=True)
    # a channel's result does not depend on which shard / context it was analysed in
    lo = gpu_fx.BatchAnalyser(C // 2, N).process_frames(frames[: C // 2])
    hi = gpu_fx.BatchAnalyser(C // 2, N).process_frames(frames[C // 2:])
    assert np.array_equal(np.concatenate([lo[0], hi[0]]), raw, equal_nan=True)
    assert np.array_equal(np.concatenate([lo[1], hi[1]]), sm, equal_nan=True)
    # channels 72 apart carry the same tone + different noise: f0 of the noiseless part agrees often,
    # and OER is always a copy of HER
    assert np.array_equal(raw[:, :, 10], raw[:, :, 9])
    assert np.all(np.isfinite(raw[:, :, [1, 2, 3, 6, 7, 9, 11]]))
    pick = np.random.default_rng(0).choice(C, 24, replace=False)
    oraw, osm = oracle.process_frames(frames[pick], N)
    close(raw[pick], oraw, "full-size spot check raw")
    close(sm[pick], osm, "full-size spot check smoothed")


def test_config3_shape_spot_check(gpu_fx, oracle):
    """BASELINE configs[2]: 4096 channels x 2048-pt frames, flux state resident in HBM across calls."""
    N, C, T = 2048, 4096, 6
    an = gpu_fx.BatchAnalyser(C, N)
    h1 = gpu_fx.synth.hops(C, T, N)
    h2 = gpu_fx.synth.hops(C, T, N, first_hop=T)
    a = an.push_hops(h1)
    b = an.push_hops(h2)
    pick = np.random.default_rng(1).choice(C, 12, replace=False)
    oraw, osm = oracle.push_hops(np.concatenate([h1[pick], h2[pick]], axis=1), N)
    close(np.concatenate([a[0][pick], b[0][pick]], axis=1), oraw, "config 3 raw")
    close(np.concatenate([a[1][pick], b[1][pick]], axis=1), osm, "config 3 smoothed")


def test_hop_stream_equals_push_hops_bitwise(gpu_fx):
    """Streaming ingest (pinned ring, H2D on a side stream) must give exactly what fx_push_hops gives,
    in order, for ragged use of the ring (collect lagging behind submit)."""
    N, C, B, nb = 2048, 5, 3, 7
    hops = signals.bursts(C, B * nb, N, seed=77)
    want = gpu_fx.BatchAnalyser(C, N).push_hops(hops)
    an = gpu_fx.BatchAnalyser(C, N)
    st = gpu_fx.HopStream(an, B, slots=3)
    got_raw, got_sm = [], []
    for b in range(nb):
        if st.in_flight() == 3:
            r, s = st.collect()
            got_raw.append(r); got_sm.append(s)
        st.push(hops[:, b * B:(b + 1) * B])
    while st.in_flight():
        r, s = st.collect()
        got_raw.append(r); got_sm.append(s)
    assert np.array_equal(np.concatenate(got_raw, 1), want[0], equal_nan=True)
    assert np.array_equal(np.concatenate(got_sm, 1), want[1], equal_nan=True)
    with pytest.raises(gpu_fx.FxError):
        st.collect()                                   # nothing in flight
    st.close()


@pytest.mark.parametrize("graph", ["0", "1"])
def test_hop_stream_settings_change_mid_stream(gpu_fx, monkeypatch, graph):
    """The ring in both of its modes (FX_STREAM_GRAPH=1: every step replayed from a captured hipGraph, whose per-call
    scalars -- frame counters, gain, onset settings, sample rate -- travel through device memory; =0: plain launches
    with the copy on a side stream) must equal fx_push_hops with the same setter calls in between, bit for bit."""
    monkeypatch.setenv("FX_STREAM_GRAPH", graph)
    N, C, B, nb = 1024, 4, 2, 12
    hops = signals.bursts(C, B * nb, N, seed=91)

    def settings(an, b):
        if b == 3:
            an.set_gain(0.5)
            an.set_onset_window_length(4)
        if b == 6:
            an.set_onset_detection_type(2)
            an.set_onset_detection_sensitivity(0.2)
        if b == 8:
            an.sample_rate_changed(44100.0)
        if b == 10:
            an.reset_state()

    ref = gpu_fx.BatchAnalyser(C, N)
    want = []
    for b in range(nb):
        settings(ref, b)
        want.append(ref.push_hops(hops[:, b * B:(b + 1) * B]))
    an = gpu_fx.BatchAnalyser(C, N)
    st = gpu_fx.HopStream(an, B, slots=3)
    got = []
    for b in range(nb):
        # a setter applies to the batches submitted after it, as with fx_push_hops
        settings(an, b)
        if st.in_flight() == 3:
            got.append(st.collect())
        st.push(hops[:, b * B:(b + 1) * B])
    while st.in_flight():
        got.append(st.collect())
    for k in (0, 1):
        assert np.array_equal(np.concatenate([g[k] for g in got], 1), np.concatenate([w[k] for w in want], 1), equal_nan=True)
    assert np.array_equal(an.get_features(), ref.get_features(), equal_nan=True)
    st.close()


@pytest.mark.parametrize("N,C,order", [(1024, 5, 0), (2048, 3, 0), (4096, 1, 0), (4096, 4, 0), (1024, 70, 1), (2048, 2, 2)])
def test_one_hop_per_call_kernel_equals_push_hops_bitwise(gpu_fx, monkeypatch, N, C, order):
    """One hop per call through the ring runs as ONE launch of fx_hop_kernel (three wavefronts per channel, results and
    completion flag written by the kernel, the host polling the flag).  It must equal fx_push_hops bit for bit -- the
    same sections of the frame code, the same tail -- with setter calls in between, with up to three calls in flight,
    and it must equal the captured two-kernel step it replaces (FX_STREAM_HOP_KERNEL=0)."""
    nb = 40
    hops = np.concatenate([signals.bursts(C, nb // 2, N, seed=5), signals.low_tones(C, nb - nb // 2, N)], axis=1)

    def settings(an, b):
        if b == 7:
            an.set_gain(0.5)
            an.set_onset_window_length(4)
        if b == 15:
            an.set_onset_detection_type(2)
            an.set_onset_detection_sensitivity(0.2)
        if b == 22:
            an.sample_rate_changed(44100.0)
        if b == 31:
            an.reset_state()

    ref = gpu_fx.BatchAnalyser(C, N, order=order)
    ref.set_tuning(one_hop_kernel=0)                    # the batch kernels (frame kernel + fused tail), not the hop kernel
    want = []
    for b in range(nb):
        settings(ref, b)
        want.append(ref.push_hops(hops[:, b:b + 1]))

    def through_ring(depth):
        an = gpu_fx.BatchAnalyser(C, N, order=order)
        st = gpu_fx.HopStream(an, 1, slots=3)
        got = []
        for b in range(nb):
            settings(an, b)
            if st.in_flight() == depth:
                got.append(st.collect())
            st.push(hops[:, b:b + 1])
        while st.in_flight():
            got.append(st.collect())
        feats = an.get_features()
        st.close()
        return got, feats

    for depth in (1, 3):
        got, feats = through_ring(depth)
        for k in (0, 1):
            assert np.array_equal(np.concatenate([g[k] for g in got], 1), np.concatenate([w[k] for w in want], 1), equal_nan=True), (depth, k)
        assert np.array_equal(feats, ref.get_features(), equal_nan=True)
    monkeypatch.setenv("FX_STREAM_HOP_KERNEL", "0")
    got, feats = through_ring(2)
    for k in (0, 1):
        assert np.array_equal(np.concatenate([g[k] for g in got], 1), np.concatenate([w[k] for w in want], 1), equal_nan=True)


def test_hop_stream_fp16_4096(gpu_fx, oracle):
    """BASELINE configs[4] shape: 4096-pt windows, fp16 samples streamed through the pinned ring."""
    N, C, B, nb = 4096, 1, 1, 12
    h16 = gpu_fx.synth.hops(C, B * nb, N).astype(np.float16)
    an = gpu_fx.BatchAnalyser(C, N)
    st = gpu_fx.HopStream(an, B, slots=2, dtype=np.float16)
    outs = []
    for b in range(nb):
        if st.in_flight() == 2:
            outs.append(st.collect())
        slot = st.slot()
        slot[...] = h16[:, b * B:(b + 1) * B]
        st.submit()
    while st.in_flight():
        outs.append(st.collect())
    oraw, osm = oracle.push_hops(h16.astype(np.float32), N)
    close(np.concatenate([o[0] for o in outs], 1), oraw, "stream fp16 raw")
    close(np.concatenate([o[1] for o in outs], 1), osm, "stream fp16 smoothed")


def test_non_finite_and_extreme_inputs_do_not_disturb_other_channels(gpu_fx, oracle):
    """NaN / inf / 1e6 / denormal samples: the call must return, and channels without such samples
    must be unaffected bit for bit.  (Parity is not claimed for the poisoned channels' harmonic slots:
    the reference indexes out of bounds there, HarmonicCharacteristics.h:205.)"""
    N, C, T = 1024, 6, 10
    hops = signals.tone_vibrato_noise(C, T, N, seed=5)
    clean = gpu_fx.BatchAnalyser(C, N).push_hops(hops)
    bad = hops.copy()
    bad[1, 3, 17] = np.nan
    bad[2, 4, 100] = np.inf
    bad[3, 5, :] = 1e6                         # as loud as the path's own fp32 arithmetic allows: beyond ~1e8 the
                                               # autocorrelation squares overflow, every cnd is NaN, the lag is -1
                                               # and the reference indexes bins out of bounds (no defined answer)
    bad[4, 2, :] = 1e-42                       # denormals
    got = gpu_fx.BatchAnalyser(C, N).push_hops(bad)
    for c in (0, 5):
        assert np.array_equal(got[0][c], clean[0][c], equal_nan=True)
        assert np.array_equal(got[1][c], clean[1][c], equal_nan=True)
    assert np.array_equal(got[0][1, :3], clean[0][1, :3])          # frames before the NaN arrive are unaffected
    assert np.isnan(got[0][1, 3, 1])                                  # RMS of the poisoned frame is NaN, as in the reference
    # finite-but-extreme channels still match the oracle
    oraw, osm = oracle.push_hops(bad[3:5], N)
    close(got[0][3:5], oraw, "extreme raw")
    close(got[1][3:5], osm, "extreme smoothed")


@pytest.mark.parametrize("which,mask", [("spectral", 1), ("harmonic", 2)])
@pytest.mark.parametrize("N", [512, 1024, 2048])
def test_single_analyser_modes(gpu_fx, oracle, which, mask, N):
    """Only one of the reference's two analyser threads constructed (FX_SPECTRAL_ONLY / FX_HARMONIC_ONLY):
    the other analyser's slots keep raw 0 and getValue NaN."""
    C, T = 5, 26
    hops = signals.bursts(C, T, N, seed=N + mask)
    an = gpu_fx.BatchAnalyser(C, N, analysers=which)
    an.set_onset_detection_type(2)
    a = an.push_hops(hops[:, :9])
    b = an.push_hops(hops[:, 9:])
    oraw, osm = oracle.push_hops(hops, N, analysers=mask, onset_type=2)
    close(np.concatenate([a[0], b[0]], 1), oraw, which + " raw")
    close(np.concatenate([a[1], b[1]], 1), osm, which + " smoothed")
    absent = [2, 9, 10, 11] if which == "spectral" else [0, 3, 4, 5, 6, 7, 8]
    assert np.isnan(b[1][:, :, absent]).all() and not b[0][:, :, absent].any()
