"""fx_push_samples / fx_stream_submit_samples: the collector's real interface (ref AudioDataCollector.h:36-94,
RealTimeAudioAnalysis.h:205-219) -- device blocks of ANY length per call.  Bar: bit for bit what fx_push_hops makes of the same
stream cut into hops, whatever the block length, sample format or memory kind; against the oracle and the committed fixtures."""
import os
import subprocess

import numpy as np
import pytest

import signals

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLOCKS = [1, 63, 441, 480, 512, 1000, 4097]


def same(a, b):
    return np.array_equal(a, b, equal_nan=True)


def feed_blocks(an, stream, block, device=False, sample_format=None):
    """stream [C][total] through push_samples in blocks of `block` samples (the last one shorter): every vector it returned, in order"""
    import torch
    C, total = stream.shape[0], stream.shape[1] // (3 if sample_format == "s24" else 1)
    per = 3 if sample_format == "s24" else 1
    raws, sms = [], []
    for at in range(0, total, block):
        piece = np.ascontiguousarray(stream[:, at * per:(at + block) * per])
        if device:
            r, s = an.push_samples(torch.from_numpy(piece).cuda(), sample_format=sample_format)
            r, s = r.cpu().numpy(), s.cpu().numpy()
        else:
            r, s = an.push_samples(piece, sample_format=sample_format)
        assert r.shape == s.shape and r.shape[0] == C and r.shape[2] == 12
        raws.append(r); sms.append(s)
    return np.concatenate(raws, axis=1), np.concatenate(sms, axis=1)


@pytest.mark.parametrize("N", [1024, 2048, 4096])
@pytest.mark.parametrize("block", BLOCKS)
@pytest.mark.parametrize("fmt", ["f32", "s16"])
def test_blocks_of_any_length_equal_push_hops_bitwise(gpu_fx, oracle, N, block, fmt):
    C, H = 3, N // 2
    T = 3 if block == 1 else 9                                 # (a sample per call: thousands of calls per hop)
    tail = 0 if block == 512 else 77                           # samples past the last whole hop stay pending
    hops = np.concatenate([signals.bursts(C, T // 3, N, seed=N + block), signals.tone_vibrato_noise(C, T - T // 3, N, seed=block)], axis=1)
    extra = signals.tone_vibrato_noise(C, 1, N, seed=5)[:, 0, :tail]
    stream = np.concatenate([hops.reshape(C, -1), extra], axis=1)
    if fmt == "s16":
        stream = np.clip(np.round(stream * 32768.0), -32768, 32767).astype(np.int16)
        hops = stream[:, :T * H].reshape(C, T, H)
    ref = gpu_fx.BatchAnalyser(C, N)
    ref.set_gain(0.75)
    want = ref.push_hops(hops)
    for device in (False, True):
        an = gpu_fx.BatchAnalyser(C, N)
        an.set_gain(0.75)
        got = feed_blocks(an, stream, block, device=device)
        assert got[0].shape == (C, T, 12), (got[0].shape, T)
        assert same(got[0], want[0]) and same(got[1], want[1]), (N, block, fmt, device)
        assert an.pending_samples() == tail
        assert same(an.get_features(), ref.get_features())
        an.close()
    # and the oracle, on the floats the kernels widen the samples to
    floats = hops.astype(np.float32) / np.float32(32768.0) if fmt == "s16" else hops
    oraw, osm = oracle.push_hops(floats, N, gain=0.75)
    signals.assert_features_close(want[0], oraw, 1e-5, oracle.FEATURE_NAMES, "raw")
    signals.assert_features_close(want[1], osm, 1e-5, oracle.FEATURE_NAMES, "smoothed")


@pytest.mark.parametrize("name,block", [("tone_2048", 480), ("bursts_1024_harmfirst", 441)])
def test_committed_fixtures_through_device_blocks(gpu_fx, name, block):
    """the reference's own outputs (tests/golden, made by the unmodified headers) met from blocks of an audio device's length"""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    hops = g["hops"]
    C, T, H = hops.shape
    an = gpu_fx.BatchAnalyser(C, int(g["window_size"]), float(g["sample_rate"]), order=int(g["order"]))
    raw, sm = feed_blocks(an, hops.reshape(C, -1), block)
    assert raw.shape == (C, T, 12)
    from oracle import fx_oracle as fo
    signals.assert_features_close(raw, g["raw"], 1e-5, fo.FEATURE_NAMES, "golden raw")
    signals.assert_features_close(sm, g["smoothed"], 1e-5, fo.FEATURE_NAMES, "golden smoothed")


def test_the_reference_collectors_own_vectors(gpu_fx):
    """tests/golden/blocks/cases.npz: what the reference's AudioDataCollector + overlapper + analysers produce when fed device blocks with setGain /
    clearBuffer in between (made by the unmodified headers, tools/refdiff/refdiff_blocks.cpp) -- replayed through fx_push_samples / fx_set_gain /
    fx_clear_pending, host blocks and device blocks"""
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from block_cases import CASES, replay, stream_of
    from oracle import fx_oracle as fo
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "blocks", "cases.npz"))
    for k, (name, N, C, hops, extra, block, order, events) in enumerate(CASES):
        stream = stream_of(name, N, C, hops, extra, seed=100 + k)
        for device in (False, True):
            an = gpu_fx.BatchAnalyser(C, N, order=order)

            def push_block(piece):
                r, s = an.push_samples(torch.from_numpy(piece).cuda() if device else piece)
                return (r.cpu().numpy(), s.cpu().numpy()) if device else (r, s)

            def control(name, value):
                {"sensitivity": an.set_onset_detection_sensitivity, "onset_window": an.set_onset_window_length, "onset_type": an.set_onset_detection_type,
                 "sample_rate": an.sample_rate_changed}[name](value)

            got = replay(stream, N, block, events, push_block, an.set_gain, an.clear_buffer, control)
            raw, sm = np.concatenate([x[0] for x in got], axis=1), np.concatenate([x[1] for x in got], axis=1)
            signals.assert_features_close(raw, g[name + "_raw"], 1e-5, fo.FEATURE_NAMES, name + " raw")
            signals.assert_features_close(sm, g[name + "_smoothed"], 1e-5, fo.FEATURE_NAMES, name + " smoothed")
            assert an.pending_samples() == stream.shape[1] % (N // 2)
            an.close()


def test_packed_24_bit_blocks_and_mixed_block_lengths(gpu_fx):
    """three-byte samples at every byte alignment a block length can produce, block lengths changing from call to call, empty blocks"""
    C, N, T = 5, 1024, 11
    H = N // 2
    rng = np.random.default_rng(3)
    v = rng.integers(-2 ** 23, 2 ** 23, (C, T * H + 100), dtype=np.int64).astype(np.int32)
    v[:, :4 * H] //= 64
    packed = np.asarray(gpu_fx.pack_s24(v))                       # [C][3 * samples]
    want = gpu_fx.BatchAnalyser(C, N).push_hops(gpu_fx.pack_s24(v[:, :T * H].reshape(C, T, H)))
    an = gpu_fx.BatchAnalyser(C, N)
    raws, sms, at = [], [], 0
    while at < v.shape[1]:
        n = int(rng.choice([0, 1, 2, 3, 5, 127, 333, 512, 513, 1500]))
        n = min(n, v.shape[1] - at)
        r, s = an.push_samples(np.ascontiguousarray(packed[:, 3 * at:3 * (at + n)]), sample_format="s24")
        raws.append(r); sms.append(s)
        at += n
    assert same(np.concatenate(raws, 1), want[0]) and same(np.concatenate(sms, 1), want[1])
    assert an.pending_samples() == 100
    with pytest.raises(ValueError):
        an.push_samples(packed[:, :30])                          # bytes are not 24-bit PCM on their dtype alone


@pytest.mark.parametrize("N", [256, 512])
@pytest.mark.parametrize("fmt", ["f16", "s24", "f32"])
@pytest.mark.parametrize("C", [1, 7])
def test_small_windows_single_channels_and_every_format(gpu_fx, N, fmt, C):
    """the smallest windows (carry rows of 512 / 1024 bytes), one channel, half floats and three-byte samples, random block lengths incl. empty ones"""
    import torch
    H, T = N // 2, 23
    rng = np.random.default_rng(N + C)
    x = signals.tone_vibrato_noise(C, T, N, seed=N).reshape(C, -1)
    if fmt == "f16":
        stream = x.astype(np.float16)
        want = gpu_fx.BatchAnalyser(C, N).push_hops(stream.reshape(C, T, H))
        per, sf = 1, None
    elif fmt == "s24":
        v = np.clip(np.round(x.astype(np.float64) * 8388608.0), -8388608, 8388607).astype(np.int32)
        stream = np.asarray(gpu_fx.pack_s24(v))
        want = gpu_fx.BatchAnalyser(C, N).push_hops(gpu_fx.pack_s24(v.reshape(C, T, H)))
        per, sf = 3, "s24"
    else:
        stream = x
        want = gpu_fx.BatchAnalyser(C, N).push_hops(stream.reshape(C, T, H))
        per, sf = 1, None
    for device in (False, True):
        an = gpu_fx.BatchAnalyser(C, N)
        raws, sms, at = [], [], 0
        while at < T * H:
            n = min(int(rng.choice([0, 1, 2, 3, 17, 100, H - 1, H, H + 1, 3 * H + 5])), T * H - at)
            piece = np.ascontiguousarray(stream[:, at * per:(at + n) * per])
            if device and n > 0:
                # a device buffer that starts 4 bytes into an allocation: dword-aligned, not 16-byte aligned
                pad = torch.zeros(piece.nbytes + 64, dtype=torch.uint8, device="cuda")
                view = pad[4:4 + piece.nbytes]
                view.copy_(torch.from_numpy(piece.view(np.uint8).reshape(-1)))
                tv = view.view({"f16": torch.float16, "s24": torch.uint8, "f32": torch.float32}[fmt]).reshape(C, -1)
                r, s_ = an.push_samples(tv, sample_format=sf)
                r, s_ = r.cpu().numpy(), s_.cpu().numpy()
            else:
                r, s_ = an.push_samples(piece, sample_format=sf)
            raws.append(r); sms.append(s_)
            at += n
        assert same(np.concatenate(raws, 1), want[0]) and same(np.concatenate(sms, 1), want[1]), (N, fmt, C, device)
        assert an.pending_samples() == 0


def test_gain_reaches_pending_samples_and_clear_buffer_zeroes_them(gpu_fx):
    """getAnalysisBuffer multiplies by the gain at READ time (AudioDataCollector.h:88): a change applies to what is still pending;
    clearBuffer (:122) turns the pending samples into zeros and keeps the indices."""
    C, N = 2, 1024
    H = N // 2
    x = signals.tone_vibrato_noise(C, 6, N, seed=9).reshape(C, -1)
    an = gpu_fx.BatchAnalyser(C, N)
    an.set_gain(0.5)
    a = an.push_samples(x[:, :2 * H + 200])                     # two hops at 0.5, 200 pending
    an.set_gain(2.0)
    b = an.push_samples(x[:, 2 * H + 200:4 * H + 50])           # hops 2, 3 wholly at 2.0 (their first 200 samples were pending)
    an.clear_buffer()                                           # the 50 pending samples become zeros
    c = an.push_samples(x[:, 4 * H + 50:6 * H])
    assert an.pending_samples() == 0
    ref = gpu_fx.BatchAnalyser(C, N)
    ref.set_gain(0.5)
    wa = ref.push_hops(x[:, :2 * H].reshape(C, 2, H))
    ref.set_gain(2.0)
    wb = ref.push_hops(x[:, 2 * H:4 * H].reshape(C, 2, H))
    y = x.copy()
    y[:, 4 * H:4 * H + 50] = 0.0
    wc = ref.push_hops(y[:, 4 * H:].reshape(C, 2, H))
    for got, want in ((a, wa), (b, wb), (c, wc)):
        assert same(got[0], want[0]) and same(got[1], want[1])


def test_pending_samples_guard_the_hop_interfaces(gpu_fx):
    C, N = 2, 1024
    an = gpu_fx.BatchAnalyser(C, N)
    x = signals.tone_vibrato_noise(C, 2, N).reshape(C, -1)
    r, s = an.push_samples(x[:, :100])
    assert r.shape == (C, 0, 12) and an.pending_samples() == 100
    with pytest.raises(gpu_fx.FxError):
        an.push_hops(x[:, :512].reshape(C, 1, 512))             # whole hops would overtake the pending samples
    with pytest.raises(gpu_fx.FxError):
        an.push_samples(x[:, :10].astype(np.float16))           # one format between hop boundaries
    an.reset_state()
    assert an.pending_samples() == 0
    an.push_hops(x[:, :512].reshape(C, 1, 512))


@pytest.mark.parametrize("N,block", [(1024, 480), (2048, 441), (4096, 1000)])
def test_ring_of_device_blocks_equals_push_hops_bitwise(gpu_fx, N, block):
    """fx_stream_push_samples / fx_stream_collect_samples: the pinned ring fed with device blocks; several blocks in flight"""
    C, T, H = 4, 10, N // 2
    hops = np.concatenate([signals.bursts(C, 4, N, seed=1), signals.tone_vibrato_noise(C, T - 4, N, seed=2)], axis=1)
    stream = np.clip(np.round(hops.reshape(C, -1) * 32768.0), -32768, 32767).astype(np.int16)
    want = gpu_fx.BatchAnalyser(C, N).push_hops(stream.reshape(C, T, H))
    an = gpu_fx.BatchAnalyser(C, N)
    st = gpu_fx.HopStream(an, 2, slots=3, dtype=np.int16)     # a slot holds up to two hops' worth of samples
    raws, sms = [], []
    for at in range(0, stream.shape[1], block):
        if st.in_flight() == 3:
            r, s = st.collect_samples(); raws.append(r); sms.append(s)
        st.push_samples(np.ascontiguousarray(stream[:, at:at + block]))
    while st.in_flight():
        r, s = st.collect_samples(); raws.append(r); sms.append(s)
    assert same(np.concatenate(raws, 1), want[0]) and same(np.concatenate(sms, 1), want[1])
    assert an.pending_samples() == 0
    st.close()


def test_wav_to_osc_device_blocks_give_identical_datagrams(gpu_fx, tmp_path):
    """examples/wav_to_osc --device-block 480: the file played to fx::AudioDataCollector::audioDeviceIOCallback in blocks of 480 samples
    (a 48 kHz device at 10 ms) -> the same datagrams, byte for byte, as whole hops; 16-bit samples direct as well"""
    from test_cpp_host import _build_example, _records
    exe = _build_example(gpu_fx, tmp_path)
    wav = str(tmp_path / "in.wav")
    x = gpu_fx.synth.samples(2, 48000 + 333, first_channel=5).T
    gpu_fx.wav.write_wav(wav, 48000, x, "pcm16")
    outs = {}
    for tag, extra in (("hops", []), ("b480", ["--device-block", "480"]), ("b441", ["--device-block", "441", "--pcm16-direct"]),
                       ("b4097", ["--device-block", "4097"])):
        dump = str(tmp_path / (tag + ".bin"))
        out = subprocess.run([exe, wav, "--window", "2048", "--channel", "1", "--gain", "1.5", "--address", "/Audio/A1", "--dump", dump] + extra,
                             capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        outs[tag] = _records(dump)
    assert len(outs["hops"]) == (48000 + 333) // 1024
    assert outs["b480"] == outs["hops"] and outs["b441"] == outs["hops"] and outs["b4097"] == outs["hops"]


@pytest.mark.parametrize("N", [1024, 2048, 4096])
@pytest.mark.parametrize("fmt", ["f32", "s16", "s24", "f16"])
def test_block_feed_of_every_one_frame_kernel_equals_the_reblocked_path_bitwise(gpu_fx, N, fmt):
    """Round 6: a call that completes exactly one hop no longer re-blocks -- fx_hop_kernel / fx_frame_tail_kernel / fx_frame_kernel<direct>
    read their window from [pending | block] and write what is left over (FrameParams::block_mode, csrc/fx_blocks.hip.h).  Each of the
    three kernels (steered by the knobs the one-frame tests use), every sample format, block lengths that leave the block row on every
    byte alignment and the hop boundary anywhere in a lane's piece: the vectors, the pending samples and the state after the stream are
    those of the same context with the feed switched off (test hook 16: every call through fx_reblock_kernel), bit for bit, host blocks and
    device blocks."""
    import torch
    C, H = 21, N // 2
    rng = np.random.default_rng(N + len(fmt))
    total = 7 * H + 3
    x = signals.tone_vibrato_noise(C, 8, N, seed=N)[:, :, :].reshape(C, -1)[:, :total]
    if fmt == "s16":
        stream, per = np.clip(np.round(x * 32768.0), -32768, 32767).astype(np.int16), 1
    elif fmt == "f16":
        stream, per = x.astype(np.float16), 1
    elif fmt == "s24":
        stream, per = np.asarray(gpu_fx.pack_s24(np.clip(np.round(x * 8388608.0), -2 ** 23, 2 ** 23 - 1).astype(np.int32))), 3
    else:
        stream, per = x, 1
    sf = "s24" if fmt == "s24" else None
    lengths = []
    at = 0
    while at < total:                          # block lengths between H/2 + 1 and H + H/2 - 1: (almost) every call completes exactly one hop
        n = int(rng.integers(H // 2 + 1, H + H // 2))
        n = min(n, total - at)
        lengths.append(n)
        at += n

    def feed(an, device):
        raws, sms, at = [], [], 0
        for n in lengths:
            piece = np.ascontiguousarray(stream[:, per * at:per * (at + n)])
            r, s = an.push_samples(torch.from_numpy(piece).cuda() if device else piece, sample_format=sf)
            if hasattr(r, "cpu"):
                r, s = r.cpu().numpy(), s.cpu().numpy()
            raws.append(r); sms.append(s)
            at += n
        return np.concatenate(raws, 1), np.concatenate(sms, 1), an.pending_samples(), an.get_features()

    ref = gpu_fx.BatchAnalyser(C, N)
    ref.set_gain(0.5)
    ref.set_test_hooks(16)                      # never feed blocks to the kernels
    want = feed(ref, False)
    assert want[0].shape[1] == total // H
    more = np.ascontiguousarray(stream[:, :per * (H - total % H)])       # the pending samples themselves: this block completes a hop out of them
    want_more = ref.push_samples(more, sample_format=sf)
    assert want_more[0].shape == (C, 1, 12)
    ref.close()
    shapes = {"hop kernel": dict(one_hop_kernel=1), "frames + tails in one launch": dict(one_hop_kernel=0, hooks=8), "frame kernel, then the tail kernel": dict(one_hop_kernel=0, hooks=4)}
    for name, knobs in shapes.items():
        for device in ((False, True) if fmt != "s24" else (False,)):
            an = gpu_fx.BatchAnalyser(C, N)
            an.set_gain(0.5)
            an.set_tuning(one_hop_kernel=knobs["one_hop_kernel"])
            an.set_test_hooks(knobs.get("hooks", 0))
            got = feed(an, device)
            assert got[2] == want[2] == total % H
            assert same(got[0], want[0]) and same(got[1], want[1]) and same(got[3], want[3]), (N, fmt, name, device)
            got_more = an.push_samples(more, sample_format=sf)
            assert same(got_more[0], want_more[0]) and same(got_more[1], want_more[1]), (N, fmt, name, device, "the block after")
            an.close()


@pytest.mark.parametrize("fmt", ["f32", "s16", "s24", "f16"])
def test_long_blocks_through_the_batch_kernels_block_fed_form_equal_the_reblocked_path_bitwise(gpu_fx, fmt):
    """Round 6, 1024-point windows: a call that completes MORE than two hops is one launch of fx_frame_kernel<1024, ..., blocks> -- every frame's
    new half from hop t of [pending | block], its old half from hop t - 1, the last frame's wave writing the left-over.  Block lengths from three
    hops to 273 per call, so that the launch is one unit, several units of the ticket queue (from ~48 hops) and the plan of thirds (from 256), with
    the block row on every byte alignment: vectors, pending samples and the state afterwards are those of the same context with the feed switched
    off (test hook 16), bit for bit, host and device blocks."""
    import torch
    N, H, C = 1024, 512, 13
    lengths = [1537, 30719, 2000, 3, 70001, 4097, 511, 140003, 1024, 25601]
    total = sum(lengths)
    rng = np.random.default_rng(len(fmt) + 77)
    base = signals.tone_vibrato_noise(C, 16, N, seed=5).reshape(C, -1)
    reps = (total + base.shape[1] - 1) // base.shape[1]
    x = (np.tile(base, (1, reps))[:, :total] * (0.6 + 0.4 * np.sin(np.arange(total) / 9973.0))[None, :]).astype(np.float32)
    x += rng.normal(0, 0.01, x.shape).astype(np.float32)
    if fmt == "s16":
        stream, per = np.clip(np.round(x * 32768.0), -32768, 32767).astype(np.int16), 1
    elif fmt == "f16":
        stream, per = x.astype(np.float16), 1
    elif fmt == "s24":
        stream, per = np.asarray(gpu_fx.pack_s24(np.clip(np.round(x * 8388608.0), -2 ** 23, 2 ** 23 - 1).astype(np.int32))), 3
    else:
        stream, per = x, 1
    sf = "s24" if fmt == "s24" else None

    def feed(an, device):
        raws, sms, at = [], [], 0
        for n in lengths:
            piece = np.ascontiguousarray(stream[:, per * at:per * (at + n)])
            r, s = an.push_samples(torch.from_numpy(piece).cuda() if device else piece, sample_format=sf)
            if hasattr(r, "cpu"):
                r, s = r.cpu().numpy(), s.cpu().numpy()
            raws.append(r); sms.append(s)
            at += n
        return np.concatenate(raws, 1), np.concatenate(sms, 1), an.pending_samples(), an.get_features()

    ref = gpu_fx.BatchAnalyser(C, N)
    ref.set_gain(0.75)
    ref.set_test_hooks(16)
    want = feed(ref, False)
    ref.close()
    assert want[0].shape[1] == total // H
    for device in ((False, True) if fmt != "s24" else (False,)):
        an = gpu_fx.BatchAnalyser(C, N)
        an.set_gain(0.75)
        got = feed(an, device)
        an.close()
        assert got[2] == want[2] == total % H
        assert same(got[0], want[0]) and same(got[1], want[1]) and same(got[3], want[3]), (fmt, device)


def test_the_applications_own_stepping_replayed_hop_by_hop(gpu_fx):
    """tests/golden/blocks/startup.npz: the reference's headers stepped the way the APPLICATION's threads step (one pass of the loop at thread
    start, one per audio callback; the reader running ahead of the writer as indexesOverlap lets it -- hops of zeros first, stale laps of the
    ring where half a window outlasts a device block).  fx_push_samples is a FIFO from the first real sample by design;
    a host that wants the app's sequence feeds fx_push_hops the hops tests/golden/collector_model.py lists, and gets the reference's vectors."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from collector_model import app_hops, gather
    from startup_cases import CASES, stream_of
    from oracle import fx_oracle as fo
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "blocks", "startup.npz"))
    for k, (name, N, C, total, block, order) in enumerate(CASES):
        stream = stream_of(N, C, total, seed=300 + k)
        hops = gather(stream, app_hops(N, block, total))
        an = gpu_fx.BatchAnalyser(C, N, order=order)
        parts = [an.push_hops(hops[:, t:t + 1]) for t in range(hops.shape[1])]          # one hop per call, as the app analyses them
        raw, sm = np.concatenate([p[0] for p in parts], 1), np.concatenate([p[1] for p in parts], 1)
        signals.assert_features_close(raw, g[name + "_raw"], 1e-5, fo.FEATURE_NAMES, name + " raw")
        signals.assert_features_close(sm, g[name + "_smoothed"], 1e-5, fo.FEATURE_NAMES, name + " smoothed")
        an.close()


@pytest.mark.parametrize("N", [1024, 2048, 4096])
@pytest.mark.parametrize("fmt", ["f32", "s16", "s24"])
def test_two_hop_calls_as_two_one_frame_launches_equal_the_batch_form_bitwise(gpu_fx, N, fmt):
    """Round 6: a call of exactly two hops per channel (a 1024-sample device buffer against a 1024-point window; 960-sample blocks every
    other call) runs as two one-frame launches over the same buffers -- fx_push_hops reads hop 1 / writes frame 1 of [C][2][..] in the second
    launch, fx_push_samples feeds the block to both and the second leaves the rest.  Against the batch kernels' two-frame form (test hook 32)
    and the re-blocked path (hook 16 | 32): vectors, pending samples and the state afterwards, bit for bit, for each of the three one-frame
    kernels, host and device memory."""
    import torch
    C, H = 9, N // 2
    total = 12 * H + 5
    x = signals.tone_vibrato_noise(C, 13, N, seed=N + 1).reshape(C, -1)[:, :total]
    if fmt == "s16":
        stream, per = np.clip(np.round(x * 32768.0), -32768, 32767).astype(np.int16), 1
    elif fmt == "s24":
        stream, per = np.asarray(gpu_fx.pack_s24(np.clip(np.round(x * 8388608.0), -2 ** 23, 2 ** 23 - 1).astype(np.int32))), 3
    else:
        stream, per = x, 1
    sf = "s24" if fmt == "s24" else None
    rng = np.random.default_rng(N)
    lengths, at = [], 0
    while at < total:                              # blocks of 1.5 .. 2.5 hops: calls complete one, two or three hops
        n = min(int(rng.integers(H + H // 2, 2 * H + H // 2)), total - at)
        lengths.append(n); at += n

    def feed_blocks_(an, device):
        raws, at = [], 0
        for n in lengths:
            piece = np.ascontiguousarray(stream[:, per * at:per * (at + n)])
            r, s = an.push_samples(torch.from_numpy(piece).cuda() if device else piece, sample_format=sf)
            raws.append((r.cpu().numpy(), s.cpu().numpy()) if hasattr(r, "cpu") else (r, s))
            at += n
        return np.concatenate([r for r, _ in raws], 1), np.concatenate([s for _, s in raws], 1), an.pending_samples(), an.get_features()

    def feed_pairs(an, device):
        hops = np.ascontiguousarray(stream[:, :per * 12 * H]).reshape(C, 12, per * H)
        if fmt == "s24":
            hops = gpu_fx.pack_s24(np.clip(np.round(x[:, :12 * H].reshape(C, 12, H) * 8388608.0), -2 ** 23, 2 ** 23 - 1).astype(np.int32))
        out = []
        for t in range(0, 12, 2):
            piece = hops[:, t:t + 2]
            r, s = an.push_hops(torch.from_numpy(np.ascontiguousarray(piece)).cuda() if (device and fmt != "s24") else piece)
            out.append((r.cpu().numpy(), s.cpu().numpy()) if hasattr(r, "cpu") else (r, s))
        return np.concatenate([r for r, _ in out], 1), np.concatenate([s for _, s in out], 1), an.get_features()

    ref = gpu_fx.BatchAnalyser(C, N); ref.set_gain(1.5); ref.set_test_hooks(16 | 32)
    want_blocks = feed_blocks_(ref, False); ref.close()
    ref = gpu_fx.BatchAnalyser(C, N); ref.set_gain(1.5); ref.set_test_hooks(32)
    want_pairs = feed_pairs(ref, False); ref.close()
    assert same(want_pairs[0], want_blocks[0][:, :12])            # the same stream either way
    for name, knobs in {"hop kernel": dict(one_hop_kernel=1), "frames + tails in one launch": dict(one_hop_kernel=0, hooks=8),
                        "frame kernel, then the tail kernel": dict(one_hop_kernel=0, hooks=4)}.items():
        for device in ((False, True) if fmt != "s24" else (False,)):
            an = gpu_fx.BatchAnalyser(C, N); an.set_gain(1.5)
            an.set_tuning(one_hop_kernel=knobs["one_hop_kernel"]); an.set_test_hooks(knobs.get("hooks", 0))
            got = feed_blocks_(an, device)
            assert got[2] == want_blocks[2] and all(same(g, w) for g, w in zip((got[0], got[1], got[3]), (want_blocks[0], want_blocks[1], want_blocks[3]))), (N, fmt, name, device, "blocks")
            an.close()
            an = gpu_fx.BatchAnalyser(C, N); an.set_gain(1.5)
            an.set_tuning(one_hop_kernel=knobs["one_hop_kernel"]); an.set_test_hooks(knobs.get("hooks", 0))
            got = feed_pairs(an, device)
            assert all(same(g, w) for g, w in zip(got, want_pairs)), (N, fmt, name, device, "two hops per call")
            an.close()
