"""fx_get_osc_datagrams: every channel's OSC feature message written ON THE DEVICE from the context's latest smoothed vectors
(ref OSCFeatureAnalysisOutput.h:89-113, message layout :107; address "/Audio/A<row>", MainComponent.cpp:170).  Bar: byte for byte
fx_osc_encode -- and the oracle's encoder on the oracle's own analysis -- of the same channel, incl. NaN, +-inf and channel numbers of
one to five digits; then the whole sink: analysis -> device-formed datagrams -> batch sender -> counting receiver."""
import time

import numpy as np
import pytest

import signals

pytestmark = pytest.mark.gpu


def _mixed_hops(C, T, N):
    """tones, loud noise (flatness = inf), silence (f0 slot 4.8, zeros), bursts: channel c takes class c mod 4"""
    parts = [signals.tone_vibrato_noise(C, T, N, seed=11), signals.loud_noise(C, T, N, seed=12), signals.silence(C, T, N), signals.bursts(C, T, N, seed=13)]
    hops = np.empty((C, T, N // 2), np.float32)
    for c in range(C):
        hops[c] = parts[c % 4][c]
    return hops


@pytest.mark.parametrize("first, C", [(0, 1100), (95, 20), (9990, 37), (65530, 6), (99990, 20)])
def test_device_datagrams_equal_the_host_encoder_bytewise(gpu_fx, oracle, first, C):
    N, T = 1024, 12
    an = gpu_fx.BatchAnalyser(C, N)
    hops = _mixed_hops(C, T, N)
    _, sm = an.push_hops(hops)
    latest = an.get_features()
    assert np.array_equal(latest, sm[:, -1], equal_nan=True)
    assert np.isinf(latest[1::4]).any()                      # the loud-noise channels really carry an inf into the messages
    d, n = an.osc_datagrams("/Audio/A", first)
    assert d.shape == (C, gpu_fx.capi.osc_message_bytes("/Audio/A", first + C - 1))
    want_d, want_n = gpu_fx.capi.osc_encode_batch("/Audio/A", first, latest)
    assert np.array_equal(n, want_n) and np.array_equal(d, want_d)
    for c in list(range(0, C, max(1, C // 16))) + [C - 1]:
        assert bytes(d[c, :n[c]]) == gpu_fx.osc_encode("/Audio/A%d" % (first + c), latest[c])
    # ... and against the oracle end to end, on channels the oracle can analyse in a moment
    for c in range(min(C, 4)):
        ch = oracle.Channel(N)
        _, osm = ch.push_hops(hops[c])
        assert bytes(d[c, :n[c]]) == oracle.osc_message("/Audio/A%d" % (first + c), osm[-1])
    an.close()


def test_device_datagrams_nan_slots_other_prefixes_and_device_destination(gpu_fx):
    """A spectral-only context leaves the harmonic slots at getValue's 0/0 = NaN (RealTimeAnalyser.h:84-88): the NaN's bits travel
    unchanged.  Prefixes of other lengths move the padding; a wider stride leaves zeros; a device buffer (torch) receives the same bytes."""
    import torch
    N, C, T = 2048, 64, 6
    an = gpu_fx.BatchAnalyser(C, N, analysers="spectral")
    an.push_hops(signals.tone_vibrato_noise(C, T, N, seed=3))
    latest = an.get_features()
    assert np.isnan(latest[:, gpu_fx.F0]).all()
    for prefix in ("/a", "/Audio/A", "/Feature-Extractor/Track/", "/" + "x" * 63):
        for stride_extra in (0, 12):
            stride = gpu_fx.capi.osc_stride(prefix, 990, C) + stride_extra
            d, n = an.osc_datagrams(prefix, 990, stride=stride)
            want_d, want_n = gpu_fx.capi.osc_encode_batch(prefix, 990, latest, stride=stride)
            assert np.array_equal(n, want_n) and np.array_equal(d, want_d), (prefix, stride_extra)
    # device destination: asynchronous on the context's stream
    stride = gpu_fx.capi.osc_stride("/Audio/A", 0, C)
    out = torch.full((C, stride), 0xEE, dtype=torch.uint8, device="cuda:0")
    lengths = np.empty(C, np.int32)
    import ctypes
    gpu_fx.capi.check(an._lib.fx_get_osc_datagrams(an._h, b"/Audio/A", 0, ctypes.c_void_p(out.data_ptr()), stride, lengths.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), gpu_fx.capi.MEM_DEVICE))
    an.sync()
    want_d, want_n = gpu_fx.capi.osc_encode_batch("/Audio/A", 0, latest)
    assert np.array_equal(out.cpu().numpy(), want_d) and np.array_equal(lengths, want_n)
    # bad arguments are refused, not truncated
    with pytest.raises(gpu_fx.FxError):
        an.osc_datagrams("/Audio/A", 990, stride=76)            # channel 1000 needs 80
    with pytest.raises(gpu_fx.FxError):
        an.osc_datagrams("/Audio/A", -1)
    with pytest.raises(gpu_fx.FxError):
        an.osc_datagrams("/" + "x" * 64, 0)
    an.close()


def test_analysis_to_wire_through_the_batch_sender(gpu_fx, oracle):
    """The whole sink at one shard's width: 8192 channels analysed, their messages formed on the device, published to the batch sender,
    three ticks to a primary and a secondary receiver; every channel's newest datagram at both is the host encoder's message of that
    channel's latest vector, and a sample of channels equals the oracle's message of the oracle's analysis."""
    sharded = __import__("importlib").import_module("feature-extractor_amd.sharded")
    N, C, T = 1024, 8192, 4
    hops = gpu_fx.synth.hops(C, T, N)
    an = gpu_fx.BatchAnalyser(C, N)
    an.push_hops(hops)
    latest = an.get_features()
    rx = [gpu_fx.capi.OscReceiver("127.0.0.1:0", threads=2, prefix="/Audio/A", keep_channels=C) for _ in range(2)]
    sink = sharded.OscSink(None, "127.0.0.1:%d" % rx[0].port, "127.0.0.1:%d" % rx[1].port, threads=2, gso=True)
    try:
        sink.update_datagrams(*an.osc_datagrams("/Audio/A", 0))
        assert [sink.send() for _ in range(3)] == [2 * C] * 3
        deadline = time.time() + 5.0
        while time.time() < deadline and any(r.stats()["datagrams"] < 3 * C for r in rx):
            time.sleep(0.01)
        for r in rx:
            st = r.stats()
            assert st["datagrams"] == 3 * C and st["malformed"] == 0, st
            for c in range(C):
                assert r.last(c) == gpu_fx.osc_encode("/Audio/A%d" % c, latest[c]), c
        for c in (0, 999, 1000, 8191):
            ch = oracle.Channel(N)
            _, osm = ch.push_hops(hops[c])
            assert rx[0].last(c) == oracle.osc_message("/Audio/A%d" % c, osm[-1])
    finally:
        sink.close()
        for r in rx:
            r.close()
        an.close()
