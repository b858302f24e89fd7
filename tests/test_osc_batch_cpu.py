"""The sink at scale, host side (no GPU): fx_osc_encode_batch / fx_osc_message_bytes against the per-message encoder and the oracle's,
and the batch sender (fx_osc_sender: sendmmsg, sender threads, 60 Hz timer, primary + secondary target) against the counting receiver.
ref OSCFeatureAnalysisOutput.h:84-136 (message, timer, target parsing), AnalyserTrackController.h:22-23 (two senders per track),
MainComponent.cpp:170 (address "/Audio/A<row>")."""
import time

import numpy as np
import pytest


def _vectors(n, seed=0):
    v = np.random.default_rng(seed).standard_normal((n, 12)).astype(np.float32)
    v[0, 5] = np.inf                 # flatness of loud noise (inf is sent, not fixed)
    v[n // 2, 2] = np.nan            # getValue before the first insert
    v[-1, 8] = -np.inf
    v[-1, 0] = -0.0
    return v


@pytest.mark.parametrize("first", [0, 5, 95, 990, 9995, 99990, 65530, 9999995])
def test_batch_encoder_is_the_per_message_encoder(fx, oracle, first):
    """Every message of the batch = fx_osc_encode = the oracle's message, across the channel numbers where the address grows a digit
    (and the padded address a word: 76 -> 80 bytes at 1000, 84 at 10 000 000)."""
    capi = fx.capi
    C = 12
    v = _vectors(C, first)
    d, n = capi.osc_encode_batch("/Audio/A", first, v)
    assert d.shape == (C, capi.osc_message_bytes("/Audio/A", first + C - 1)) and d.shape[1] % 4 == 0
    for c in range(C):
        want = oracle.osc_message("/Audio/A%d" % (first + c), v[c])
        assert n[c] == len(want) == capi.osc_message_bytes("/Audio/A", first + c)
        assert bytes(d[c, :n[c]]) == want == fx.osc_encode("/Audio/A%d" % (first + c), v[c])
        assert not d[c, n[c]:].any()                    # the rest of the slot is zeros
    # a wider stride than needed is allowed
    d2, n2 = capi.osc_encode_batch("/Audio/A", first, v, stride=d.shape[1] + 8)
    assert np.array_equal(n, n2) and np.array_equal(d2[:, :d.shape[1]], d) and not d2[:, d.shape[1]:].any()


def test_batch_encoder_refuses_bad_arguments(fx):
    capi = fx.capi
    v = _vectors(4)
    assert capi.osc_message_bytes("/Audio/A", 0) == 76 and capi.osc_message_bytes("/Audio/A", 1000) == 80 and capi.osc_message_bytes("/a", 7) == 68
    assert capi.osc_message_bytes("/Audio/A", -1) == -1 and capi.osc_message_bytes("/" + "x" * 64, 0) == -1
    assert capi.osc_message_bytes("/" + "x" * 63, 0) == 68 + 64
    with pytest.raises(fx.FxError):
        capi.osc_encode_batch("/Audio/A", 998, v, stride=76)            # channel 1000 needs 80
    with pytest.raises(fx.FxError):
        capi.osc_encode_batch("/Audio/A", 0, v, stride=78)              # not a multiple of 4
    with pytest.raises(fx.FxError):
        capi.osc_encode_batch("/Audio/A", -1, v)
    d, n = capi.osc_encode_batch("/t", 0, np.zeros((0, 12), np.float32))
    assert d.shape[0] == 0 and n.shape == (0,)


@pytest.mark.parametrize("threads, gso", [(1, False), (3, False), (1, True), (4, True)])
def test_sender_delivers_every_message_of_every_tick_to_both_targets(fx, threads, gso):
    """2500 tracks (messages of 76 and 80 bytes), five ticks by hand: both receivers count 5 x 2500 well-formed messages and hold, for every
    track, exactly the bytes of the batch.  With gso the runs of equal length go out as segmented sends (or plainly, where the kernel
    refuses: the datagrams are the same either way)."""
    capi = fx.capi
    C = 2500
    v = _vectors(C, 3)
    d, n = capi.osc_encode_batch("/Audio/A", 0, v)
    rx = [capi.OscReceiver("127.0.0.1:0", threads=2, prefix="/Audio/A", keep_channels=C, gro=bool(k)) for k in range(2)]
    tx = capi.OscSender("127.0.0.1:%d" % rx[0].port, "127.0.0.1:%d" % rx[1].port, threads=threads, gso=gso)
    try:
        assert tx.send() == 0                                            # nothing published yet
        tx.update(d, n)
        assert [tx.send() for _ in range(5)] == [2 * C] * 5
        st = tx.stats()
        assert st["datagrams"] == 10 * C and st["dropped"] == 0 and st["ticks"] == 6 and st["late_ticks"] == 0
        deadline = time.time() + 5.0
        while time.time() < deadline and any(r.stats()["datagrams"] < 5 * C for r in rx):
            time.sleep(0.01)
        for r in rx:
            s = r.stats()
            assert s["datagrams"] == 5 * C and s["malformed"] == 0 and s["bytes"] == 5 * int(n.sum())
            for c in range(C):
                assert r.last(c) == bytes(d[c, :n[c]]), c
        # a new publication replaces the old one for the next tick
        v2 = _vectors(C, 4)
        d2, n2 = capi.osc_encode_batch("/Audio/A", 0, v2)
        tx.update(d2, n2)
        assert tx.send() == 2 * C
        time.sleep(0.2)
        assert rx[0].last(1234) == bytes(d2[1234, :n2[1234]]) and rx[1].last(7) == bytes(d2[7, :n2[7]])
    finally:
        tx.close()
        for r in rx:
            r.close()


def test_sender_timer_paces_60_hz(fx):
    """startTimerHz (60), ref OSCFeatureAnalysisOutput.h:133: ~60 ticks a second, each one every published message, none late."""
    capi = fx.capi
    C = 300
    d, n = capi.osc_encode_batch("/Audio/A", 0, _vectors(C))
    rx = capi.OscReceiver("127.0.0.1:0", prefix="/Audio/A", keep_channels=C)
    tx = capi.OscSender("127.0.0.1:%d" % rx.port, threads=2)
    try:
        tx.update(d, n)
        tx.start(60.0)
        time.sleep(1.0)
        tx.stop()
        st = tx.stats()
        assert 45 <= st["ticks"] <= 66, st              # (generous below: a loaded CI host may skip a tick; never more than 60 Hz gives)
        assert st["datagrams"] == st["ticks"] * C and st["dropped"] == 0
        assert st["late_ticks"] <= 6 and st["max_tick_ms"] < 100.0, st
        time.sleep(0.2)
        assert rx.stats()["datagrams"] == st["datagrams"]
        tx.start(200.0)                 # restartable, at another rate
        time.sleep(0.25)
        tx.stop()
        assert tx.stats()["ticks"] - st["ticks"] >= 30
    finally:
        tx.close()
        rx.close()


def test_sender_without_a_listener_counts_drops_and_survives(fx):
    """Nobody on the port: the ICMP answers come back as errors on the connected socket; datagrams are counted as dropped or sent, the
    sender neither blocks nor fails."""
    import socket
    capi = fx.capi
    s = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    d, n = capi.osc_encode_batch("/Audio/A", 0, _vectors(64))
    tx = capi.OscSender("127.0.0.1:%d" % port)
    try:
        tx.update(d, n)
        for _ in range(5):
            tx.send()
        st = tx.stats()
        assert st["datagrams"] + st["dropped"] == 5 * 64
    finally:
        tx.close()
    with pytest.raises(fx.FxError):
        capi.OscSender("not-an-address.invalid")
    # a host name is resolved, as juce::OSCSender::connect resolves one
    rx = capi.OscReceiver("127.0.0.1:0", prefix="/Audio/A", keep_channels=64)
    named = capi.OscSender("localhost:%d" % rx.port)
    named.update(d, n)
    assert named.send() == 64
    time.sleep(0.2)
    assert rx.stats()["datagrams"] == 64 and rx.last(63) == bytes(d[63, :n[63]])
    named.close(); rx.close()
    with pytest.raises(fx.FxError):
        capi.OscSender("127.0.0.1:9000", threads=0)
