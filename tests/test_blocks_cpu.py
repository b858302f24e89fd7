"""The collector's semantics, pinned by the reference's own collector: tests/golden/blocks/cases.npz holds what the reference's
AudioDataCollector + overlapper + analysers (headers compiled unmodified, tools/refdiff/refdiff_blocks.cpp) produce when they are fed device
blocks of any length with setGain / clearBuffer in between.  Here the MODEL that fx_push_samples implements -- a FIFO of raw samples per
channel, whole hops read as soon as they are there, the gain applied when a hop is read (ref AudioDataCollector.h:88), clearBuffer zeroing what
is pending and keeping the indices (:122) -- is replayed on the CPU oracle and must reproduce those vectors bit for bit.  (The GPU replays the
same cases through fx_push_samples in tests/test_gpu_samples.py.)"""
import os
import sys
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

from block_cases import CASES, replay, stream_of  # noqa: E402

FIXTURE = os.path.join(HERE, "golden", "blocks", "cases.npz")


def _same(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


@pytest.mark.parametrize("k", range(len(CASES)), ids=[c[0] for c in CASES])
def test_the_model_of_the_collector_reproduces_the_reference_collectors_vectors(oracle, k):
    g = np.load(FIXTURE)
    name, N, C, hops, extra, block, order, events = CASES[k]
    stream = stream_of(name, N, C, hops, extra, seed=100 + k)
    assert np.uint32(zlib.crc32(stream.tobytes())) == g[name + "_crc"], "the regenerated stream is not the one the fixture was made from"
    H = N // 2
    chans = [oracle.Channel(N, 48000.0, order) for _ in range(C)]
    pending = [np.zeros(0, np.float32) for _ in range(C)]
    raws, sms = [[] for _ in range(C)], [[] for _ in range(C)]

    def push_block(piece):
        for c in range(C):
            pending[c] = np.concatenate([pending[c], piece[c]])
            while pending[c].size >= H:
                r, s = chans[c].push_hops(pending[c][:H])          # (the oracle multiplies the hop by its current gain: the gain at READ time)
                raws[c].append(r[0]); sms[c].append(s[0])
                pending[c] = pending[c][H:]

    def set_gain(v):
        for ch in chans:
            ch.set_gain(v)

    def clear():
        for c in range(C):
            pending[c] = np.zeros_like(pending[c])                  # contents to zero, the count stays

    def control(name, value):
        for ch in chans:
            {"sensitivity": ch.set_onset_sensitivity, "onset_window": ch.set_onset_window, "onset_type": ch.set_onset_type, "sample_rate": ch.set_sample_rate}[name](value)

    replay(stream, N, block, events, push_block, set_gain, clear, control)
    raw, sm = np.array(raws), np.array(sms)
    assert raw.shape == g[name + "_raw"].shape
    assert _same(raw, g[name + "_raw"]) and _same(sm, g[name + "_smoothed"]), name
    assert all(p.size == stream.shape[1] % H for p in pending)


# ---- the application's own stepping (round 6): which samples its threads really read ----
from startup_cases import CASES as STARTUP_CASES, stream_of as startup_stream_of  # noqa: E402
from collector_model import app_hops, describe, gather  # noqa: E402

STARTUP = os.path.join(HERE, "golden", "blocks", "startup.npz")


@pytest.mark.parametrize("k", range(len(STARTUP_CASES)), ids=[c[0] for c in STARTUP_CASES])
def test_the_index_model_of_the_apps_threads_reproduces_the_reference_stepped_as_the_app_steps(oracle, k):
    """tests/golden/blocks/startup.npz: the reference's headers with their threads' loops run as the app runs them (once at start, once per audio
    callback).  The oracle, fed the hops tests/golden/collector_model.py says the app reads -- zeros first, stale laps where half a window
    outlasts a device block -- gives those vectors bit for bit: the model is what the app does, and fx_push_hops on those hops is how a host
    reproduces it."""
    g = np.load(STARTUP)
    name, N, C, total, block, order = STARTUP_CASES[k]
    stream = startup_stream_of(N, C, total, seed=300 + k)
    assert np.uint32(zlib.crc32(stream.tobytes())) == g[name + "_crc"]
    hops = gather(stream, app_hops(N, block, total))
    assert hops.shape[1] == g[name + "_raw"].shape[1]
    raw, sm = oracle.push_hops(hops, N, order=order)
    assert _same(raw, g[name + "_raw"]) and _same(sm, g[name + "_smoothed"]), name


def test_what_the_app_reads_at_its_defaults():
    """The statements DESIGN.md section 5 and INTEGRATION.md section 2 make about the app's start-up, held to the model."""
    assert describe(2048, 512)[:3] == (4, 0, "re-reads")          # the app's own window, a 512-sample device: 4 hops of zeros, then laps read twice
    assert describe(2048, 441)[:3] == (4, 0, "fifo")
    assert describe(1024, 512)[:3] == (8, 0, "fifo")              # a whole lap of zeros, then the stream one lap late
    assert describe(1024, 480)[:2] == (1, 512)                    # the first half window of the stream is never analysed
    # the canonical stepping (what fx_push_samples implements) analyses total // (N/2) hops; the app at its defaults twice as many
    n = describe(2048, 512, 200000)[3]
    assert 2 * (200000 // 1024) - 4 <= n <= 2 * (200000 // 1024) + 4
