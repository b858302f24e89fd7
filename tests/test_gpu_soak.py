"""The live path end to end under a clock (tools/live_soak.cpp): device blocks of 480 samples every 10 ms -> fx::LiveAnalyser (FIFO, worker,
pinned ring, blocks fed to the one-frame kernels) -> messages formed on the GPU -> fx::OSCBatchSender's 60 Hz timer -> a local receiver.
A few seconds here (the minutes-long runs are in profiles/): nothing dropped anywhere, every channel's newest datagram is the message of
the last published vector, and the vectors themselves -- after some hundreds of blocks that never line up with a hop -- are the oracle's
on the same stream (rebuilt here from the tool's integer formula).
ref AudioDataCollector.h:36-94, RealTimeAnalyser.h:141-234, OSCFeatureAnalysisOutput.h:84-136."""
import os
import struct
import subprocess

import numpy as np
import pytest

import signals

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(fx, tmp_path):
    exe = str(tmp_path / "live_soak")
    lib_dir = os.path.dirname(fx.library_path())
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "live_soak.cpp"), "-o", exe,
                           "-L", lib_dir, "-lfx_hip", "-Wl,-rpath," + lib_dir, "-pthread"])
    return exe


def _stream(channels, block, blocks, pool_blocks):
    """tools/live_soak.cpp's sample_value(), vectorised: [channels][blocks * block] floats"""
    i = (np.arange(blocks)[:, None] % pool_blocks * block + np.arange(block)[None, :]).reshape(-1).astype(np.uint64)
    out = np.empty((len(channels), i.size), np.float32)
    for k, c in enumerate(channels):
        step = 200 + 37 * (c % 97)
        phase = ((i * step) & 0xFFFF).astype(np.int64)
        tri = np.where(phase < 32768, phase, 65535 - phase) - 16384
        h = (np.uint64(c) * np.uint64(2654435761) + i * np.uint64(40503) + np.uint64(12345)) & np.uint64(0xFFFFFFFF)
        h ^= h >> np.uint64(15); h = (h * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF); h ^= h >> np.uint64(13)
        # C's tri / 2 truncates towards zero
        half = np.where(tri >= 0, tri // 2, -((-tri) // 2))
        v = half + (h >> np.uint64(20)).astype(np.int64) - 2048
        out[k] = v.astype(np.float32) / np.float32(32768.0)
    return out


@pytest.mark.parametrize("channels, window, block", [(300, 1024, 480), (64, 2048, 441)])
def test_live_soak_drops_nothing_and_matches_the_oracle(gpu_fx, oracle, tmp_path, channels, window, block):
    exe = _build(gpu_fx, tmp_path)
    dump = str(tmp_path / "soak.bin")
    p = subprocess.run([exe, "channels=%d" % channels, "window=%d" % window, "block=%d" % block, "seconds=3", "sender_threads=2", "dump=" + dump],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "live_soak: ok" in p.stdout, p.stdout + p.stderr
    assert "dropped at the FIFO 0," in p.stdout and "analysis / ring errors 0" in p.stdout and ", 0 dropped," in p.stdout, p.stdout
    raw = open(dump, "rb").read()
    C, N, B, blocks, pool = struct.unpack("5i", raw[:20])
    assert (C, N, B, pool) == (channels, window, block, 16) and blocks >= 5 * 48000 // block - 2
    smoothed = np.frombuffer(raw[20:], np.float32).reshape(C, 12)
    sample = sorted({c for c in (0, 1, 96, 97, C // 2, C - 1) if c < C})
    x = _stream(sample, B, blocks, pool)
    hops = x[:, :x.shape[1] // (N // 2) * (N // 2)].reshape(len(sample), -1, N // 2)
    for k, c in enumerate(sample):
        ch = oracle.Channel(N)
        _, osm = ch.push_hops(hops[k])
        signals.assert_features_close(smoothed[c][None, None], osm[-1][None, None], 1e-5, oracle.FEATURE_NAMES, "channel %d after %d hops" % (c, hops.shape[1]))
