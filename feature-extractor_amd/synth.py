"""Synthetic multichannel audio used by bench.py and the tests.

x[c][n] = 0.4 sin(p) + 0.2 sin(2p) + 0.1 sin(3p) + u,  p = 2*pi*f_c*n/sr,
f_c = 55 * 2^((c mod 72)/12) Hz,  u uniform(-0.05, 0.05) from splitmix64(0x5EED ^ c<<32 ^ n).
The same bytes feed the GPU path and the CPU baseline.
"""
import os

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def channel_frequency(c):
    return 55.0 * 2.0 ** ((np.asarray(c) % 72) / 12.0)


def _workers():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def samples(channels, num_samples, sample_rate=48000.0, first_channel=0, first_sample=0, dtype=np.float32):
    """[channels][num_samples] synthetic stream, channel ids first_channel .. first_channel+channels-1.
    Generated in channel blocks so the fp64 temporaries stay small; the blocks are independent and numpy
    releases the GIL inside its loops, so they are spread over a few threads (same bytes either way)."""
    out = np.empty((channels, num_samples), dtype)
    n = np.arange(first_sample, first_sample + num_samples, dtype=np.uint64)[None, :]
    nf = n.astype(np.float64)
    block = max(1, min(channels, (1 << 21) // max(1, num_samples)))

    def fill(c0):
        c1 = min(channels, c0 + block)
        c = np.arange(first_channel + c0, first_channel + c1, dtype=np.uint64)[:, None]
        with np.errstate(over="ignore"):
            key = np.uint64(0x5EED) ^ (c << np.uint64(32)) ^ n
        u = (splitmix64(key) >> np.uint64(11)).astype(np.float64) * (2.0 ** -53) * 0.1 - 0.05
        phase = 2.0 * np.pi * channel_frequency(c.astype(np.float64)) * nf / sample_rate
        out[c0:c1] = (0.4 * np.sin(phase) + 0.2 * np.sin(2 * phase) + 0.1 * np.sin(3 * phase) + u).astype(dtype)

    starts = list(range(0, channels, block))
    workers = min(_workers(), len(starts))
    if workers <= 1:
        for c0 in starts:
            fill(c0)
    else:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(workers) as pool:
            list(pool.map(fill, starts))
    return out


def hops(channels, num_hops, window_size, sample_rate=48000.0, first_channel=0, first_hop=0, dtype=np.float32):
    """[channels][num_hops][window_size/2] consecutive hops of the stream."""
    h = window_size // 2
    x = samples(channels, num_hops * h, sample_rate, first_channel, first_hop * h, dtype)
    return x.reshape(channels, num_hops, h)


def frames(channels, num_frames, window_size, sample_rate=48000.0, first_channel=0, dtype=np.float32):
    """[channels][num_frames][window_size] pre-assembled 50%-overlap windows; frame t covers
    samples [(t-1)*N/2, (t+1)*N/2) of the stream with the stream silent before sample 0
    (what RealTimeAudioDataOverlapper yields, ref RealTimeAudioAnalysis.h:202,205-219)."""
    h = window_size // 2
    x = samples(channels, num_frames * h, sample_rate, first_channel, 0, dtype)
    x = np.concatenate([np.zeros((channels, h), dtype), x], axis=1)
    out = np.empty((channels, num_frames, window_size), dtype)
    for t in range(num_frames):
        out[:, t, :] = x[:, t * h: t * h + window_size]
    return out
