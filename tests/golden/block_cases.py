"""The cases of tests/golden/blocks/cases.npz, shared by the generator (build container: the reference's own AudioDataCollector + analysers fed
with device blocks, tools/refdiff/refdiff_blocks.cpp) and the tests that replay them (CPU: the model on the oracle; GPU: fx_push_samples).
A case = a stream [C][total], a window, a device block length, an order mode and a list of events (at_sample, "gain", value) / (at_sample, "clear")
that take effect before the block that starts at or after at_sample."""
import numpy as np

import signals

CASES = [
    # name, N, C, hops (+ extra samples), block, order, events
    ("blocks_480_gain_clear_1024", 1024, 3, 24, 100, 480, 0, [(0, "gain", 0.5), (3000, "gain", 2.0), (5000, "clear"), (9000, "gain", 0.75)]),
    ("blocks_441_2048", 2048, 2, 12, 300, 441, 1, [(4410, "clear"), (6000, "gain", 1.5)]),
    ("blocks_1000_4096", 4096, 2, 8, 0, 1000, 0, [(7000, "gain", 0.25), (7000, "clear")]),
    ("blocks_63_512", 512, 2, 20, 17, 63, 2, [(1000, "gain", 3.0)]),
    # the runtime controls in mid-stream (ref RealTimeAnalyser.h:111-114, 244-258), blocks of exactly one hop so that every change falls on a hop boundary
    ("controls_1024", 1024, 3, 40, 0, 512, 0, [(0, "onset_type", 2), (11 * 512, "onset_window", 4), (11 * 512, "gain", 0.5), (17 * 512, "sensitivity", 0.2),
                                                 (23 * 512, "onset_type", 0), (23 * 512, "sample_rate", 44100.0), (30 * 512, "onset_window", 9), (33 * 512, "onset_type", 1)]),
]


def stream_of(name, N, C, hops, extra, seed):
    x = np.concatenate([signals.bursts(C, hops // 2, N, seed=seed), signals.tone_vibrato_noise(C, hops - hops // 2, N, seed=seed + 1)], axis=1).reshape(C, -1)
    if extra:
        x = np.concatenate([x, signals.tone_vibrato_noise(C, 1, N, seed=seed + 2)[:, 0, :extra]], axis=1)
    return np.ascontiguousarray(x, np.float32)


def replay(stream, N, block, events, push_block, set_gain, clear, control=None):
    """Drive a block consumer the way the generator drove the reference: events before the block they precede, then the block.  push_block(piece [C][n])
    returns the frames that block completed (or None); the concatenation is returned.  control(name, value) takes the analysers' runtime setters
    ("sensitivity", "onset_window", "onset_type", "sample_rate")."""
    total = stream.shape[1]
    pending = list(events)
    out = []
    for at in range(0, total, block):
        while pending and pending[0][0] <= at:
            e = pending.pop(0)
            if e[1] == "gain":
                set_gain(e[2])
            elif e[1] == "clear":
                clear()
            else:
                control(e[1], e[2])
        got = push_block(np.ascontiguousarray(stream[:, at:at + block]))
        if got is not None:
            out.append(got)
    return out
