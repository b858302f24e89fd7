"""What the reference APPLICATION's analysis threads read out of AudioDataCollector's ring, as indices (no audio, no reference code): a model
of ref AudioDataCollector.h:36-105 (a 4096-sample ring, writeIndex advanced by every audio callback, readIndex by every getAnalysisBuffer, the
reader spinning while indexesOverlap) driven the way the app drives it (RealTimeAnalyser.h:141-177, :201-234: the threads' loop runs once when
they are started -- AnalyserTrackController.h:184-185 -- and once per notify(); juce::Thread::notify is an auto-reset event).

indexesOverlap lets the reader run AHEAD of the writer whenever readIndex >= writeIndex + expectedSamplesPerBlock, so the app
  * analyses hops of zeros (the ring's initial contents) before any audio has arrived,
  * may never analyse the first samples of the stream (the reader started beyond them),
  * and, where half a window is longer than a device block (the app's default: 2048-point windows, 512-sample blocks), keeps getting ahead and
    analyses audio from a lap of the ring ago, again and again.
fx_push_samples does none of that: it is a FIFO from the first real sample (SURVEY.md 8c prescribes that stepping).  A
host that wants the app's sequence bit for bit feeds fx_push_hops the hops this model lists (tests/test_gpu_samples.py does, against vectors
made by the reference's own headers stepped this way: tools/refdiff/refdiff_blocks.cpp --notify-per-block)."""
import numpy as np

RING = 4096


def app_hops(window_size, block, total_samples, expected_samples_per_block=None):
    """The hops the app's threads analyse when a stream of total_samples arrives in callbacks of `block` samples (the last one shorter).
    -> list of int64 arrays [window_size/2]: for every sample of a hop the index of the stream sample the ring held there, -1 = a zero that
    was never written."""
    H = window_size // 2
    expected = block if expected_samples_per_block is None else expected_samples_per_block
    ring = np.full(RING, -1, np.int64)
    w = r = 0
    hops = []
    signalled, spinning = False, True                # thread start: the loop's first pass goes straight into getNextBuffer

    def overlap():
        if w < r and r < w + expected:
            return True
        if r < w and w < r + H:
            return True
        return False

    def run():
        nonlocal r, signalled, spinning
        while True:
            if spinning:
                if overlap():
                    return
                hops.append(ring[(r + np.arange(H)) % RING].copy())
                r = (r + H) % RING
                spinning = False
            if not signalled:
                return
            signalled, spinning = False, True

    run()
    at = 0
    while at < total_samples:
        n = min(block, total_samples - at)
        ring[(w + np.arange(n)) % RING] = at + np.arange(n)
        w = (w + n) % RING
        at += n
        signalled = True
        run()
    return hops


def describe(window_size, block, total_samples=200000):
    """(leading all-zero hops, first stream sample analysed, 'fifo' | 're-reads') of the app for this window and block length"""
    hops = app_hops(window_size, block, total_samples)
    H = window_size // 2
    zeros = 0
    while zeros < len(hops) and (hops[zeros] < 0).all():
        zeros += 1
    rest = hops[zeros:]
    first = int(rest[0][0]) if rest else -1
    fifo = all((h == first + k * H + np.arange(H)).all() for k, h in enumerate(rest))
    return zeros, first, "fifo" if fifo else "re-reads", len(hops)


def gather(stream, hop_indices):
    """the hops themselves: stream [C][total] -> [C][len(hop_indices)][H], zeros where the ring had never been written"""
    idx = np.stack(hop_indices)
    out = stream[:, np.maximum(idx, 0)]
    out[:, idx < 0] = 0.0
    return np.ascontiguousarray(out, np.float32)


if __name__ == "__main__":
    print("window  block  hops analysed  leading zero hops  first sample analysed  then")
    for N in (1024, 2048, 4096):
        for block in (441, 480, 512, 1024):
            z, first, kind, n = describe(N, block)
            print("%6d  %5d  %13d  %17d  %21d  %s" % (N, block, n, z, first, kind))
