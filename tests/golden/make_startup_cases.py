"""Writes tests/golden/blocks/startup.npz (build container only: needs /root/reference): what the reference's analysers produce when they are
stepped the way the APPLICATION steps them -- one pass of each thread's loop when it is started, before any audio, and one per audio
callback (tools/refdiff/refdiff_blocks.cpp --notify-per-block; AnalyserTrackController.h:184-185, RealTimeAnalyser.h:141-177 / :201-234,
AudioDataCollector.h:68-105).  The hops the app then analyses are NOT the stream cut at whole hops: zeros first, and where half a window is
longer than a device block, audio from a lap of the ring ago (tests/golden/collector_model.py lists them).  PARITY STAYS UNPINNED at the JUCE
boundary (see make_golden.py); what these vectors pin is which samples the app's threads read, whose code is entirely under /root/reference.

Run from the repo root:  python tests/golden/make_startup_cases.py"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "tools", "refdiff"))

import refdiff  # noqa: E402
from startup_cases import CASES, stream_of  # noqa: E402
from collector_model import app_hops  # noqa: E402


def main():
    out = {"source": "raw / smoothed: the reference's own AudioDataCollector + RealTimeAudioDataOverlapper + analysers (headers compiled unmodified against "
                     "tools/refdiff/juce_standin.h), stepped as the application's threads step (tools/refdiff/refdiff_blocks.cpp --notify-per-block); "
                     "log10(float) correctly rounded; streams are regenerated from tests/golden/startup_cases.py and CRC-checked"}
    for k, (name, N, C, total, block, order) in enumerate(CASES):
        stream = stream_of(N, C, total, seed=300 + k)
        raw, sm = refdiff.run_blocks(stream, N, block, order=order, app_stepping=True)
        hops = app_hops(N, block, total)
        assert raw.shape[1] == len(hops), (name, raw.shape, len(hops))
        out[name + "_raw"], out[name + "_smoothed"] = raw, sm
        out[name + "_crc"] = np.uint32(zlib.crc32(stream.tobytes()))
        print(name, raw.shape, "hops analysed", len(hops), "of", total // (N // 2), "whole hops in the stream")
    np.savez_compressed(os.path.join(HERE, "blocks", "startup.npz"), **out)


if __name__ == "__main__":
    main()
