"""Writes tests/golden/blocks/cases.npz (build container only: needs /root/reference).  `raw` / `smoothed` of every case are produced by the
REFERENCE'S OWN AudioDataCollector (ref AudioDataCollector.h:36-94: audioDeviceIOCallback with device blocks of any length, the ring, the gain at
read time, clearBuffer), overlapper and analysers, compiled unmodified against tools/refdiff/juce_standin.h and driven by
tools/refdiff/refdiff_blocks.cpp.  PARITY STAYS UNPINNED at the JUCE boundary (see make_golden.py); what these vectors pin is the COLLECTOR'S
semantics, whose code is entirely under /root/reference.

Run from the repo root:  python tests/golden/make_block_cases.py"""
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
from block_cases import CASES, stream_of  # noqa: E402


def main():
    out = {"source": "raw / smoothed: the reference's own AudioDataCollector + RealTimeAudioDataOverlapper + analysers (headers compiled unmodified against "
                     "tools/refdiff/juce_standin.h, tools/refdiff/refdiff_blocks.cpp), fed device blocks through audioDeviceIOCallback; log10(float) correctly rounded; "
                     "streams are regenerated from tests/golden/block_cases.py and CRC-checked"}
    for k, (name, N, C, hops, extra, block, order, events) in enumerate(CASES):
        stream = stream_of(name, N, C, hops, extra, seed=100 + k)
        raw, sm = refdiff.run_blocks(stream, N, block, order=order, events=events)
        assert raw.shape[1] == stream.shape[1] // (N // 2)
        out[name + "_raw"], out[name + "_smoothed"] = raw, sm
        out[name + "_crc"] = np.uint32(zlib.crc32(stream.tobytes()))
        print(name, raw.shape, "onsets", int(raw[:, :, 0].sum()))
    os.makedirs(os.path.join(HERE, "blocks"), exist_ok=True)
    np.savez_compressed(os.path.join(HERE, "blocks", "cases.npz"), **out)


if __name__ == "__main__":
    main()
