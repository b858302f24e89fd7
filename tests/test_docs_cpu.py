"""The documents' statements about the kernels as built are held to the library itself: DESIGN.md section 3.8 carries a table of
registers and scratch per kernel, and this test reads the same numbers out of feature-extractor_amd/lib/libfx_hip.so
(tools/kernel_resources.py: the gfx950 code objects' metadata).  A kernel change that moves a number fails here until the table is
regenerated (`python tools/kernel_resources.py --markdown`)."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"), reason="ROCm's llvm tools not installed")


def design_table():
    rows = {}
    for line in open(os.path.join(ROOT, "DESIGN.md")):
        m = re.match(r"\| `(fxk::[^`]+)` \| (\d+) \| (\d+) \|", line)
        if m:
            rows[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    return rows


def test_design_resource_table_is_the_library_as_built(fx):
    import kernel_resources as kr
    fx.load_library(build_if_missing=True)
    built = {k["pretty"]: (k["vgpr"], k.get("scratch", 0)) for k in kr.kernels_of() if k["pretty"].startswith(kr.DESIGN_PREFIXES)}
    doc = design_table()
    assert len(built) >= 30 and doc, "no kernels found / no table in DESIGN.md"
    assert doc == built, {k: (doc.get(k), built.get(k)) for k in set(doc) | set(built) if doc.get(k) != built.get(k)}
    # what the prose says about scratch is these rows and nothing else
    spills = sorted(k for k, (_, s) in built.items() if s)
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    for k in spills:
        assert ("`%s`" % k) in text
    assert "kernels with scratch: %d" % len(spills) in open(os.path.join(ROOT, "profiles", "r06_resources.txt")).read()


def test_no_frame_tail_kernel_below_1024_points(fx):
    """the 256- / 512-point instantiations of fx_frame_tail_kernel (88 / 96 B of scratch, reachable only through a test hook) are gone"""
    import kernel_resources as kr
    fx.load_library(build_if_missing=True)
    names = [k["pretty"] for k in kr.kernels_of()]
    assert "fxk::fx_frame_tail_kernel<1024, false>" in names
    assert not any(n in names for n in ("fxk::fx_frame_tail_kernel<256, false>", "fxk::fx_frame_tail_kernel<512, false>", "fxk::fx_frame_tail_kernel<256>", "fxk::fx_frame_tail_kernel<512>"))


def test_design_md_stays_a_design_document():
    """Round 5's review: DESIGN.md had grown into a 900-line lab notebook of 200-900-character lines.  It is the design as it stands, in at most
    300 lines of at most 120 columns; chronology lives in profiles/NOTEBOOK.md."""
    lines = open(os.path.join(ROOT, "DESIGN.md"), encoding="utf-8").read().splitlines()
    assert len(lines) <= 300, len(lines)
    long = [(i, len(l)) for i, l in enumerate(lines, 1) if len(l) > 120]
    assert not long, long
