"""The compile-time switches left in csrc/ cannot rot: every one of them still compiles for gfx950 (hipcc -fsyntax-only, no GPU),
there are few of them, and the public header carries no diagnostics."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "feature-extractor_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# what each remaining switch is for
SWITCHES = {
    "FX_PART": "which kernels an object of the library holds (build.py compiles fx_kernels.hip three times with different options)",
    "FX_WITH_TAIL_KERNELS": "set by fx_kernels.hip from FX_PART: the object that holds the __global__ tail kernels",
    "FX_EXP_WIDE_BAND": "test build: the logRMS bracket catches (almost) every frame, so the exact gate threshold runs everywhere",
    "FX_EXP_STOP_AT": "costing builds (tools/section_costs.sh): a frame's work ends at stop point k",
    "FX_EXP_FMA_TWIDDLES": "experiment build (round 5): the twiddle products' second multiply fused into the sum -- spectra no longer the reference's; "
                           "measures what FMA is worth on the chip (profiles/r05_fma_experiment.txt), never shipped",
}

VARIANTS = ([("part%d" % k, ["-DFX_PART=%d" % k]) for k in range(4)]
            + [("wide_band", ["-DFX_EXP_WIDE_BAND"])]
            + [("fma_twiddles", ["-DFX_EXP_FMA_TWIDDLES"])]
            + [("stop_at_%d" % k, ["-DFX_EXP_STOP_AT=%d" % k]) for k in range(1, 12)])


def switches_in_sources():
    found = set()
    for path in glob.glob(os.path.join(CSRC, "*")):
        for line in open(path, errors="replace"):
            m = re.match(r"\s*#\s*(?:if|ifdef|ifndef|elif)\b(.*)", line)
            if not m:
                continue
            for name in re.findall(r"\b(FX_[A-Z0-9_]+)\b", m.group(1)):
                if not name.endswith("_H"):
                    found.add(name)
    return found


def test_few_switches_and_each_one_is_known():
    found = switches_in_sources()
    assert found == set(SWITCHES), "undocumented or stale switches: %r" % sorted(found ^ set(SWITCHES))
    assert len(found) <= 8


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("name,flags", VARIANTS, ids=[v[0] for v in VARIANTS])
def test_variant_compiles(name, flags):
    cmd = [HIPCC, "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fsyntax-only", "-Wno-unused-command-line-argument"] + flags + [
        "-x", "hip", os.path.join(CSRC, "fx_kernels.hip")]
    p = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp")
    assert p.returncode == 0 and "error:" not in p.stderr, p.stderr[-2000:]


def test_public_header_is_free_of_diagnostics():
    text = open(os.path.join(ROOT, "include", "fx.h")).read()
    for word in ("stamp", "FX_PAIR_STAMPS", "fx_debug_", "debug_flags", "fx_set_tuning_internal", "FX_HOOK_"):
        assert word not in text, word
    # the test hooks live in csrc/fx_kernels.h (fx_set_tuning_internal), which no host includes
    assert "fx_set_tuning_internal" in open(os.path.join(CSRC, "fx_kernels.h")).read()
