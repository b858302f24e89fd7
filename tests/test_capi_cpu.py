"""CPU-side checks of the C ABI: the library builds for gfx950, loads, exports every symbol
include/fx.h declares, refuses to run without a GPU (no CPU fallback), and its host-only OSC helpers
match the oracle."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "fx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fx_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_what_the_binding_expects(fx):
    from importlib import import_module
    capi = import_module("feature-extractor_amd.capi")
    assert declared_symbols() == sorted(capi.EXPORTS)


def test_library_builds_loads_and_exports_every_symbol(fx):
    lib = fx.load_library()
    assert os.path.exists(fx.library_path())
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.fx_abi_version() == fx.capi.ABI_VERSION == 6


def test_library_contains_gfx950_code(fx):
    fx.load_library()
    blob = open(fx.library_path(), "rb").read()
    assert b"gfx950" in blob
    assert b"fx_frame_kernel" in blob


def _has_gpu():
    import torch
    return torch.cuda.is_available()


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU error path")
def test_no_cpu_fallback_without_device(fx):
    with pytest.raises(fx.FxError) as e:
        fx.BatchAnalyser(4, 1024)
    assert e.value.code == 2                    # FX_ERR_NO_DEVICE


def test_argument_validation_happens_before_device_use(fx):
    lib = fx.load_library()
    h = ctypes.c_void_p()
    assert lib.fx_create(ctypes.byref(h), 0, 4, 1000, 48000.0, 0) == 1      # not a power of two
    assert b"window_size" in lib.fx_last_error()
    assert lib.fx_create(ctypes.byref(h), 0, 0, 1024, 48000.0, 0) == 1
    assert lib.fx_create(ctypes.byref(h), 0, 4, 8192, 48000.0, 0) == 1
    assert lib.fx_create(None, 0, 4, 1024, 48000.0, 0) == 1
    assert lib.fx_create(ctypes.byref(h), 0, 4, 1024, 48000.0, 4 | 8) == 1       # spectral-only AND harmonic-only
    assert lib.fx_create(ctypes.byref(h), 0, 4, 1024, 48000.0, 3) == 1           # order 3 does not exist
    assert lib.fx_push_hops(None, None, 1, 0, 0, None, None) == 1
    assert lib.fx_sync(None) == 1


def test_osc_helpers_match_oracle(fx, oracle):
    v = np.random.default_rng(0).standard_normal(12).astype(np.float32)
    assert fx.osc_encode("/Audio/A0", v) == oracle.osc_message("/Audio/A0", v)
    assert len(fx.osc_encode("/Audio/A0", v)) == 76
    assert fx.osc_encode("/Audio/A13", v) == oracle.osc_message("/Audio/A13", v)
    o12 = fx.pack_osc12(v)
    assert np.array_equal(o12, v[[0, 1, 2, 3, 8, 4, 5, 6, 7, 9, 10, 11]])
    assert np.array_equal(fx.pack_osc10(v), v[[0, 1, 2, 3, 8, 4, 5, 7, 9, 11]])     # README.md:57 order


def test_synth_is_deterministic_and_frames_match_hops(fx):
    a = fx.synth.hops(3, 5, 1024, first_channel=7)
    b = fx.synth.hops(3, 5, 1024, first_channel=7)
    assert np.array_equal(a, b) and a.dtype == np.float32
    c = fx.synth.hops(1, 5, 1024, first_channel=8)
    assert np.array_equal(a[1], c[0])            # a channel's samples do not depend on the shard it is in
    fr = fx.synth.frames(3, 5, 1024, first_channel=7)
    assert np.array_equal(fr[:, 0, :512], np.zeros((3, 512), np.float32))
    assert np.array_equal(fr[:, 2, :512], a[:, 1]) and np.array_equal(fr[:, 2, 512:], a[:, 2])


def test_this_hosts_twiddle_table_has_the_symmetries_the_kernels_use(fx):
    """fx_twiddle_symmetry (host arithmetic, no GPU): the reference's float twiddle table as this host's cos / sin produce it.
    Bit 0 (the 16-point first pass's mirrored constants) is required -- fx_create refuses a host without it; bit 1 (quarter
    turns in the 4096-point table) only selects the 4096-point kernel's fast path, but a host that loses it should be noticed."""
    lib = fx.load_library(build_if_missing=True)
    for n in (256, 512, 1024, 2048, 4096):
        assert lib.fx_twiddle_symmetry(n) == 3, n
    assert lib.fx_twiddle_symmetry(1000) == 0 and lib.fx_twiddle_symmetry(8192) == 0


def test_work_unit_plans_cover_every_frame_once(fx, monkeypatch):
    """fx_plan_units (host arithmetic, no GPU, no environment): however a call is cut into work units for the frame kernel,
    the unit lengths are positive, add up to the call's frames per channel, fit the kernel's table, and only the last unit
    may be a partial round of wavefronts; windows of 2048 / 4096 points and the spectral analyser alone are never cut."""
    from importlib import import_module
    capi = import_module("feature-extractor_amd.capi")
    fx.load_library(build_if_missing=True)
    cap = capi.MAX_UNITS
    # the planner is pure: variables in the environment must not reach it (only fx_create reads them, once)
    monkeypatch.setenv("FX_FRAMES_PER_CHUNK", "0")
    monkeypatch.setenv("FX_CHUNK_PLAN", "1,2,3")

    def plan(N, flags, k, T, tuning=None):
        return capi.plan_units(N, flags, k, T, tuning)

    for N, k in ((256, 8), (512, 8), (1024, 8), (1024, 3), (2048, 4), (4096, 7)):
        for flags in (0, 4, 8):                       # both analysers, FX_SPECTRAL_ONLY, FX_HARMONIC_ONLY
            for T in list(range(1, 700)) + [1000, 4096, 5000, 65536, 1000000]:
                sizes = plan(N, flags, k, T)
                assert 1 <= len(sizes) <= cap and sum(sizes) == T and min(sizes) >= 1, (N, flags, k, T, sizes)
                assert all(s % k == 0 for s in sizes[:-1]), (N, flags, k, T, sizes)
                if N > 1024 or flags == 4:
                    assert sizes == [T]
    assert plan(1024, 0, 8, 512) == [168, 112, 80, 48, 32, 24, 16, 16, 16]      # the bench shape
    assert plan(1024, 0, 8, 128) == [64, 64] and plan(1024, 0, 8, 100) == [56, 44] and plan(1024, 0, 8, 90) == [90]
    # the knobs arrive in a struct fx_tuning (a context takes its own from the environment once, in fx_create)
    t = capi.Tuning.defaults()
    assert plan(1024, 0, 8, 512, t) == plan(1024, 0, 8, 512)
    t.frames_per_unit = 0
    assert plan(1024, 0, 8, 512, t) == [512]
    t.frames_per_unit = 16
    forced = plan(4096, 0, 7, 129, t)         # the override applies to every window size (tests force cut launches with it)
    assert len(forced) >= 4 and sum(forced) == 129 and all(v % 7 == 0 for v in forced[:-1])
    assert plan(4096, 0, 7, 100, t) == [21, 21, 21, 21, 16]
    t = capi.Tuning.defaults().set_plan([300, 200, 12])
    assert plan(1024, 0, 8, 512, t) == [300, 200, 12] and plan(1024, 0, 8, 511, t) != [300, 200, 12]


def test_tuning_comes_from_the_environment_once(fx, monkeypatch):
    """fx_tuning_from_env is the one reader of the FX_* variables (fx_create calls it once per context); unset variables
    leave every knob at "measured best"."""
    from importlib import import_module
    capi = import_module("feature-extractor_amd.capi")
    fx.load_library(build_if_missing=True)
    for var in ("FX_WAVES", "FX_CHANNELS_PER_WG", "FX_WAVES_PER_FRAME", "FX_FRAMES_PER_CHUNK", "FX_CHUNK_PLAN", "FX_STREAM_GRAPH",
                "FX_STREAM_HOP_KERNEL", "FX_STREAM_ZEROCOPY", "FX_ONE_HOP_KERNEL", "FX_CALL_TIMING", "FX_HANDOVER_SPINS", "FX_STREAM_FILL_STREAMING"):
        monkeypatch.delenv(var, raising=False)
    d, e = capi.Tuning.defaults(), capi.Tuning.from_env()
    assert bytes(d) == bytes(e)
    assert (d.waves_per_channel, d.frames_per_unit, d.stream_graph, d.one_hop_kernel, d.call_timing, d.handover_spin_limit) == (0, -1, -1, -1, -1, 0)
    monkeypatch.setenv("FX_WAVES", "3")
    monkeypatch.setenv("FX_FRAMES_PER_CHUNK", "0")
    monkeypatch.setenv("FX_CHUNK_PLAN", "300,200,12")
    monkeypatch.setenv("FX_STREAM_HOP_KERNEL", "0")
    monkeypatch.setenv("FX_HANDOVER_SPINS", "64")
    e = capi.Tuning.from_env()
    assert (e.waves_per_channel, e.frames_per_unit, e.stream_hop_kernel, e.handover_spin_limit) == (3, 0, 0, 64)
    assert list(e.unit_plan[:e.unit_plan_len]) == [300, 200, 12]
