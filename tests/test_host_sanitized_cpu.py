"""The host shim under sanitizers, with every HIP call site failed once (CPU build only: GPU sanitizers are not available).

csrc/fx_capi.cpp and csrc/fx_comm.cpp -- ring slots, captured steps, the fill pool's threads, the re-blocking plumbing of fx_push_samples,
the RCCL gather -- are compiled UNCHANGED for the host against tests/cpp/fake_hip/ (a malloc-backed hip_runtime.h whose every call can be
made to fail, launch stubs for csrc/fx_kernels.h, a one-rank librccl.so.1) with -fsanitize=address,undefined, and tests/cpp/host_sanitize.cpp
walks six scenarios once per HIP call with that call failing: no crash, no overrun, no leak, no wedged ring, the next call works; a seventh checks
fx_push_samples' arithmetic on the host (random block lengths x formats x windows: the hops handed to the kernels, put end to end, are the stream).  A second
build with -fsanitize=thread runs the fill pool (1 .. 64 threads, resized up and down, jobs back to back).
What this found when it was written (round 5): grow() freed a scratch buffer twice when hipFree itself reported a failure."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "feature-extractor_amd", "csrc")
FAKE = os.path.join(ROOT, "tests", "cpp", "fake_hip")

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not installed")


def _build(tmp, sanitizer, exe):
    out = os.path.join(tmp, exe)
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=" + sanitizer, "-fno-omit-frame-pointer", "-Wno-tsan",
           "-I", FAKE, "-I", os.path.join(ROOT, "include"), "-I", CSRC,
           os.path.join(CSRC, "fx_capi.cpp"), os.path.join(CSRC, "fx_comm.cpp"), os.path.join(FAKE, "fake_hip.cpp"),
           os.path.join(ROOT, "tests", "cpp", "host_sanitize.cpp"), "-o", out, "-ldl", "-lpthread"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    return out


@pytest.fixture(scope="module")
def fake_rccl(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("fake_rccl"))
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-I", FAKE, os.path.join(FAKE, "fake_rccl.cpp"), "-o", os.path.join(d, "librccl.so.1")])
    return d


def _run(exe, mode, lib_dir, extra_env=None):
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = lib_dir               # (no ROCm directory: fx_comm.cpp's dlopen must find the one-rank fake, and nothing else)
    env.pop("ROCM_PATH", None); env.pop("ROCM_HOME", None)
    env["ASAN_OPTIONS"] = "detect_leaks=1:abort_on_error=0"
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    env["TSAN_OPTIONS"] = "halt_on_error=1"
    env.update(extra_env or {})
    return subprocess.run([exe, mode], capture_output=True, text=True, env=env, timeout=900)


def test_every_hip_call_site_failed_once_under_asan_and_ubsan(tmp_path, fake_rccl):
    exe = _build(str(tmp_path), "address,undefined", "host_asan")
    p = _run(exe, "asan", fake_rccl)
    tail = (p.stdout + p.stderr)[-4000:]
    assert p.returncode == 0, tail
    assert "host_sanitize: 0 problem(s)" in p.stdout, tail
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, tail
    # every scenario was walked, and the walk is not vacuous: hundreds of call sites, all but the destroy / timing paths reported
    walked = [line for line in p.stdout.splitlines() if "injected failures walked" in line]
    assert len(walked) == 6, p.stdout
    total = sum(int(line.split("run:")[1].split()[0]) for line in p.stdout.splitlines() if "clean run:" in line)
    assert total > 800, p.stdout
    # and the arithmetic of fx_push_samples on the host: the hops handed to the (fake) kernels, end to end, are the stream cut at whole hops
    assert any(line.startswith("block arithmetic") and "clean run" in line for line in p.stdout.splitlines()), p.stdout


def test_fill_pool_under_tsan(tmp_path, fake_rccl):
    exe = _build(str(tmp_path), "thread", "host_tsan")
    p = _run(exe, "tsan", fake_rccl)
    tail = (p.stdout + p.stderr)[-4000:]
    if "FATAL: ThreadSanitizer: unexpected memory mapping" in p.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow in this container")
    assert p.returncode == 0, tail
    assert "WARNING: ThreadSanitizer" not in p.stderr, tail
    assert "host_sanitize: 0 problem(s)" in p.stdout, tail
