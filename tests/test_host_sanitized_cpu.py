"""The host shim under sanitizers, with every HIP call site failed once (CPU build only: GPU sanitizers are not available).

csrc/fx_capi.cpp and csrc/fx_comm.cpp -- ring slots, captured steps, the fill pool's threads, the re-blocking plumbing of fx_push_samples,
the RCCL gather -- are compiled UNCHANGED for the host against tests/cpp/fake_hip/ (a malloc-backed hip_runtime.h whose every call can be
made to fail, launch stubs for csrc/fx_kernels.h, a librccl.so.1 whose ranks are processes that meet in POSIX shared memory) with -fsanitize=address,undefined, and tests/cpp/host_sanitize.cpp
walks six scenarios once per HIP call with that call failing: no crash, no overrun, no leak, no wedged ring, the next call works; a seventh checks
fx_push_samples' arithmetic on the host (random block lengths x formats x windows: the hops handed to the kernels, put end to end, are the stream).  A second
build with -fsanitize=thread runs the fill pool (1 .. 64 threads, resized up and down, jobs back to back).
What this found when it was written (round 5): grow() freed a scratch buffer twice when hipFree itself reported a failure.

Round 6: the world > 1 branches of csrc/fx_comm.cpp -- the exchange of channel counts, the per-source receive offsets, ranks that are not the
sink, the double-buffered staging -- run here for the first time anywhere: tests/cpp/comm_ranks.cpp forks 2 / 4 / 8 rank processes (fake HIP,
fake RCCL, ASan + UBSan) with ragged shards, sinks other than rank 0, host and device destinations, every gather of a run in flight together;
then every RCCL call and every HIP call of a sink and of a non-sink rank is failed once: no hang, no crash, no leak, the context analyses
again.  What this cannot show is xGMI: the transport is memcpy."""
import glob
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
           os.path.join(CSRC, "fx_capi.cpp"), os.path.join(CSRC, "fx_comm.cpp"), os.path.join(CSRC, "fx_osc_sender.cpp"), os.path.join(FAKE, "fake_hip.cpp"),
           os.path.join(ROOT, "tests", "cpp", "host_sanitize.cpp"), "-o", out, "-ldl", "-lpthread"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    return out


@pytest.fixture(scope="module")
def fake_rccl(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("fake_rccl"))
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-shared", "-fPIC", "-I", FAKE, os.path.join(FAKE, "fake_rccl.cpp"), "-o", os.path.join(d, "librccl.so.1"), "-lrt", "-lpthread"])
    return d


def _build_comm_ranks(tmp, comm_source=None):
    out = os.path.join(tmp, "comm_ranks_asan")
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-rdynamic",
           "-I", FAKE, "-I", os.path.join(ROOT, "include"), "-I", CSRC,
           os.path.join(CSRC, "fx_capi.cpp"), comm_source or os.path.join(CSRC, "fx_comm.cpp"), os.path.join(FAKE, "fake_hip.cpp"),
           os.path.join(ROOT, "tests", "cpp", "comm_ranks.cpp"), "-o", out, "-ldl", "-lpthread"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    return out


@pytest.fixture(scope="module")
def comm_ranks(tmp_path_factory):
    return _build_comm_ranks(str(tmp_path_factory.mktemp("comm_ranks")))


def _ranks(exe, lib_dir, *args, timeout=300, patience_ms=30000):
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = lib_dir
    env.pop("ROCM_PATH", None); env.pop("ROCM_HOME", None)
    env["FAKE_HIP_DEVICES"] = "8"                  # one "GPU" per rank process
    env["FAKE_RCCL_TIMEOUT_MS"] = str(patience_ms)
    env["ASAN_OPTIONS"] = "detect_leaks=1:abort_on_error=0"
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    p = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, env=env, timeout=timeout)
    return p, (p.stdout + p.stderr)[-4000:]


def _no_segment_left():
    left = glob.glob("/dev/shm/fxfakerccl-*")
    for f in left:
        os.unlink(f)
    assert not left, "communicator segments left in /dev/shm: %s" % left


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
    assert any(line.startswith("osc sender / receiver") and "clean run" in line for line in p.stdout.splitlines()), p.stdout
    assert any(line.startswith("live engine") and "clean run" in line for line in p.stdout.splitlines()), p.stdout


def test_fill_pool_under_tsan(tmp_path, fake_rccl):
    exe = _build(str(tmp_path), "thread", "host_tsan")
    p = _run(exe, "tsan", fake_rccl)
    tail = (p.stdout + p.stderr)[-4000:]
    if "FATAL: ThreadSanitizer: unexpected memory mapping" in p.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow in this container")
    assert p.returncode == 0, tail
    assert "WARNING: ThreadSanitizer" not in p.stderr, tail
    assert "host_sanitize: 0 problem(s)" in p.stdout, tail
    assert any(line.startswith("osc sender / receiver") and "clean run" in line for line in p.stdout.splitlines()), p.stdout
    assert any(line.startswith("live engine") and "clean run" in line for line in p.stdout.splitlines()), p.stdout


@pytest.mark.parametrize("world, shards, sinks, dst", [
    (2, "8192,37", "0,0,0,1,0", "mixed"),
    (4, "8192,8191,1,37", "0,0,0,3,1,0", "mixed"),
    (4, "8192,8191,1,37", "0,0,0,3,1,0", "host"),           # (host-only / device-only destinations at one world size)
    (4, "8192,8191,1,37", "0,0,0,3,1,0", "device"),
    (8, "100,1,37,64,5,3,2,999", "0,0,7,0,3,3,0", "mixed"),
])
def test_comm_ranks_fake_rccl(comm_ranks, fake_rccl, world, shards, sinks, dst):
    """fx_comm.cpp with MORE THAN ONE rank: ragged shards, the sink moving between ranks, tables in host and in device memory, no
    fx_comm_sync until every round's gather has been issued.  The parent process compares each rank's block of every gathered table
    with what that rank held itself."""
    p, tail = _ranks(comm_ranks, fake_rccl, world, "shards=" + shards, "sinks=" + sinks, "dst=" + dst)
    assert p.returncode == 0, tail
    assert "comm_ranks ok: %d ranks, %d channels, %d rounds" % (world, sum(map(int, shards.split(","))), len(sinks.split(","))) in p.stdout, tail
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, tail
    assert p.stderr.count("RCCL communicator of %d rank(s)" % world) == world, tail
    _no_segment_left()


def test_comm_ranks_check_is_not_vacuous(tmp_path, fake_rccl):
    """The same program against an fx_comm.cpp whose receive offsets are wrong must fail: the gather test can see a misplaced block."""
    src = open(os.path.join(CSRC, "fx_comm.cpp")).read()
    good = "d_dst + (size_t) m->first[(size_t) src] * FX_NUM_FEATURES"
    assert src.count(good) == 1
    mutant = str(tmp_path / "fx_comm_mutant.cpp")
    open(mutant, "w").write(src.replace(good, "d_dst + (size_t) m->first[(size_t) (src ? src - 1 : 0)] * FX_NUM_FEATURES"))
    exe = _build_comm_ranks(str(tmp_path), mutant)
    p, tail = _ranks(exe, fake_rccl, 3)
    assert p.returncode != 0 and "differs from that rank's own features" in p.stderr, tail
    _no_segment_left()


@pytest.mark.parametrize("kind", ["fail", "hipfail"])
def test_comm_ranks_every_call_site_of_a_rank_failed_once(comm_ranks, fake_rccl, kind):
    """Three rank processes, the sink moving 0 -> 2 -> 0; the k-th RCCL call (fail) or HIP call (hipfail) of rank 0, then of rank 2, fails, for
    every k until the run ends before call k is reached.  Whatever the failing call was -- the communicator's creation, the count exchange, a
    send, a receive, the group's end, an event or a copy around them -- every rank returns (an error where it was told one, else its
    results), gives everything back, analyses again on the same context, and the shared segment is gone."""
    reported = 0
    for rank in (0, 2):
        quiet = 0
        for k in range(1, 400):
            p, tail = _ranks(comm_ranks, fake_rccl, 3, "sinks=0,2,0", "%s=%d:%d" % (kind, rank, k), timeout=120, patience_ms=3000)   # (a rank that dies before it joins is waited for, as RCCL's bootstrap would: 3 s here)
            assert p.returncode == 0, "%s=%d:%d\n%s" % (kind, rank, k, tail)
            assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, tail
            line = [l for l in p.stdout.splitlines() if l.startswith("comm_ranks injected")]
            assert line, tail
            n = int(line[0].split("failing:")[1].split()[0])
            reported += n > 0
            quiet = quiet + 1 if n == 0 else 0
            if quiet >= 3:            # (a failed free or timing event may pass silently; three calls in a row that nobody noticed = past the end)
                break
        else:
            raise AssertionError("the walk never reached the end of the run")
    assert reported >= (20 if kind == "fail" else 40), reported
    _no_segment_left()
