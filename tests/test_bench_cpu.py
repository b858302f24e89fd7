"""bench.py's host-side logic (no GPU): counter records are never reported stale, counter arithmetic follows the MI355X
guide, the rank environment is the same for both launch forms, a wedged phase ends the process, and `--gpus N` from a
bare shell starts its own ranks."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_help_and_defaults():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True)
    assert out.returncode == 0 and "--frames" in out.stdout and "--channels-per-gpu" in out.stdout
    a = bench.build_parser().parse_args([])
    assert (a.gpus, a.window, a.channels_per_gpu, a.frames) == (1, 1024, None, None)       # rank_main: 1024 x 512 at one GPU


def test_committed_counters_are_never_reported_stale(tmp_path):
    """A committed counter record is a fallback for runs that cannot read counters themselves, and only if it was taken
    from exactly the kernel sources that are running: after any change under csrc/ it reads as absent, not as a number."""
    sha = bench.kernel_sources_sha()
    assert len(sha) == 16 and sha == bench.kernel_sources_sha()
    path = str(tmp_path / "counters.json")
    pmc = {"FETCH_SIZE": 1058689.4, "WRITE_SIZE": 115145.3, "SQ_INSTS_VALU": 2734.6 * 524288}
    json.dump({"1024:1024:512": {"pmc": pmc, "kernel_sources_sha": sha}}, open(path, "w"))
    rec, why = bench.committed_counters(1024, 1024, 512, path=path)
    assert rec and rec["pmc"]["FETCH_SIZE"] == pmc["FETCH_SIZE"] and "same kernel sources" in why
    rec, why = bench.committed_counters(1024, 1024, 512, path=path, sources_sha="0" * 16)
    assert rec is None and "stale" in why
    rec, why = bench.committed_counters(2048, 1024, 512, path=path)
    assert rec is None                                                               # other shapes: unknown, not guessed
    rec, why = bench.committed_counters(1024, 1024, 512, path=str(tmp_path / "missing.json"))
    assert rec is None
    # whatever is committed in profiles/ obeys the same rule
    rec, why = bench.committed_counters(1024, 1024, 512)
    assert rec is None or rec["kernel_sources_sha"] == sha


def test_counter_arithmetic_follows_the_guide():
    """HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB; gfx950 tallies 128-B read requests at 64 B), per launch; instruction
    counts per frame; per-wave activity as a share of wave cycles."""
    n = 1024 * 512
    f = bench.counters_to_fields({"FETCH_SIZE": 1058689.4, "WRITE_SIZE": 115145.3, "SQ_INSTS_VALU": 2734.6 * n, "SQ_INSTS_LDS": 400.0 * n,
                                  "SQ_WAVE_CYCLES": 1000.0, "SQ_ACTIVE_INST_VALU": 230.0, "SQ_WAIT_INST_LDS": 128.0}, n)
    algorithmic = (4 * 1024 + 48) * n
    assert algorithmic <= f["traffic"] <= 1.1 * algorithmic
    assert abs(f["valu_insts_per_frame"] - 2734.6) < 1e-6 and abs(f["lds_insts_per_frame"] - 400.0) < 1e-9
    assert abs(f["valu_active_per_wave"] - 0.23) < 1e-12 and abs(f["lds_issue_stall_per_wave"] - 0.128) < 1e-12
    assert bench.counters_to_fields({}, n) == {}
    assert abs(bench.flops_per_frame(1024) - 235520.0) < 1e-6 and 5.0e5 < bench.flops_per_frame(2048) < 5.2e5


def test_usable_cores_is_sane():
    n = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_rank_environment_is_the_same_for_both_launch_forms(monkeypatch):
    """What self_launch hands its children and what a rank started by the driver's own torch.distributed.run command sets
    for itself is one function: HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC, which RCCL's peer buffers need here)."""
    env = bench.rank_environment({})
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "OMP_NUM_THREADS" in env
    assert bench.rank_environment({"HSA_ENABLE_IPC_MODE_LEGACY": "1"})["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"   # an explicit choice wins
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    bench.rank_environment()
    assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_watchdog_ends_a_wedged_phase():
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "with bench.Watchdog(1, 'a phase that never ends'):\n"
            "    time.sleep(30)\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=25)
    assert p.returncode == 124 and "did not finish within 1 s" in p.stderr
    with bench.Watchdog(30, "a phase that ends"):
        pass                                                                       # cancelled: nothing fires later


def test_committed_bench_line_has_the_contract_fields():
    paths = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if p.endswith("_bench.json") and p.startswith("r03"))
    if not paths:
        pytest.skip("no round-3 bench line committed yet")
    d = json.loads(open(os.path.join(ROOT, "profiles", paths[-1])).read().strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["dtype"] == "f32" and "workload" in d["config"]
    r = d["roofline"]
    assert r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = r["compute"]
    assert r["bound"] == "valu" and abs(c["frac"] - c["achieved_tflops"] / c["peak_no_fma"]) < 1e-12 and c["peak_tflops"] == 157.3


def test_gpus_n_without_a_launcher_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus 4` from a bare shell must start 4 child ranks through torch.distributed.run (and never
    exec) without touching the GPU itself; with WORLD_SIZE set (the driver's torchrun form) it must not."""
    import subprocess as sp
    seen = {}

    class Done:
        returncode = 0

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    monkeypatch.setattr(sp, "run", fake_run)
    monkeypatch.setattr(bench, "visible_gpus", lambda: 4)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert "127.0.0.1" in cmd and cmd[-4:] == ["--gpus", "4", "--steps", "7"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # too few GPUs for RCCL: refuse loudly instead of hanging in the communicator
    monkeypatch.setattr(bench, "visible_gpus", lambda: 1)
    try:
        bench.main()
        raise AssertionError("expected SystemExit")
    except SystemExit as e:
        assert "only 1 GPU" in str(e.code)


def test_no_nested_profiler_children():
    """A bench.py that runs under rocprofv3 must not start rocprofv3 children of its own (they would inherit the tool library,
    initialise the GPU in the launcher and then exec)."""
    import bench
    assert not bench.under_profiler({})
    assert bench.under_profiler({"LD_PRELOAD": "/opt/rocm/lib/librocprofiler-sdk-tool.so"})
    assert bench.under_profiler({"ROCPROF_OUTPUT_PATH": "/tmp/x"})
    assert bench.under_profiler({"ROCP_TOOL_LIBRARIES": "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so"})
