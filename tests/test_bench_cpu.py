"""bench.py's host-side helpers (no GPU): the committed PMC traffic record matches the bench default shape, the core
count honours the cgroup quota, and the argument defaults are the workload DESIGN.md describes."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_traffic_record_matches_the_default_shape():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True)
    assert out.returncode == 0 and "--frames" in out.stdout and "--channels-per-gpu" in out.stdout
    rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert (rec["window"], rec["channels"], rec["frames"]) == (1024, 1024, 512)
    traffic = bench.load_traffic(1024, 1024, 512)
    algorithmic = (4 * 1024 + 48) * 1024 * 512
    assert traffic is not None and algorithmic <= traffic <= 1.1 * algorithmic      # no wasted re-reads
    assert bench.load_traffic(2048, 1024, 512) is None                              # other shapes: unknown, not guessed


def test_usable_cores_is_sane():
    n = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_committed_bench_line_has_the_contract_fields():
    line = open(os.path.join(ROOT, "profiles", "r02_bench.json")).read().strip().splitlines()[-1]
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["dtype"] == "f32" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and "sample" in c
    # the extras the round-2 review asked for: unfriendly inputs and the other window sizes
    assert set(d["data_dependence"]) >= {"noise", "silence"} and set(d["other_windows"]) == {"2048", "4096"}


def test_valu_model_record():
    """profiles/valu_model.json (tools/valu_model.py) is what bench.py's roofline.valu_issue_frac is computed from"""
    m = bench.valu_model(1024)
    assert m and 2000 < m["valu_per_frame"] < 3500 and 1.1 <= m["mean_issue_ns"] <= 1.85
    assert bench.valu_model(2048) and bench.valu_model(4096) and bench.valu_model(512) is None


def test_gpus_n_without_a_launcher_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus 4` from a bare shell must start 4 child ranks through torch.distributed.run (and never
    exec); with WORLD_SIZE set (the driver's torchrun form) it must not."""
    import subprocess as sp
    seen = {}

    class Done:
        returncode = 0

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    import torch
    monkeypatch.setattr(sp, "run", fake_run)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
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
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    try:
        bench.main()
        raise AssertionError("expected SystemExit")
    except SystemExit as e:
        assert "only 1 GPU" in str(e.code)
