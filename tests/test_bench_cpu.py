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
    line = open(os.path.join(ROOT, "profiles", "r01q_bench.json")).read().strip().splitlines()[-1]
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["dtype"] == "f32" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and "sample" in c
