#!/usr/bin/env python3
# (the env shebang is harmless HERE only: this process never touches a GPU -- everything that does is a child it starts -- so the exec it
# implies happens before any GPU state exists.  Under rocprofv3, or from a process that has initialised HIP, start it as `python3 tools/...`.)
"""First contact with a multi-GPU node: everything about the N > 1 path in ONE run and ONE JSON, so that a scaling result below
target can be diagnosed without a second run.  (No >= 2-GPU node was available while this was built: see DESIGN.md section 6.)

    tools/first_multigpu_run.sh [--out gpurun_out/first_multigpu.json] [--steps 20] [--warmup 5] [--max-gpus 8]

What it runs, each as a child process with a time-out (this process never touches a GPU):
  1. tests/cpp/comm_ranks <n> for n = 2 and the largest of {4, 8} the node has: the RCCL gather through the C ABI from a torch-free C++
     program (fork + pipes, ragged shards), rank 0 checking the gathered table against every rank's own;
  2. bench.py --gpus n for n = 1, 2, 4, 8 (those the node has): configs[3], 8192 channels per GPU, the gather to rank 0 every step.
What the JSON holds: per n the bench line (whole-job frames/s, max over ranks), `per_rank` (every rank's own frames/s, frame-kernel and
tail-kernel time per step, RCCL's rank count, gathers issued / timed, mean and longest gather time on the side stream), the scaling
factors against n = 1, and plain-language findings (a slow rank, a gather that is not hidden, a communicator of the wrong size).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def findings(n, line):
    """Plain-language reading of one bench line with per-rank records."""
    out = []
    ranks = line.get("per_rank") or []
    if len(ranks) != n:
        out.append("n=%d: %d per-rank records (expected %d)" % (n, len(ranks), n))
        return out
    rates = [r["frames_per_s"] for r in ranks]
    if min(rates) < 0.93 * max(rates):
        slow = min(ranks, key=lambda r: r["frames_per_s"])
        out.append("n=%d: rank %d (device %s) ran %.1f %% slower than the fastest rank; its frame kernel took %.3f ms per step against a median of %.3f"
                   % (n, slow["rank"], slow.get("device"), 100.0 * (1.0 - min(rates) / max(rates)), slow["frame_kernel_ms_per_step"],
                      sorted(r["frame_kernel_ms_per_step"] for r in ranks)[len(ranks) // 2]))
    step_ms = line["ms_per_step"]
    for r in ranks:
        ex = r.get("exchange")
        if not ex:
            continue
        if ex["rccl_ranks"] != n:
            out.append("n=%d: rank %d: RCCL counts %d ranks in its communicator" % (n, r["rank"], ex["rccl_ranks"]))
        if ex.get("gather_ms_mean") is not None and ex["gather_ms_mean"] > 0.25 * step_ms:
            out.append("n=%d: rank %d: a gather takes %.3f ms on the side stream, %.0f %% of a %.3f ms step -- it may no longer hide behind the next step's kernels"
                       % (n, r["rank"], ex["gather_ms_mean"], 100.0 * ex["gather_ms_mean"] / step_ms, step_ms))
    kernels = max(r["frame_kernel_ms_per_step"] + r["tail_kernels_ms_per_step"] for r in ranks)
    if step_ms > 1.15 * kernels:
        out.append("n=%d: a step takes %.3f ms but the slowest rank's kernels only %.3f ms: %.0f %% of the step is not kernel time (launch gaps, the "
                   "host loop, or a gather the step waits for)" % (n, step_ms, kernels, 100.0 * (1.0 - kernels / step_ms)))
    return out


def assemble(visible, comm, lines, started=None):
    """The report: comm = {n: {...}} results of comm_ranks, lines = {n: parsed bench line or {"error": ...}}."""
    rep = {"what": "first multi-GPU run of the channel-sharded path (BASELINE configs[3]); tools/first_multigpu_run.py",
           "visible_gpus": visible, "comm_ranks": {str(k): v for k, v in comm.items()}, "bench": {str(k): v for k, v in lines.items()},
           "scaling": {}, "findings": []}
    if started is not None:
        rep["wall_s"] = time.time() - started
    base = lines.get(1, {}).get("value")
    for n, line in sorted(lines.items()):
        if "value" not in line:
            rep["findings"].append("n=%d: no bench line (%s)" % (n, line.get("error", "?")))
            continue
        if base:
            rep["scaling"][str(n)] = {"frames_per_s": line["value"], "speedup_vs_1": line["value"] / base, "efficiency": line["value"] / base / n}
        if n > 1:
            rep["findings"] += findings(n, line)
    for n, c in sorted(comm.items()):
        if c.get("rc") not in (0, 77):
            rep["findings"].append("comm_ranks %d failed (rc %s): %s" % (n, c.get("rc"), (c.get("tail") or "")[-200:]))
    if not rep["findings"]:
        rep["findings"].append("nothing unusual")
    return rep


def run(cmd, timeout, env=None):
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
        return p.returncode, p.stdout, p.stderr
    except subprocess.TimeoutExpired as e:
        return 124, (e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or ""), "timed out after %d s" % timeout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "first_multigpu.json"))
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--max-gpus", type=int, default=8)
    ap.add_argument("--timeout", type=int, default=600, help="seconds per child")
    ap.add_argument("--rehearse-ranks", type=int, default=0,
                    help="REHEARSAL on a box with fewer GPUs than ranks: run bench.py --gpus 1, 2, 4 ... up to this many rank processes with --backend gloo "
                         "(the ranks share the GPUs there are and gather through host memory): the real shard plan, control group, per-step gather, sink check "
                         "and per-rank records -- everything but RCCL between GPUs.  Its scaling factors mean nothing (one GPU does all the work) and the report says so.")
    ap.add_argument("--channels-per-gpu", type=int, default=None)
    ap.add_argument("--frames", type=int, default=None)
    args = ap.parse_args()
    import bench
    started = time.time()
    env = bench.rank_environment(dict(os.environ))
    visible = bench.visible_gpus() or 0
    gpus = min(visible, args.max_gpus)
    # build the library and the C++ rank program here, before anything runs
    fxbuild = __import__("importlib").import_module("feature-extractor_amd.build")
    lib = fxbuild.build()
    lib_dir = os.path.dirname(lib)
    exe = os.path.join(ROOT, "tools", "_bin", "comm_ranks")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    # (the HIP runtime for the sink's device-resident destination tables: hipMalloc / hipMemcpy / hipFree)
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(rocm, "include"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "comm_ranks.cpp"), "-o", exe, "-L", lib_dir, "-lfx_hip", "-Wl,-rpath," + lib_dir,
                           "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-ldl"])
    comm = {}
    rehearsal = args.rehearse_ranks > 0
    for n in ([] if rehearsal else sorted({2, max([k for k in (2, 4, 8) if k <= gpus] or [2])})):
        # ragged shards, the sink moving off rank 0 once, host and device destinations, every round's gather in flight together -- the
        # shape tests/test_host_sanitized_cpu.py runs against the multi-process fake RCCL
        shards = ",".join(str(v) for v in ([8192, 8191, 1, 37, 4096, 5, 640, 2][:n]))
        sinks = "0,0,0,%d,0" % (n - 1)
        rc, so, se = run([exe, str(n), "shards=" + shards, "sinks=" + sinks], args.timeout, env)
        comm[n] = {"rc": rc, "skipped": rc == 77, "tail": (so + se)[-1500:]}
    lines = {}
    extra = []
    if args.channels_per_gpu is not None:
        extra += ["--channels-per-gpu", str(args.channels_per_gpu)]
    if args.frames is not None:
        extra += ["--frames", str(args.frames)]
    for n in (1, 2, 4, 8):
        if n > (args.rehearse_ranks if rehearsal else max(gpus, 1)):
            continue
        rc, so, se = run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", str(args.steps), "--warmup", str(args.warmup),
                          "--no-extra", "--no-cpu-baseline", "--no-pmc", "--rank-timeout", str(args.timeout)] + (["--backend", "gloo"] if rehearsal and n > 1 else []) + extra,
                         args.timeout + 60, env)
        line = None
        for l in so.splitlines():
            if l.startswith("{"):
                try:
                    line = json.loads(l)
                except ValueError:
                    pass
        lines[n] = line if line is not None else {"error": "rc %d: %s" % (rc, (se or so)[-800:])}
        if line is not None:
            lines[n]["stderr_tail"] = se[-1200:]
    rep = assemble(visible, comm, lines, started)
    if rehearsal:
        rep["rehearsal"] = ("REHEARSAL on %d visible GPU(s): the rank processes of n > 1 share them (--backend gloo, gather through host memory).  Shard plan, control "
                            "group, per-step gather, sink check and per-rank records are the real ones; `scaling` is NOT a scaling measurement and RCCL / xGMI did not run."
                            % visible)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(rep, open(args.out, "w"), indent=1)
    print(json.dumps({"out": args.out, "visible_gpus": visible, "scaling": rep["scaling"], "findings": rep["findings"]}, indent=1))


if __name__ == "__main__":
    main()
