#!/usr/bin/env python3
"""bench.py -- frames/sec of the full 12-feature bundle on synthetic audio, per BASELINE.json.

Workload at N=1 GPU: BASELINE.json configs[1] -- 1024 synthetic 48 kHz channels x 1024-pt fp32
frames (50 % overlap, pre-assembled [C][T][1024], resident in HBM), the whole RealTimeAnalyser
bundle (spectral + pitch + harmonic + RMS + smoothing + onset).  A "step" is one pass of the hot
path over one batch of T consecutive frames per channel.  With N > 1 GPUs the workload is BASELINE.json
configs[3]: every rank analyses its own contiguous 8192-channel shard of 8192*N channels (weak scaling;
channels are independent, so there is no data-path collective) and the smoothed feature vectors are
gathered to rank 0, the OSC sink, over RCCL (through the C ABI: fx_comm_* / fx_gather_smoothed).

Launch forms (both work, and both run the ranks in the same environment -- rank_environment()):
    python bench.py --gpus N ...                                   (bench.py starts its N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N ...                    (the driver's form)

Prints ONE JSON line on rank 0.
"""
import argparse
import csv
import glob
import importlib
import json
import math
import os
import shutil
import signal
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X fp32 vector peak (MI355X_MICROARCH.md), counted with FMA = 2 flop
FP32_NO_FMA_PEAK_TFLOPS = FP32_VECTOR_PEAK_TFLOPS / 2.0   # the bit-exact FFT DAG may not fuse a*b+c: one flop per lane per issue


def flops_per_frame(window):
    """the as-written bundle is four complex N-point FFTs (5 N log2 N each) + ~30 N of reductions."""
    return 4 * 5.0 * window * math.log2(window) + 30.0 * window


def rank_environment(env=None):
    """What a rank process needs in its environment BEFORE the HIP runtime starts, whichever way it was launched
    (self_launch passes it to its children; main() applies it to os.environ first thing, so the driver's own
    `python -m torch.distributed.run ... bench.py` form runs the ranks under the same settings)."""
    env = os.environ if env is None else env
    # the host driver supports dmabuf IPC only: without this RCCL's peer buffers fail with hipIpcGetMemHandle: invalid argument
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return env


class Watchdog:
    """A phase that must finish in `seconds` or the process exits non-zero with a message (a wedged rank -- one peer never
    reaching ncclCommInitRank or a barrier -- otherwise hangs the whole multi-GPU run).  The process EXITS; nothing is
    re-executed."""

    def __init__(self, seconds, what):
        self.what, self.seconds = what, seconds
        self.timer = threading.Timer(seconds, self._fire)
        self.timer.daemon = True

    def _fire(self):
        print("bench.py: rank %s: '%s' did not finish within %d s -- giving up (exit 124)"
              % (os.environ.get("RANK", "0"), self.what, self.seconds), file=sys.stderr, flush=True)
        os._exit(124)

    def __enter__(self):
        self.timer.start()
        return self

    def __exit__(self, *exc):
        self.timer.cancel()
        return False


def usable_cores():
    """Host threads this process can really run at once: the affinity mask, capped by the cgroup CPU quota
    (the GPU boxes show 256 hardware threads under a 16-CPU quota; 256 runnable threads there only get
    throttled)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                       # cgroup v2: "<quota|max> <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = float(f.read()), float(g.read())
                if q > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def _flush_c_stdio():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


def cpu_baseline(fx, window, frames_per_channel, seconds=30.0):
    """The CPU oracle (a port of the reference's algorithm, oracle/fx_oracle.c) on this host's
    cores, on a bounded sample of the same synthetic workload: one pthread per core over disjoint
    channel blocks, mirroring the reference's thread-pair-per-channel model."""
    from oracle import fx_oracle as fo
    fo.lib()
    cores = usable_cores()
    T = min(frames_per_channel, 16)
    probe = fx.synth.frames(4, T, window)
    fo.batch_frames(probe, window, threads=1)            # first call: page faults, FFT tables
    t0 = time.perf_counter()
    fo.batch_frames(probe, window, threads=1)
    per_frame = (time.perf_counter() - t0) / (4 * T)
    single = 1.0 / per_frame
    # about `seconds` of CPU work in total: a block of 64 channels per thread, analysed repeatedly
    chans = 64 * cores
    base = fx.synth.frames(64, T, window)
    data = np.ascontiguousarray(np.tile(base, (cores, 1, 1)))
    reps = int(max(1, round(seconds / (per_frame * T * 64 * cores))))
    t0 = time.perf_counter()
    for _ in range(reps):
        fo.batch_frames(data, window, threads=cores)
    dt = time.perf_counter() - t0
    chans *= reps
    return {"value": chans * T / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "single_thread_value": single,
            "sample": "%d channels x %d frames of the %d-pt synthetic workload on %d pthreads (%d hardware threads visible, "
                      "cgroup CPU quota applied), %.1f s wall = %.0f CPU-s (oracle/fx_oracle.c, gcc -O2)"
                      % (chans, T, window, cores, os.cpu_count() or 1, dt, dt * cores)}


# ---------------------------------------------------------------------------------------------------------------
# counters: measured IN THIS RUN by rocprofv3 children of this process; committed records only as a labelled fallback
# ---------------------------------------------------------------------------------------------------------------
def kernel_sources_sha():
    """sha256 over the kernel sources (csrc/*.h, *.hip, *.cpp): what a committed counter record must have been taken
    from to describe the library that is running."""
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "feature-extractor_amd", "csrc", "*"))):
        if path.endswith((".h", ".hip", ".cpp")):
            h.update(os.path.basename(path).encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def committed_counters(window, channels, frames, path=None, sources_sha=None):
    """profiles/counters.json (written by `bench.py --write-counters` on a GPU box): counters of earlier runs, keyed by
    shape.  Used only when the counters cannot be read in this run, and only if the record was taken from exactly the
    kernel sources that are running -- a stale record is reported as absent, never as a number."""
    path = path or os.path.join(ROOT, "profiles", "counters.json")
    sha = sources_sha or kernel_sources_sha()
    try:
        rec = json.load(open(path)).get("%d:%d:%d" % (window, channels, frames))
    except Exception:
        return None, "no committed record"
    if not rec:
        return None, "no committed record for this shape"
    if rec.get("kernel_sources_sha") != sha:
        return None, "committed record is stale (taken from kernel sources %s, running %s)" % (rec.get("kernel_sources_sha"), sha)
    return rec, "committed record (profiles/counters.json, same kernel sources)"


PMC_PASSES = [
    # one rocprofv3 run each (MI355X guide: SQ has 8 slots; FETCH_SIZE and WRITE_SIZE do not fit one TCC pass)
    ("sq", ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT"]),
    ("fetch", ["FETCH_SIZE"]),
    ("write", ["WRITE_SIZE"]),
]


def _parse_counter_csv(out_dir, kernel_substr):
    sums, counts = {}, {}
    for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kernel_substr in row.get("Kernel_Name", ""):
                c = row["Counter_Name"]
                sums[c] = sums.get(c, 0.0) + float(row["Counter_Value"])
                counts[c] = counts.get(c, 0) + 1
    return {c: sums[c] / counts[c] for c in sums}, (max(counts.values()) if counts else 0)


def under_profiler(environ=None):
    """True when this process was started by rocprofv3 / rocprof (their tool libraries ride in on these variables)."""
    e = os.environ if environ is None else environ
    if any("rocprof" in e.get(k, "").lower() for k in ("LD_PRELOAD", "HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY")):
        return True
    return any(k.startswith(("ROCPROF_", "ROCPROFILER_")) for k in e)


def measure_counters(window, channels, frames, input_file, timeout_s=150, passes=None, analysers="both"):
    """Per-launch counters of the frame kernel at this shape, read now: each pass is a child `rocprofv3 --pmc ... --
    python3 bench.py --pmc-child ...` (a fresh process; this one is never re-executed).  Returns (dict or None, note)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    if under_profiler():
        # a child launcher would inherit the tool library, initialise the GPU with it and then exec its target: the one thing
        # a GPU-initialised process must not do on this pool.  The outer profiler is collecting what it was asked for.
        return None, "not read: this run is itself under a profiler"
    # never compile under the profiler: the child only LOADS the library (a stale one makes it exit non-zero), so build here, in
    # the parent, where no tool library is preloaded and the GPU has not been touched by a launcher chain
    fxbuild = importlib.import_module("feature-extractor_amd.build")
    if fxbuild.needs_build():
        fxbuild.build()
    tmp = tempfile.mkdtemp(prefix="fx_pmc_", dir="/tmp")
    env = dict(os.environ)
    env["TMPDIR"] = "/tmp"
    got, launches = {}, 0
    try:
        for name, counters in (passes or PMC_PASSES):
            out = os.path.join(tmp, name)
            cmd = [exe, "--pmc"] + counters + ["--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
                                               "--pmc-child", "--window", str(window), "--channels-per-gpu", str(channels),
                                               "--frames", str(frames), "--input-file", input_file, "--steps", "3", "--warmup", "1", "--analysers", analysers]
            # its own session: on a time-out the whole group goes (launcher AND the wrapped bench.py, which would otherwise keep
            # the GPU busy under the measurements that follow)
            p = subprocess.Popen(cmd, env=env, cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
            try:
                _, err = p.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except OSError:
                    pass
                p.communicate()
                return None, "rocprofv3 pass '%s' exceeded %d s" % (name, timeout_s)
            if p.returncode != 0:
                return None, "rocprofv3 pass '%s' failed (rc %d): %s" % (name, p.returncode, (err or "").strip().splitlines()[-1:] or "")
            vals, n = {}, 0
            for kern in ("fx_frame_kernel<%d" % window, "fx_pair_kernel<%d" % window):       # (FX_WAVES_PER_FRAME=2 in the environment runs pairs)
                vals, n = _parse_counter_csv(out, kern)
                if vals:
                    break
            if not vals:
                return None, "rocprofv3 pass '%s' recorded no frame-kernel dispatch" % name
            got.update(vals)
            launches = max(launches, n)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    got["launches"] = launches
    return got, "rocprofv3 --pmc children of this run (%d passes, %d launches each)" % (len(passes or PMC_PASSES), launches)


def counters_to_fields(pmc, n_frames):
    """The bench line's counter-derived fields from per-launch counter means (MI355X guide, HBM section: read bytes =
    2 x FETCH_SIZE on gfx950 for wide coalesced streaming reads, WRITE_SIZE exact; both in KB)."""
    out = {}
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        out["traffic"] = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
    if "SQ_INSTS_VALU" in pmc:
        out["valu_insts_per_frame"] = pmc["SQ_INSTS_VALU"] / n_frames
    if "SQ_INSTS_LDS" in pmc:
        out["lds_insts_per_frame"] = pmc["SQ_INSTS_LDS"] / n_frames
    if pmc.get("SQ_WAVE_CYCLES"):
        # share of a resident wavefront's lifetime in which it is issuing a VALU instruction / stalled on the LDS pipe
        out["valu_active_per_wave"] = pmc.get("SQ_ACTIVE_INST_VALU", 0.0) / pmc["SQ_WAVE_CYCLES"]
        out["lds_issue_stall_per_wave"] = pmc.get("SQ_WAIT_INST_LDS", 0.0) / pmc["SQ_WAVE_CYCLES"]
    return out


def pmc_child(args):
    """The process rocprofv3 wraps: the same launches as the timed region, on the same input bytes, nothing else."""
    import torch
    fx = importlib.import_module("feature-extractor_amd")
    if importlib.import_module("feature-extractor_amd.build").needs_build():
        raise SystemExit("bench.py --pmc-child: libfx_hip.so is older than its sources; build it first (never under the profiler)")
    fx.load_library(build_if_missing=False)
    torch.cuda.set_device(0)
    frames = torch.from_numpy(np.load(args.input_file)).cuda(0)
    C, T = frames.shape[0], frames.shape[1]
    an = fx.BatchAnalyser(C, args.window, device=0, analysers=args.analysers)
    raw = torch.empty((C, T, 12), dtype=torch.float32, device=frames.device)
    sm = torch.empty_like(raw)
    hops = frames.shape[2] == args.window // 2          # (tools/pmc_quick.py --hops: the overlapper's input, half a window per frame)
    for _ in range(args.warmup + args.steps):
        if hops:
            an.push_hops(frames, out_raw=raw, out_smoothed=sm)
        else:
            an.process_frames(frames, out_raw=raw, out_smoothed=sm)
    an.sync()
    an.close()


def stream_bench(fx, args, C, T, N, device):
    """End-to-end streaming: hops live in host memory, go through the pinned ring with H2D copies on a
    side stream overlapped with analysis, results come back to the host.  PCIe-inclusive."""
    dtype = np.float16 if args.fp16 else np.float32
    hops = fx.synth.hops(C, T, N).astype(dtype)
    an = fx.BatchAnalyser(C, N, device=device)
    st = fx.HopStream(an, T, slots=3, dtype=dtype)
    fill_threads = max(1, min(8, usable_cores() // 2))

    def step():
        if st.in_flight() == 3:
            st.collect(want_raw=False)
        st.push(hops, fill_threads=fill_threads)    # the producer's copy into pinned memory (fx_stream_push's thread pool) is part of the path

    for _ in range(args.warmup):
        step()
    while st.in_flight():
        st.collect(want_raw=False)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    while st.in_flight():
        st.collect(want_raw=False)
    dt = time.perf_counter() - t0
    frames = C * T * args.steps
    in_bytes = hops.nbytes * args.steps
    print(json.dumps({"metric": "frames/sec, streaming ingest (host hops -> pinned ring -> GPU -> host features)",
                      "value": frames / dt, "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                      "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "dtype": "f16" if args.fp16 else "f32",
                      "data": "synthetic", "pcie_inclusive": True, "host_to_device_GBps": in_bytes / dt / 1e9,
                      "config": {"workload": "%d channels x %d hops/batch x %d-pt windows (%d samples/hop), 3-slot pinned ring" % (C, T, N, N // 2),
                                 "channels": C, "hops_per_batch": T, "window": N}}), flush=True)


def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def visible_gpus():
    """GPU count as seen by a short-lived child, so that the launching parent never starts a HIP runtime of its own."""
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                             capture_output=True, text=True, timeout=300)
        return int(out.stdout.strip().splitlines()[-1])
    except Exception:
        return None


def self_launch(args):
    """`python bench.py --gpus N` from a bare shell: this process makes no GPU call of its own (the device count comes from
    a short-lived child); it starts the N ranks as fresh children (torch.distributed.run, one process per GPU) under
    rank_environment(), lets rank 0's JSON line through on stdout and exits with their status.  Never an exec of a
    GPU-initialised process."""
    n = visible_gpus()
    if args.backend == "nccl" and n is not None and n < args.gpus:
        raise SystemExit("--gpus %d but only %d GPU(s) visible; RCCL needs one GPU per rank "
                         "(use --backend gloo to let ranks share devices for debugging)" % (args.gpus, n))
    env = rank_environment(dict(os.environ))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def time_steps(an, frames, raw, sm, steps, warmup=2):
    """(frames/s, frame-kernel ms per launch) of `steps` process_frames calls on one analyser."""
    import torch
    if raw is None:
        raw = torch.empty((frames.shape[0], frames.shape[1], 12), dtype=torch.float32, device=frames.device)
    if sm is None:
        sm = torch.empty_like(raw)
    for _ in range(warmup):
        an.process_frames(frames, out_raw=raw, out_smoothed=sm)
    an.sync()
    an.profile_begin()
    t0 = time.perf_counter()
    for _ in range(steps):
        an.process_frames(frames, out_raw=raw, out_smoothed=sm)
    an.sync()
    dt = time.perf_counter() - t0
    fms, _, calls = an.profile_end()
    return frames.shape[0] * frames.shape[1] * steps / dt, fms / max(calls, 1)


class GpuEngine:
    """Where a rank's data lives and what analyses it: the product library on this rank's GPU.  (tests/test_sharded_cpu.py
    drives rank_main() with a CPU stand-in of the same shape to check the N>1 control flow and the shard arithmetic.)"""
    name = "gpu"

    def __init__(self, local_rank, backend):
        import torch
        self.torch = torch
        if backend == "gloo":
            local_rank = local_rank % max(1, torch.cuda.device_count())
        self.device = local_rank
        torch.cuda.set_device(local_rank)
        self.fx = importlib.import_module("feature-extractor_amd")
        self.rccl_capable = True

    def frames(self, host):
        return self.torch.from_numpy(host).cuda(self.device)

    def empty(self, shape):
        return self.torch.empty(shape, dtype=self.torch.float32, device="cuda:%d" % self.device)

    def analyser(self, count, window, **kw):
        return self.fx.BatchAnalyser(count, window, device=self.device, **kw)

    def synchronize(self):
        self.torch.cuda.synchronize()


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--channels-per-gpu", type=int, default=None,
                    help="default: 1024 at --gpus 1 (BASELINE configs[1]), 8192 at --gpus N>1 (configs[3])")
    ap.add_argument("--frames", type=int, default=None,
                    help="consecutive frames per channel per step (SURVEY 8d: T >= 64); default 512 at --gpus 1, 128 at --gpus N>1 "
                         "(8192 channels x 128 frames = 4.3 GB of input per GPU)")
    ap.add_argument("--window", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra measurements (spectral-only, noise / silence, other windows, live cadence)")
    ap.add_argument("--only-extra", default=None, help="comma-separated names: run only these extra measurements (e.g. osc_sink,sustained)")
    ap.add_argument("--sustained-seconds", type=float, default=12.0, help="length of the `sustained` extra's back-to-back run of the headline step")
    ap.add_argument("--no-pmc", action="store_true", help="do not read hardware counters in this run (rocprofv3 --pmc children)")
    ap.add_argument("--write-counters", action="store_true", help="also store this run's counters in profiles/counters.json (the labelled fallback)")
    ap.add_argument("--signal", default="synth", choices=["synth", "noise", "silence"],
                    help="synth = the BASELINE synthetic mix (default); noise / silence probe data-dependent paths")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl: the feature gather runs over RCCL through the C ABI (fx_gather_smoothed), one GPU per rank.  "
                         "gloo: debugging aid for boxes with fewer GPUs than ranks (ranks share devices, features are gathered through host memory)")
    ap.add_argument("--stream", action="store_true",
                    help="host-resident hops through the pinned ring (fx_stream_*): PCIe-inclusive rate, reported as an extra line")
    ap.add_argument("--fp16", action="store_true", help="--stream only: fp16 samples")
    ap.add_argument("--debug-collective", action="store_true",
                    help="with --gpus 1: create a one-rank RCCL communicator and run the N>1 code path (gather included), "
                         "then check the gathered block against the local one")
    ap.add_argument("--rank-timeout", type=int, default=600, help="seconds a rank may spend joining the group / communicator or in one barrier before it gives up")
    ap.add_argument("--analysers", default="both", choices=["both", "spectral", "harmonic"],
                    help="--pmc-child only (tools/pmc_quick.py): which of the reference's two analysers the profiled context constructs")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--input-file", default=None, help=argparse.SUPPRESS)
    return ap


def main():
    args = build_parser().parse_args()
    rank_environment()                      # before anything can start the HIP runtime (both launch forms)
    if args.pmc_child:
        return pmc_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    out = rank_main(args, GpuEngine(local_rank, args.backend), rank, world)
    if out is not None:
        print(json.dumps(out), flush=True)


def rank_main(args, engine, rank, world):
    """One rank of the bench: its shard of the channels, the timed steps, the gather to the sink; rank 0 returns the line."""
    import torch
    import torch.distributed as dist
    from datetime import timedelta

    collective = world > 1 or args.debug_collective
    if collective:
        # control plane (barriers, the communicator id, the max over ranks): a CPU group.  The data-path exchange is
        # RCCL inside the library.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        with Watchdog(args.rank_timeout, "joining the gloo control group (world %d)" % world):
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=args.rank_timeout))

    fx = importlib.import_module("feature-extractor_amd")
    sharded = importlib.import_module("feature-extractor_amd.sharded")
    N = args.window
    C = args.channels_per_gpu if args.channels_per_gpu is not None else (1024 if world == 1 else 8192)
    T = args.frames if args.frames is not None else (512 if world == 1 else 128)
    if args.stream:
        return stream_bench(fx, args, C, T, N, engine.device)
    total_channels = C * world
    first, count = sharded.my_shard(total_channels, rank, world)

    host_frames = fx.synth.frames(count, T, N, first_channel=first)
    if args.signal == "noise":
        host_frames = np.random.default_rng(first).normal(0, 0.1, host_frames.shape).astype(np.float32)
    elif args.signal == "silence":
        host_frames = np.zeros_like(host_frames)
    want_pmc = (world == 1 and engine.name == "gpu" and not args.no_pmc and not args.debug_collective)
    input_file = None
    if want_pmc:
        try:
            fd, input_file = tempfile.mkstemp(prefix="fx_bench_in_", suffix=".npy", dir="/tmp")
            os.close(fd)
            np.save(input_file, host_frames)      # the counter passes analyse the same bytes
        except OSError:
            input_file = None
    frames = engine.frames(host_frames)
    del host_frames
    an = engine.analyser(count, N)
    raw = engine.empty((count, T, 12))
    sm = engine.empty((count, T, 12))

    # What travels between GPUs is what the OSC sink samples (ref OSCFeatureAnalysisOutput.h:89-113): the latest
    # smoothed vector of every channel, [C][12] per rank per step, gathered to rank 0.
    rccl = collective and args.backend == "nccl" and engine.rccl_capable
    gathered = None
    if rccl:
        ident = [fx.BatchAnalyser.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0)
        with Watchdog(args.rank_timeout, "fx_comm_create (ncclCommInitRank, world %d)" % world):
            an.comm_create(rank, world, ident[0])
        total, firsts = an.comm_layout()
        if total != total_channels or firsts[rank] != first:
            raise SystemExit("rank %d: communicator layout (%d channels, mine from %d) differs from the shard plan (%d, %d)"
                             % (rank, total, firsts[rank], total_channels, first))
        if rank == 0:      # two destination buffers: a consumer may read one while the next gather fills the other
            gathered = [engine.empty((total_channels, 12)) for _ in range(2)]
    host_pending = [None]
    counter = [0]

    def drain():
        an.sync()
        if rccl:
            an.comm_sync()
        elif host_pending[0] is not None and host_pending[0][1] is not None:
            host_pending[0][1].wait()

    def barrier():
        drain()
        engine.synchronize()
        if collective:
            dist.barrier()
            engine.synchronize()

    def step():
        slot = counter[0] & 1
        counter[0] += 1
        an.process_frames(frames, out_raw=raw, out_smoothed=sm)
        if rccl:
            # asynchronous, device-ordered: snapshot on the library's stream, RCCL on its side stream
            an.gather_features(dst=0, out=gathered[slot] if rank == 0 else None)
        elif collective:
            if host_pending[0] is not None and host_pending[0][1] is not None:
                host_pending[0][1].wait()
            host_pending[0] = sharded.gather_features(torch.from_numpy(an.get_features()), total_channels, dst=0,
                                                      async_op=True, single_rank_collective=True)

    if collective:
        with Watchdog(args.rank_timeout, "the first gather (communicator channels are set up here)"):
            step()              # one untimed exchange so that the communicator's channels exist even with --warmup 0
            drain()
    for _ in range(args.warmup):
        step()
    barrier()
    an.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    frame_ms, epi_ms, calls = an.profile_end()

    per_rank = None
    if collective:
        # every rank's own clock, kernel time and view of the exchange, so that a sub-linear scaling result can be read off one run:
        # a slow rank, a slow gather or a communicator of the wrong size each show up in their own column
        mine = {"rank": rank, "device": getattr(engine, "device", None), "channels": count, "seconds": dt,
                "frames_per_s": count * T * args.steps / dt, "frame_kernel_ms_per_step": frame_ms / max(calls, 1),
                "tail_kernels_ms_per_step": epi_ms / max(calls, 1)}
        if rccl:
            an.comm_sync()
            mine["exchange"] = an.comm_stats()
        gathered_records = [None] * world
        dist.all_gather_object(gathered_records, mine)
        per_rank = gathered_records
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if collective and rank == 0:
        # the sink's copy of this rank's block must be what the analyser holds
        want = torch.from_numpy(an.get_features())
        if rccl:
            got = gathered[(counter[0] - 1) & 1][first:first + count].cpu()
        else:
            got = sharded.resolve(host_pending[0][0])[first:first + count]
        if not torch.equal(torch.nan_to_num(want), torch.nan_to_num(sm[:, -1, :].cpu())):
            raise SystemExit("latest vectors differ from the last frame's smoothed vectors")
        same = bool(torch.equal(torch.nan_to_num(got), torch.nan_to_num(want)))
        print("collective check: gathered block %s the local features (%s, %d rank(s))"
              % ("equals" if same else "DIFFERS FROM", "RCCL via fx_gather_smoothed" if rccl else "gloo via host", world),
              file=sys.stderr, flush=True)
        if not same:
            raise SystemExit(3)

    # RCCL writes its banner through C stdio, which is block-buffered on a pipe and would otherwise surface after
    # the JSON line (from any rank) when the processes exit: push it out now, on every rank, before rank 0 prints
    _flush_c_stdio()
    if collective:
        dist.barrier()
    out = None
    if rank == 0:
        frames_total = total_channels * T * args.steps
        value = frames_total / dt
        bytes_per_frame = 4 * N + 48                     # sample bytes + 12 floats out
        launch_bytes = bytes_per_frame * count * T
        avg_launch_s = frame_ms / 1e3 / max(calls, 1)
        achieved = launch_bytes / avg_launch_s / 1e9
        achieved_tflops = flops_per_frame(N) * count * T / avg_launch_s / 1e12
        cfg = "configs[1]" if world == 1 else "configs[3]"
        roof = {"bound": "valu",
                "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                "kernel": "fx_frame_kernel<%d>" % N, "avg_launch_ms": avg_launch_s * 1e3,
                "algorithmic_bytes_per_launch": launch_bytes, "launches_timed": calls,
                "epilogue_ms_per_step": epi_ms / max(calls, 1),
                "compute": {"flops_per_frame": flops_per_frame(N), "achieved_tflops": achieved_tflops,
                            "peak_tflops": FP32_VECTOR_PEAK_TFLOPS, "peak_no_fma": FP32_NO_FMA_PEAK_TFLOPS,
                            "frac": achieved_tflops / FP32_NO_FMA_PEAK_TFLOPS, "frac_of_fma_peak": achieved_tflops / FP32_VECTOR_PEAK_TFLOPS,
                            "note": "algorithmic flops (SURVEY 8d: 4 FFTs x 5 N log2 N + 30 N) x frames per launch / the frame kernel's "
                                    "HIP-event time; the FFT's rounding DAG is the reference's, which never fuses a*b+c, so the "
                                    "usable vector peak is the no-FMA half; `frac` is against that"},
                "note": "achieved / peak / frac are SURVEY 8(d)'s HBM figures (algorithmic bytes = (4*N + 48) B/frame x frames per launch / "
                        "HIP-event kernel time against 8 TB/s): HBM traffic is ~1.05x the algorithmic bytes, nothing is re-read, and at "
                        "~57 flop/B the binding unit is VALU issue -- `bound` names it and `compute` is its roofline"}
        if want_pmc and input_file:
            pmc, note = measure_counters(N, count, T, input_file)
            if pmc is None:
                rec, why = committed_counters(N, count, T)
                roof["counters_source"] = "not read in this run (%s); %s" % (note, why)
                pmc = rec["pmc"] if rec else None
            else:
                roof["counters_source"] = note
            if pmc:
                roof.update(counters_to_fields(pmc, count * T))
                if args.write_counters and "rocprofv3" in roof["counters_source"] and "not read" not in roof["counters_source"]:
                    path = os.path.join(ROOT, "profiles", "counters.json")
                    try:
                        allrec = json.load(open(path))
                    except Exception:
                        allrec = {}
                    allrec["%d:%d:%d" % (N, count, T)] = {"pmc": pmc, "kernel_sources_sha": kernel_sources_sha(),
                                                          "avg_launch_ms_unprofiled": avg_launch_s * 1e3}
                    json.dump(allrec, open(path, "w"), indent=1, sort_keys=True)
        else:
            roof["counters_source"] = "not requested (--no-pmc, N>1 or debug run)"
        out = {
            "metric": "frames/sec (1024-pt FFT, 10-feature bundle)" if N == 1024 else "frames/sec (%d-pt FFT, 10-feature bundle)" % N,
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" if args.signal == "synth" else args.signal,
            "config": {"workload": "%s: %d channels/GPU x %d-pt fp32 frames, %d consecutive frames per channel per step, "
                                   "full 12-feature RealTimeAnalyser bundle (spectral+pitch+harmonic+RMS+smoothing+onset), "
                                   "frames pre-assembled and resident in HBM" % (cfg, C, N, T),
                       "channels_per_gpu": C, "total_channels": total_channels, "frames_per_step": T, "window": N, "sample_rate": 48000,
                       "sharding": ("channels, contiguous blocks of %d; per step a gather of the latest smoothed vectors [C][12] of every rank "
                                    "to rank 0 (the OSC sink) %s" % (C, "over RCCL (fx_gather_smoothed, grouped ncclSend/ncclRecv on a side stream)"
                                                                     if rccl else "through host memory (gloo debugging backend)"))
                                   if collective else "single GPU"},
            "roofline": roof,
        }
        if per_rank is not None:
            out["per_rank"] = per_rank
        if world == 1 and not args.no_extra and not args.debug_collective and engine.name == "gpu":
            extras(out, args, engine, fx, an, frames, raw, sm, N, count, T, launch_bytes, avg_launch_s,
                   spectral_input=input_file if (want_pmc and not under_profiler()) else None)
        if input_file:
            try:
                os.unlink(input_file)
            except OSError:
                pass
        if not args.no_cpu_baseline and world == 1 and not args.debug_collective:
            try:
                out["cpu_baseline"] = cpu_baseline(fx, N, T)
            except Exception as e:          # the headline line must come out whatever happens to a side measurement
                out["cpu_baseline"] = {"error": str(e)}
        else:
            out["cpu_baseline"] = None

    if collective:
        dist.barrier()
        if rccl:
            an.comm_destroy()
        dist.destroy_process_group()
    return out


def extras(out, args, engine, fx, an, frames, raw, sm, N, count, T, launch_bytes, avg_launch_s, spectral_input=None):
    """Side measurements on the same line (never `value`).  Each one is guarded: a failure is reported in its own field
    and the headline line still comes out."""
    import torch
    extra_steps = 10
    dev = engine.device

    def guarded(name, fn):
        try:
            out[name] = fn()
        except Exception as e:
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}

    def spectral_only():
        # BASELINE configs[1] read literally is the spectral analyser alone ("fused window + FFT + magnitude +
        # SpectralCharacteristics reductions in one kernel"): the same workload with only the RealTimeSpectralAnalyser
        an_s = fx.BatchAnalyser(count, N, device=dev, analysers="spectral")
        # (best of two passes of 20 steps, as for the other windows: the first launches on a fresh context -- first touch of 90 MB of scratch
        # records, clocks -- ran this kernel 15 % slow, which is what rounds 2 and 3 reported for it)
        fps, fms = max(time_steps(an_s, frames, raw, sm, 2 * extra_steps, warmup=5) for _ in range(2))
        an_s.close()
        ach = launch_bytes / (fms / 1e3) / 1e9
        rec = {"value": fps, "unit": "frames/s", "kernel": "fx_frame_kernel<%d, spectral>" % N, "avg_launch_ms": fms,
               "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS},
               "note": "FX_SPECTRAL_ONLY: RMS, centroid, spread, flatness, LER, flux, slope, onset (8 of the 12 slots)"}
        # what binds it: one SQ counter pass of this kernel, read now (a rocprofv3 child, as for the headline kernel)
        if spectral_input is not None:
            pmc, why = measure_counters(N, count, T, spectral_input, passes=PMC_PASSES[:1], analysers="spectral")
            if pmc:
                f = counters_to_fields(pmc, count * T)
                rec["valu_insts_per_frame"] = f.get("valu_insts_per_frame")
                rec["valu_active_per_wave"] = f.get("valu_active_per_wave")
                rec["valu_active_x_waves_per_simd"] = f.get("valu_active_per_wave", 0.0) * 4.0
                rec["counters_note"] = "SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x the 4 wavefronts per SIMD the LDS holds: the share of the SIMD's issue slots this kernel uses (%s)" % why
            else:
                rec["counters_note"] = "not read: %s" % why
        return rec

    def data_dependence():
        # data-dependent paths (ref PitchAnalyser.h:161-190: the lag search ends early on tonal input and runs to the
        # arg-min fallback on noise; silence takes every early exit): same shape, other contents
        dd = {}
        for name in ("noise", "silence"):
            if name == "noise":
                g = torch.Generator(device=frames.device)
                g.manual_seed(1234)
                other = torch.randn(frames.shape, generator=g, device=frames.device, dtype=torch.float32) * 0.1
            else:
                other = torch.zeros_like(frames)
            fps, fms = time_steps(an, other, raw, sm, extra_steps)
            dd[name] = {"value": fps, "unit": "frames/s", "frame_kernel_ms": fms, "relative_to_synth": (avg_launch_s * 1e3) / fms}
            del other
        dd["note"] = "same shape as the headline run; noise = N(0, 0.1^2) white, silence = zeros; relative_to_synth = synth kernel time / this kernel time"
        an.reset_state()
        return dd

    def other_windows():
        # the reference application's default window (AnalyserTrackController.h:20-21) and the streaming config's window
        others = {}
        for (n2, c2, t2, label) in ((2048, 4096, 64, "configs[2] shape: 4096 channels x 2048-pt"), (4096, 1024, 64, "configs[4] window: 1024 channels x 4096-pt")):
            if n2 == N:
                continue
            fr2 = torch.from_numpy(fx.synth.frames(c2, t2, n2)).cuda(dev)
            an2 = fx.BatchAnalyser(c2, n2, device=dev)
            # (best of two passes: the first launches on a fresh context run up to 8 % slow -- clocks, first touch of the
            # scratch buffers -- and this is an extra, not the timed region)
            fps, fms = max(time_steps(an2, fr2, None, None, 2 * extra_steps, warmup=5) for _ in range(2))
            an2.close()
            b2 = (4 * n2 + 48) * c2 * t2
            tf = flops_per_frame(n2) * c2 * t2 / (fms / 1e3) / 1e12
            others[str(n2)] = {"value": fps, "unit": "frames/s", "frame_kernel_ms": fms, "workload": "%s, %d frames per step" % (label, t2),
                               "hbm_frac": b2 / (fms / 1e3) / 1e9 / HBM_PEAK_GBPS, "achieved_tflops": tf, "compute_frac_no_fma": tf / FP32_NO_FMA_PEAK_TFLOPS}
            # the same launches in the low-latency family: every frame on a PAIR of wavefronts (fx_pair_kernel, FX_LOW_LATENCY)
            an2 = fx.BatchAnalyser(c2, n2, device=dev, low_latency=True)
            fps_p, fms_p = max(time_steps(an2, fr2, None, None, 2 * extra_steps, warmup=5) for _ in range(2))
            an2.close()
            others[str(n2)]["pair_kernel"] = {"value": fps_p, "unit": "frames/s", "frame_kernel_ms": fms_p}
            if n2 == 4096:
                # which of the 4096-point kernel's two twiddle paths this host's cos / sin select (include/fx.h, fx_twiddle_symmetry): bit 1 set =
                # two rows of the last pass formed from LDS as quarter turns; clear = read from the table in global memory (same values, ~3 % slower)
                others[str(n2)]["twiddle_symmetry"] = int(fx.load_library(build_if_missing=False).fx_twiddle_symmetry(4096))
            del fr2
        return others

    def live_cadence():
        # The reference's own cadence at scale: every channel's analysers run once per hop as it arrives
        # (ref AudioDataCollector.h:66-94, RealTimeAnalyser.h:201-234): MANY channels x ONE hop per call, device-resident
        # hops through fx_push_hops (which picks fx_hop_kernel or the batch kernels by the size of the call).
        res = {}
        for c3 in (1024, 8192):
            hops = torch.from_numpy(fx.synth.hops(c3, 16, N)).cuda(dev)
            an3 = fx.BatchAnalyser(c3, N, device=dev)
            r3 = torch.empty((c3, 1, 12), dtype=torch.float32, device=hops.device)
            s3 = torch.empty_like(r3)
            views = [hops[:, k:k + 1].contiguous() for k in range(16)]
            n_calls = 400
            torch.cuda.synchronize(dev)
            with torch.cuda.stream(an3.torch_stream()):          # the caller works on the library's stream: no cross-stream waits per call
                for k in range(n_calls + 50):
                    if k == 50:
                        an3.sync()
                        t_s = time.perf_counter()
                    an3.push_hops(views[k % 16], out_raw=r3, out_smoothed=s3)
                an3.sync()
                dt3 = time.perf_counter() - t_s
            an3.close()
            res[str(c3)] = {"value": c3 * n_calls / dt3, "unit": "frames/s", "us_per_call": dt3 / n_calls * 1e6,
                            "real_time_factor": (c3 * n_calls / dt3) / (c3 * 48000.0 / (N // 2))}
            del hops, views
        res["note"] = ("%d-pt windows, ONE hop (%d samples) per channel per call, calls back to back on the library's stream (fx_push_hops, device-resident "
                       "hops; up to 2^20 samples per call one launch of fx_hop_kernel -- three wavefronts per channel + the hop's tail -- above "
                       "that the one-frame frame kernel -- a wavefront per channel, eight channels per workgroup, flux state in global memory -- and the hops' "
                       "tails on a quarter wavefront per channel, in the same launch while the chip holds all workgroups at once); real_time_factor = "
                       "frames/s over the frames/s that many live 48 kHz channels produce" % (N, N // 2))
        return res

    def device_blocks():
        # The collector's real interface at scale (fx_push_samples, round 5; ref AudioDataCollector.h:36-94, RealTimeAudioAnalysis.h:205-219):
        # an audio device delivers blocks of ITS length -- 480 samples at 48 kHz / 10 ms -- not hops.  Every call cuts (pending + block) into
        # hops on the device (fx_reblock_kernel: pure byte movement), analyses the hops that completed and keeps the rest.  Device-resident
        # blocks, calls back to back on the library's stream, beside the same samples delivered as whole hops.
        res = {}
        block, n_blocks = 480, 64                                  # 30 720 samples = 60 hops of 512 per channel
        for c4 in (1024, 8192):
            hops = torch.from_numpy(fx.synth.hops(c4, block * n_blocks // (N // 2), N)).cuda(dev)
            flat = hops.reshape(c4, -1)
            pieces = [flat[:, k * block:(k + 1) * block].contiguous() for k in range(n_blocks)]
            views = [hops[:, k:k + 1].contiguous() for k in range(hops.shape[1])]
            an4 = fx.BatchAnalyser(c4, N, device=dev)
            rec = {}
            with torch.cuda.stream(an4.torch_stream()):
                def one_pass(name):
                    an4.reset_state()
                    t_s = time.perf_counter()
                    if name == "blocks":
                        for piece in pieces:
                            an4.push_samples(piece)
                    else:
                        for v in views:
                            an4.push_hops(v)
                    an4.sync()
                    return time.perf_counter() - t_s

                # one untimed pass of each, then the two ways ALTERNATE, best of four each: whichever is measured first on a context that has just
                # been created runs ~4 % slow (clocks, first touches), and rounds 5 / 6a measured `blocks` first and `hops` second
                best = {"blocks": None, "hops": None}
                for name in best:
                    one_pass(name)
                for _ in range(4):
                    for name in ("hops", "blocks"):
                        dt4 = one_pass(name)
                        best[name] = dt4 if best[name] is None or dt4 < best[name] else best[name]
                for name in ("blocks", "hops"):
                    calls = n_blocks if name == "blocks" else len(views)
                    rec[name] = {"calls": calls, "us_per_call": best[name] / calls * 1e6, "frames_per_s": c4 * len(views) / best[name],
                                 "real_time_factor": (c4 * len(views) / best[name]) / (c4 * 48000.0 / (N // 2))}
                # one call with every sample: the re-blocking kernel's share of a batch call (whole hops from an aligned buffer are analysed in place,
                # so the block is offset by one pending sample)
                first, rest = flat[:, :1].contiguous(), flat[:, 1:].contiguous()
                dt_all = dt_ref = None
                for _ in range(3):                                  # (the first pass allocates the library's hop buffer and scratch: best of three)
                    an4.reset_state()
                    an4.push_samples(first)
                    an4.sync()
                    t_s = time.perf_counter()
                    an4.push_samples(rest)
                    an4.sync()
                    d = time.perf_counter() - t_s
                    dt_all = d if dt_all is None or d < dt_all else dt_all
                    an4.reset_state()
                    t_s = time.perf_counter()
                    an4.push_hops(hops)
                    an4.sync()
                    d = time.perf_counter() - t_s
                    dt_ref = d if dt_ref is None or d < dt_ref else dt_ref
            rec["one_call"] = {"samples_per_channel": int(flat.shape[1] - 1), "ms_through_push_samples": dt_all * 1e3, "ms_as_whole_hops": dt_ref * 1e3,
                               "sample_bytes": int((flat.shape[1] - 1) * c4 * 4),
                               "note": "1024-point windows: the batch kernel reads [pending | block] itself (no re-blocking pass since round 6)"}
            if c4 == 1024:
                # the same blocks from HOST memory, where an audio callback has them: fx_push_samples on a host block (a synchronous call: copy in,
                # analysis, vectors out) and the pinned ring's form (fx_stream_push_samples / fx_stream_collect_samples, three blocks in flight)
                host_pieces = [p4.cpu().numpy() for p4 in pieces[:8]]
                an4.reset_state()
                for k in range(8):
                    an4.push_samples(host_pieces[k % 8])
                an4.reset_state()
                t_s = time.perf_counter()
                for k in range(n_blocks):
                    an4.push_samples(host_pieces[k % 8])
                dt_h = time.perf_counter() - t_s
                an4.reset_state()
                ring4 = fx.HopStream(an4, 1, slots=3, dtype=np.float32)
                got = 0
                t_s = None
                warm4 = 400                                       # (a new ring's first few hundred blocks carry its allocations' first touches: 80 us in steady state, spikes of milliseconds before)
                for k in range(n_blocks + warm4):
                    if k == warm4:
                        while ring4.in_flight():
                            ring4.collect_samples()
                        t_s = time.perf_counter()
                    if ring4.in_flight() == 3:
                        got += ring4.collect_samples()[0].shape[1]
                    ring4.push_samples(host_pieces[k % 8])
                while ring4.in_flight():
                    got += ring4.collect_samples()[0].shape[1]
                dt_r = time.perf_counter() - t_s
                ring4.close()
                rec["from_host_memory"] = {"push_samples_us_per_call": dt_h / n_blocks * 1e6, "ring_us_per_block": dt_r / n_blocks * 1e6,
                                           "block_bytes": int(host_pieces[0].nbytes), "real_time_factor_ring": (block / 48000.0) / (dt_r / n_blocks)}
            an4.close()
            res[str(c4)] = rec
            del hops, flat, pieces, views
        # the byte mover behind it, alone: a block that completes no hop (the call is the re-blocking launch and nothing else), HIP events on the library's
        # stream behind a 1 GB fill so that the launch is queued when the events pass and the Infinity Cache is cold.  HBM-bound: 2 x the sample bytes.
        an6 = fx.BatchAnalyser(16384, 4096, device=dev)
        x6 = torch.rand((16384, 2047), device="cuda:%d" % dev) - 0.5
        busy = torch.empty(1 << 30, dtype=torch.uint8, device="cuda:%d" % dev)
        lib6 = an6.torch_stream()
        best6 = None
        with torch.cuda.stream(lib6):
            for _ in range(6):
                an6.reset_state()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                busy.zero_()
                e0.record(lib6)
                an6.push_samples(x6, want_raw=False, want_smoothed=False)
                e1.record(lib6)
                an6.sync()
                ms6 = e0.elapsed_time(e1)
                best6 = ms6 if best6 is None or ms6 < best6 else best6
        an6.close()
        moved6 = 2 * x6.numel() * 4
        res["reblock_kernel"] = {"workload": "16384 channels x 2047 fp32 samples onto an empty carry (no hop completes)", "bytes_moved": moved6, "us": best6 * 1e3,
                                 "roofline": {"bound": "hbm", "achieved": moved6 / (best6 / 1e3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                              "frac": moved6 / (best6 / 1e3) / 1e9 / HBM_PEAK_GBPS,
                                              "note": "algorithmic bytes = sample bytes read + written; from cold HBM (a 1 GB fill runs in front); the runtime's own "
                                                      "device-to-device copy of the same bytes reaches 3.7-3.9 TB/s measured the same way (profiles/r05_reblock_rate.txt)"}}
        del x6, busy
        res["note"] = ("%d-pt windows; `blocks`: %d calls of fx_push_samples with %d-sample device blocks per channel (an audio device at 48 kHz / 10 ms) = the "
                       "same %d hops per channel that `hops` delivers as one fx_push_hops call per hop; best of three passes; real_time_factor as in live_cadence"
                       % (N, n_blocks, block, block * n_blocks // (N // 2)))
        return res

    def streaming_hop():
        # BASELINE configs[4]: 1 channel, 4096-pt windows, fp16 samples, ONE hop per call through the pinned ring (one launch of
        # fx_hop_kernel / fx_hop_pair_kernel, the host polls a flag): the round trip of a hop, for the default kernel family and for
        # FX_LOW_LATENCY (every analyser on a pair of wavefronts), as this interpreter sees it and as a C++ host sees it
        # (tools/stream_latency.cpp, built by __graft_entry__.build(): a child process, nothing is exec'ed from here)
        res = {"unit": "us per 2048-sample hop", "workload": "configs[4]: 1 channel x 4096-pt windows, fp16 samples, one hop per call, submit + collect"}
        h4 = fx.synth.hops(1, 64, 4096, first_channel=24).astype(np.float16)
        for name, low in (("default", False), ("low_latency", True)):
            an4 = fx.BatchAnalyser(1, 4096, device=dev, low_latency=low)
            st4 = fx.HopStream(an4, 1, slots=3, dtype=np.float16)
            n_calls = 2000
            for k in range(n_calls + 200):
                if k == 200:
                    t_s = time.perf_counter()
                st4.slot()[...] = h4[:, k % 64:k % 64 + 1]
                st4.submit()
                st4.collect(want_raw=False)
            res[name] = {"python_round_trip_us": (time.perf_counter() - t_s) / n_calls * 1e6, "calls": n_calls}
            st4.close()
            an4.close()
            exe = os.path.join(ROOT, "tools", "_bin", "stream_latency")
            if os.path.exists(exe):
                p = subprocess.run([exe, "4096", "1", "1", "4000", "tone", "lowlat" if low else "default"], capture_output=True, text=True, timeout=120)
                for line in p.stdout.splitlines():
                    if line.startswith("JSON "):
                        res[name]["c_host"] = json.loads(line[5:])
                if p.returncode != 0 or "c_host" not in res[name]:
                    res[name]["c_host"] = {"error": (p.stderr or p.stdout)[-300:]}
            else:
                res[name]["c_host"] = {"error": "tools/_bin/stream_latency not built (__graft_entry__.build())"}
        res["round_trip_us"] = res["default"]["python_round_trip_us"]          # (the field earlier rounds reported)
        return res

    def stream_ingest():
        # The path a caller with HOST audio stands on (ref AudioDataCollector.h:36-94, AudioFilePlayer.h:41-61): hops in
        # host memory -> the pinned ring (fx_stream_*) -> H2D on a side stream || analysis || vectors back on a third queue -> host.
        # PCIe-inclusive by construction, never `value`.  Its roofline is the H2D link: `peak` = what hipMemcpyAsync from pinned
        # memory reaches for the same bytes in this run, `achieved` = sample bytes per second the ring really moved.  The producer's
        # batch sits in ordinary memory and is copied into the slot by `fill_threads` threads (fx_stream_push: a thread pool inside
        # the library; the reference's per-channel collectors write their own channel's samples); the same loop with the slots left
        # as they are (producer excluded) is reported beside it.
        cores = usable_cores()
        # (measured on the 16-CPU quota of these boxes, fp32, 1024 ch x 64 hops: 3 slots x 12 fill threads 0.78-0.79 of the memcpy rate, 3 x 8 0.96,
        # 4 x 6 0.96, 4 x 12 0.92-0.93, 5 x 12 0.95: the fill pool must leave cores to the caller's thread and to the runtime's own)
        fill_threads = max(4, min(8, cores // 2))
        ring_slots = 4
        res = {"fill_threads": fill_threads, "slots": ring_slots,
               "note": "frames/s with host-resident hops, PCIe-inclusive (never `value`); roofline.bound = pcie: achieved = sample bytes/s through the "
                       "ring with the producer filling slots in place (median of three passes; frac_min / frac_max = the slowest / fastest pass), "
                       "peak = the best of five groups of pinned hipMemcpyAsync H2D copies of the shape's byte counts, measured once per shape in this run"}

        def memcpy_rate(sizes, runs=5, copies=4):
            """The link as this run finds it: the BEST of `runs` timed groups of pinned hipMemcpyAsync H2D copies, over the byte counts of the
            shape's formats, after a warm-up copy.  One figure per shape: the denominator of every format's fraction (round 4 measured it once
            per format with a single group, and a slow group made a fraction above 1)."""
            best = 0.0
            for nbytes in sizes:
                host = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
                devb = torch.empty(nbytes, dtype=torch.uint8, device="cuda:%d" % dev)
                devb.copy_(host, non_blocking=True)
                torch.cuda.synchronize(dev)
                for _ in range(runs):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(copies):
                        devb.copy_(host, non_blocking=True)
                    e1.record()
                    torch.cuda.synchronize(dev)
                    best = max(best, nbytes * copies / (e0.elapsed_time(e1) / 1e3) / 1e9)
                del host, devb
            return best

        for (n2, c2, t2) in ((1024, 1024, 64), (2048, 4096, 32)):
            base = fx.synth.hops(128, t2, n2)
            src32 = np.ascontiguousarray(np.tile(base, (c2 // 128, 1, 1)))
            peak = memcpy_rate([src32.nbytes, src32.nbytes // 2])
            shape = {"workload": "%d channels x %d hops per batch x %d-pt windows (%d new samples per frame)" % (c2, t2, n2, n2 // 2),
                     "pinned_h2d_GBps": peak, "passes": 3}
            for fmt in ("f32", "f16", "s16", "s24"):
                if fmt == "s24":          # packed 24-bit PCM: three bytes per sample, [C][T][3 N/2] bytes
                    src = fx.pack_s24(np.round(src32.astype(np.float64) * 8388607.0).astype(np.int32))
                else:
                    src = src32 if fmt == "f32" else (src32.astype(np.float16) if fmt == "f16" else np.round(src32 * 32767.0).astype(np.int16))
                an5 = fx.BatchAnalyser(c2, n2, device=dev)
                st5 = fx.HopStream(an5, t2, slots=ring_slots, dtype=src.dtype)

                def run(fill, steps, warm=4):
                    for k in range(steps + warm):
                        if k == warm:
                            while st5.in_flight():
                                st5.collect(want_raw=False)
                            t0 = time.perf_counter()
                        if st5.in_flight() == ring_slots:
                            st5.collect(want_raw=False)
                        if fill or k < ring_slots:
                            st5.push(src, fill_threads=fill_threads)
                        else:
                            st5.slot()
                            st5.submit()
                    while st5.in_flight():
                        st5.collect(want_raw=False)
                    return time.perf_counter() - t0

                # three passes of each loop: the host side of this path shares its memory system with whoever else is on the machine, and one
                # pass can be anywhere between 0.6 and 0.95 of the link on the same box (tools/ingest_sweep.py); `value` is the median
                steps5 = 24
                fills = sorted(run(True, steps5) for _ in range(3))
                nofills = sorted(run(False, steps5) for _ in range(3))
                dt_fill, dt_nofill = fills[1], nofills[1]
                st5.close()
                an5.close()
                fr = c2 * t2 * steps5
                gbps = lambda dt: src.nbytes * steps5 / dt / 1e9
                shape[fmt] = {"value": fr / dt_fill, "unit": "frames/s", "ms_per_batch": dt_fill / steps5 * 1e3, "bytes_per_batch": src.nbytes,
                              "value_min": fr / fills[2], "value_max": fr / fills[0],
                              "producer_excluded": {"value": fr / dt_nofill, "h2d_GBps": gbps(dt_nofill), "value_min": fr / nofills[2], "value_max": fr / nofills[0]},
                              "roofline": {"bound": "pcie", "achieved": gbps(dt_fill), "peak": peak, "unit": "GB/s",
                                           "frac": gbps(dt_fill) / peak, "frac_min": gbps(fills[2]) / peak, "frac_max": gbps(fills[0]) / peak,
                                           "frac_producer_excluded": gbps(dt_nofill) / peak}}
                del src
            res[str(n2)] = shape
            del src32
        return res

    def offline():
        # The legacy offline analyser's three full-spectrum functions (ref AudioAnalysis.h:463-515, :566-609, :623-665) on
        # device-resident magnitude frames: a block per channel; since round 5 the per-bin TERMS (incl. the pow() calls) are formed by the whole
        # block in parallel into LDS and only the additions / the IEEE product run serially, in the reference's bin order (round 6: a lane per
        # chain, loads ahead of the additions) -- exact rather than fast.  Timed so that the cost of that choice is on record: calls per second of 1024 analysers x one 1025-bin
        # frame (a 2048-point window's magnitudes).  (Round 4's numbers were of the form where one thread did everything.)
        import ctypes
        lib = fx.load_library(build_if_missing=False)
        C5, B5 = 1024, 1025
        an5 = fx.offline.AudioAnalyser(C5, 24000.0, device=dev)
        g = torch.Generator(device="cuda:%d" % dev).manual_seed(5)
        mags = torch.rand((C5, B5), generator=g, device="cuda:%d" % dev, dtype=torch.float32)
        data = torch.randn((C5, B5 - 1, 2), generator=g, device="cuda:%d" % dev, dtype=torch.float32)
        out4 = torch.empty((C5, 4), device=mags.device, dtype=torch.float32)
        out1 = torch.empty((C5,), device=mags.device, dtype=torch.float32)
        out3 = torch.empty((C5, 3), device=mags.device, dtype=torch.float32)
        peaks = torch.empty((C5,), device=mags.device, dtype=torch.int32)
        freqs = torch.empty((C5,), device=mags.device, dtype=torch.float64)
        vp = ctypes.c_void_p
        calls = {
            "spectral_characteristics": lambda: lib.fx_offline_spectral_characteristics(an5._h, vp(mags.data_ptr()), B5, vp(out4.data_ptr()), fx.capi.MEM_DEVICE),
            "spectral_slope": lambda: lib.fx_offline_spectral_slope(an5._h, vp(mags.data_ptr()), B5, vp(out1.data_ptr()), fx.capi.MEM_DEVICE),
            "auto_correlation": lambda: lib.fx_offline_auto_correlation(an5._h, vp(data.data_ptr()), B5 - 1, vp(peaks.data_ptr()), vp(freqs.data_ptr()), fx.capi.MEM_DEVICE),
            "harmonic_characteristics": lambda: lib.fx_offline_harmonic_characteristics(an5._h, vp(mags.data_ptr()), B5, vp(out3.data_ptr()), fx.capi.MEM_DEVICE),
        }
        torch.cuda.synchronize(dev)
        res = {"workload": "%d analysers x one frame of %d magnitudes per call, device-resident" % (C5, B5),
               "note": "a block per channel: per-bin terms formed in parallel into LDS, every sum / the IEEE product serial in the reference's bin order on a lane of its own "
                       "(exactness over speed: what is left is the latency of 2 x bins dependent fp64 additions, ~10 ns each)"}
        for name, call in calls.items():
            for _ in range(3):
                fx.capi.check(call())
            fx.capi.check(lib.fx_offline_sync(an5._h))
            t0 = time.perf_counter()
            for _ in range(20):
                fx.capi.check(call())
            fx.capi.check(lib.fx_offline_sync(an5._h))
            dt = (time.perf_counter() - t0) / 20
            res[name] = {"us_per_call": dt * 1e6, "frames_per_s": C5 / dt, "bytes_in_per_call": int(mags.numel() * 4 if name != "auto_correlation" else data.numel() * 4),
                         "GBps": (mags.numel() * 4 if name != "auto_correlation" else data.numel() * 4) / dt / 1e9}
        an5.close()
        return res

    def osc_sink():
        # The sink at scale (ref OSCFeatureAnalysisOutput.h:84-113,133: one message per track per 60 Hz tick; AnalyserTrackController.h:22-23;
        # MainComponent.cpp:170): messages formed on the device (fx_get_osc_datagrams), handed to the kernel by the batch sender
        # (fx_osc_sender: sender threads, sendmmsg, optionally segmented sends) and counted by a local receiver.  Per channel count:
        # (1) what it costs to have the datagrams on the host, device-formed against vectors + host encoder; (2) the most the sender hands
        # to the kernel back to back; (3) two seconds of the 60 Hz timer with a fresh publication every tick: ticks late, datagrams lost.
        capi = fx.capi
        cores = usable_cores()
        res = {"loopback": "127.0.0.1, sender and receiver in this process, %d usable cores" % cores,
               "bar": "60 Hz x channels datagrams/s handed to the kernel and received, no tick late (8192 channels: 4.9e5 /s; 65 536: 3.9e6 /s)"}
        variants = [("segmented_sends_gro_receiver", True, True), ("segmented_sends", True, False), ("sendmmsg", False, False)]
        for C in (1024, 8192, 65536):
            an_o = fx.BatchAnalyser(C, N, device=dev)
            g = torch.Generator(device="cuda:%d" % dev).manual_seed(C)
            hop = 0.25 * torch.randn((C, 2, N // 2), generator=g, device="cuda:%d" % dev, dtype=torch.float32)
            an_o.push_hops(hop, want_raw=False, want_smoothed=False)
            an_o.sync()
            rec = {}
            reps = 20
            d, n = an_o.osc_datagrams("/Audio/A", 0)
            t0 = time.perf_counter()
            for _ in range(reps):
                d, n = an_o.osc_datagrams("/Audio/A", 0)
            t_dev = (time.perf_counter() - t0) / reps
            t0 = time.perf_counter()
            for _ in range(reps):
                d2, n2 = capi.osc_encode_batch("/Audio/A", 0, an_o.get_features())
            t_host = (time.perf_counter() - t0) / reps
            rec["datagrams_on_host_ms"] = {"device_formed": t_dev * 1e3, "vectors_then_host_encoder": t_host * 1e3, "identical": bool((d == d2).all() and (n == n2).all()),
                                           "bytes": int(d.nbytes)}
            threads = max(1, min(8, cores // 2))
            for name, gso, gro in variants:
                rx = capi.OscReceiver("127.0.0.1:0", threads=threads, prefix="/Audio/A", keep_channels=C, gro=gro)
                tx = capi.OscSender("127.0.0.1:%d" % rx.port, threads=threads, gso=gso)
                tx.update(d, n)
                # (2) back to back
                t0 = time.perf_counter()
                sent = 0
                while time.perf_counter() - t0 < 1.0:
                    sent += tx.send()
                dt = time.perf_counter() - t0
                time.sleep(0.3)
                got = rx.stats()["datagrams"]
                st0 = tx.stats()
                v = {"threads": threads, "back_to_back": {"handed_to_kernel_per_s": sent / dt, "received_share": got / max(sent, 1), "syscalls_per_tick": st0["syscalls"] / max(st0["ticks"], 1)}}
                # (3) the timer, a fresh publication per tick from the analysis side
                tx.start(60.0)
                t0 = time.perf_counter()
                pubs = 0
                while time.perf_counter() - t0 < 2.0:
                    tx.update(*an_o.osc_datagrams("/Audio/A", 0))
                    pubs += 1
                    time.sleep(max(0.0, t0 + pubs / 60.0 - time.perf_counter()))
                tx.stop()
                time.sleep(0.3)
                st = tx.stats()
                got1 = rx.stats()["datagrams"] - got
                ticks = st["ticks"] - st0["ticks"]
                sent1 = st["datagrams"] - st0["datagrams"]
                ok = bool(st["late_ticks"] == 0 and st["dropped"] == st0["dropped"] and ticks >= 110 and got1 >= 0.999 * sent1 and sent1 == ticks * C)
                sample_ok = all(rx.last(c) for c in (0, C // 2, C - 1))
                v["paced_60hz"] = {"seconds": 2.0, "ticks": ticks, "late_ticks": st["late_ticks"], "publications": pubs, "datagrams_per_s": sent1 / 2.0, "received_share": got1 / max(sent1, 1),
                                   "dropped_by_sender": st["dropped"] - st0["dropped"], "max_tick_ms": st["max_tick_ms"], "sustained": "yes" if ok else "no", "sample_channels_seen": bool(sample_ok)}
                rec[name] = v
                tx.close()
                rx.close()
            best = rec["segmented_sends_gro_receiver"]["back_to_back"]["handed_to_kernel_per_s"]
            rec["bottleneck"] = ("per-datagram work of the kernel's UDP path: one thread hands over %.2g datagrams/s with sendmmsg, %.2g with segmented sends to a receiver that "
                                 "splits them itself (loopback never cuts the buffer); a receiver that takes every datagram through the stack is the slower side"
                                 % (rec["sendmmsg"]["back_to_back"]["handed_to_kernel_per_s"] / threads, best / threads))
            res[str(C)] = rec
            an_o.close()
        return res

    def sustained():
        # >= 10 s of the headline step back to back with no host synchronisation between steps: HIP events on the context's stream
        # every few steps give frames/s per one-second window; clocks, power and temperature come from a CHILD process that polls the driver's
        # sysfs files (tools/smi_poll.py -- it never touches HIP).  The burst figure of the headline line is a 20-step region; this is the state
        # a node under load is in.
        import subprocess
        seconds = float(args.sustained_seconds)
        lib_stream = an._torch_stream(torch.device("cuda", dev))
        poll = None
        samples = []
        try:
            pr = torch.cuda.get_device_properties(dev)
            where = ["--pci", "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)] if hasattr(pr, "pci_bus_id") else ["--device", str(dev)]
            poll = subprocess.Popen([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "smi_poll.py"), "--interval", "0.2"] + where,
                                    stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        except Exception:
            poll = None
        every, lag = 10, 8
        step_s = max(avg_launch_s, 1e-4)
        max_marks = int(seconds / (every * step_s) * 1.5) + 64
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(max_marks)]
        used = 0
        with torch.cuda.stream(lib_stream):
            for _ in range(5):
                an.process_frames(frames, out_raw=raw, out_smoothed=sm)
            an.sync()
            t_wall0 = time.time()
            t0 = time.perf_counter()
            marks[0].record()
            used = 1
            steps_done = 0
            while used < max_marks:
                for _ in range(every):
                    an.process_frames(frames, out_raw=raw, out_smoothed=sm)
                steps_done += every
                marks[used].record()
                used += 1
                # The host may run ahead of the device by `lag` rounds (~0.2 s of queued steps: the queue is never empty, the device never
                # waits for the host) and no further, so that the run ends on the DEVICE's clock.  The wait is for a mark that lies `lag`
                # rounds back, not for the step just issued.
                if used > lag:
                    marks[used - lag].synchronize()
                    if marks[0].elapsed_time(marks[used - lag]) >= seconds * 1e3:
                        break
            an.sync()
            wall = time.perf_counter() - t0
        t_wall1 = time.time()
        if poll is not None:
            poll.terminate()
            try:
                text, _ = poll.communicate(timeout=5)
            except Exception:
                poll.kill()
                text = ""
            for line in text.splitlines():
                try:
                    rec = json.loads(line)
                except ValueError:
                    continue
                if "t" in rec and t_wall0 <= rec["t"] <= t_wall1:
                    samples.append(rec)
                elif "source" in rec:
                    smi_source = rec
        at = [marks[0].elapsed_time(marks[k]) / 1e3 for k in range(used)]            # seconds since the first mark, at the end of every `every` steps
        per_step_frames = count * T
        total_s = at[-1]
        windows = []
        k0 = 0
        for w in range(int(total_s)):
            k1 = max(k for k in range(used) if at[k] <= w + 1.0)
            if k1 > k0:
                windows.append((k1 - k0) * every * per_step_frames / (at[k1] - at[k0]))
            k0 = k1
        windows_sorted = sorted(windows)
        value = steps_done * per_step_frames / total_s
        burst = out.get("value") or value
        first_ms = 1e3 * (at[min(used - 1, max(1, int(1.0 / (every * step_s))))] / (min(used - 1, max(1, int(1.0 / (every * step_s)))) * every))
        last_k = max(0, used - 1 - max(1, int(1.0 / (every * step_s))))
        last_ms = 1e3 * (at[-1] - at[last_k]) / ((used - 1 - last_k) * every)

        def stat(key):
            xs = [s_[key] for s_ in samples if key in s_]
            return {"min": min(xs), "median": sorted(xs)[len(xs) // 2], "max": max(xs), "samples": len(xs)} if xs else None

        sclk = stat("sclk_mhz_hwmon") or stat("sclk_mhz")
        rec = {"value": value, "unit": "frames/s", "seconds": total_s, "steps": steps_done, "host_wall_s": wall,
               "workload": out.get("config", {}).get("workload"),
               "per_second_windows": {"min": windows_sorted[0], "median": windows_sorted[len(windows_sorted) // 2], "max": windows_sorted[-1], "n": len(windows)} if windows else None,
               "frac_of_burst": value / burst, "burst_value": burst,
               "step_ms_first_second": first_ms, "step_ms_last_second": last_ms,
               "sclk_mhz": sclk, "sclk_mhz_min": sclk["min"] if sclk else None, "mclk_mhz": stat("mclk_mhz"), "power_w": stat("power_w"), "temp_c": stat("temp_c"), "busy_pct": stat("busy_pct"),
               "smi_source": locals().get("smi_source"),
               "note": "the device queue is never empty (the host waits only for a mark 80 steps back); time from HIP events on the context's stream every %d steps; clocks / power / temperature polled by a child process "
                       "(tools/smi_poll.py: amdgpu sysfs, never HIP)" % every}
        # the roofline fractions at the sustained rate, beside the burst ones
        if out.get("roofline") and out["roofline"].get("frac") is not None:
            rec["roofline_frac_sustained"] = out["roofline"]["frac"] * (avg_launch_s * 1e3) / last_ms if last_ms > 0 else None
        return rec

    def live_soak():
        # The live path end to end under a clock, as a CHILD process (tools/live_soak.cpp, built by __graft_entry__.build(); nothing is exec'ed
        # from here): 8192 channels x 1024 points, 480-sample device blocks at 48 kHz through fx::LiveAnalyser (page-locked FIFO, the worker
        # analyses each block in place; the one-frame kernels read the block themselves), messages formed on the GPU, fx::OSCBatchSender's 60 Hz
        # timer to a local receiver.  Eight seconds after two of warm-up; profiles/r06_soak.txt has the minutes-long runs.
        import subprocess
        exe = os.path.join(ROOT, "tools", "_bin", "live_soak")
        if not os.path.exists(exe):
            return {"error": "tools/_bin/live_soak not built (__graft_entry__.build())"}
        p = subprocess.run([exe, "channels=8192", "window=%d" % N, "block=480", "seconds=8", "sender_threads=4"], capture_output=True, text=True, timeout=180)
        line = [l for l in p.stdout.splitlines() if l.startswith('{"live_soak"')]
        if not line:
            return {"error": "live_soak gave no record (rc %d): %s" % (p.returncode, (p.stdout + p.stderr)[-300:])}
        rec = json.loads(line[-1])["live_soak"]
        rec["note"] = ("latency = block arrival on the audio thread -> vectors published and every channel's OSC message formed; the 60 Hz timer then sends whatever was published "
                       "last (0 .. 16.7 ms later, as the reference's timer does); real_time_factor = wall time / the worker's busy time")
        return rec

    table = [("spectral_only", spectral_only), ("data_dependence", data_dependence), ("other_windows", other_windows), ("live_cadence", live_cadence),
             ("device_blocks", device_blocks), ("streaming_hop", streaming_hop), ("stream_ingest", stream_ingest), ("offline", offline),
             ("osc_sink", osc_sink), ("sustained", sustained), ("live_soak", live_soak)]
    only = set(args.only_extra.split(",")) if getattr(args, "only_extra", None) else None
    for name, fn in table:
        if only is None or name in only:
            guarded(name, fn)


if __name__ == "__main__":
    main()
