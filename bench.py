#!/usr/bin/env python3
"""bench.py -- frames/sec of the full 12-feature bundle on synthetic audio, per BASELINE.json.

Workload at N=1 GPU: BASELINE.json configs[1] -- 1024 synthetic 48 kHz channels x 1024-pt fp32
frames (50 % overlap, pre-assembled [C][T][1024], resident in HBM), the whole RealTimeAnalyser
bundle (spectral + pitch + harmonic + RMS + smoothing + onset).  A "step" is one pass of the hot
path over one batch of T consecutive frames per channel.  With N > 1 GPUs the workload is BASELINE.json
configs[3]: every rank analyses its own contiguous 8192-channel shard of 8192*N channels (weak scaling;
channels are independent, so there is no data-path collective) and the smoothed feature vectors are
gathered to rank 0, the OSC sink, over RCCL (through the C ABI: fx_comm_* / fx_gather_smoothed).

Launch forms (both work):
    python bench.py --gpus N ...                                   (bench.py starts its N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N ...                    (the driver's form)

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


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


def load_traffic(window, channels, frames):
    """HBM bytes per frame-kernel launch from committed PMC passes (profiles/pmc_traffic.json,
    written by tools/pmc_traffic.py from rocprofv3 --pmc runs of this same command)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(path))
        if rec.get("window") == window and rec.get("channels") == channels and rec.get("frames") == frames:
            return rec.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def stream_bench(fx, args, C, T, N, device):
    """End-to-end streaming: hops live in host memory, go through the pinned ring with H2D copies on a
    side stream overlapped with analysis, results come back to the host.  PCIe-inclusive."""
    dtype = np.float16 if args.fp16 else np.float32
    hops = fx.synth.hops(C, T, N).astype(dtype)
    an = fx.BatchAnalyser(C, N, device=device)
    st = fx.HopStream(an, T, slots=3, dtype=dtype)

    def step():
        if st.in_flight() == 3:
            st.collect(want_raw=False)
        st.slot()[...] = hops                       # the producer's copy into pinned memory is part of the path
        st.submit()

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


def self_launch(args):
    """`python bench.py --gpus N` from a bare shell: this process has made no GPU call; it starts the N ranks
    as fresh children (torch.distributed.run, one process per GPU), lets rank 0's JSON line through on stdout
    and exits with their status.  Never an exec of a GPU-initialised process."""
    import subprocess
    import torch
    if args.backend == "nccl" and torch.cuda.device_count() < args.gpus:
        raise SystemExit("--gpus %d but only %d GPU(s) visible; RCCL needs one GPU per rank "
                         "(use --backend gloo to let ranks share devices for debugging)" % (args.gpus, torch.cuda.device_count()))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def valu_model(window):
    """VALU instructions per frame and their mean issue cost, from committed rocprofv3 / microbenchmark records
    (profiles/valu_model.json, written by tools/valu_model.py); None if there is no record for this window."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "valu_model.json")))
        return rec.get(str(window))
    except Exception:
        return None


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--channels-per-gpu", type=int, default=None,
                    help="default: 1024 at --gpus 1 (BASELINE configs[1]), 8192 at --gpus N>1 (configs[3])")
    ap.add_argument("--frames", type=int, default=None,
                    help="consecutive frames per channel per step (SURVEY 8d: T >= 64); default 512 at --gpus 1, 128 at --gpus N>1 "
                         "(8192 channels x 128 frames = 4.3 GB of input per GPU; 1.81e8 frames/s per GPU, as the single-GPU line)")
    ap.add_argument("--window", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra measurements (spectral-only, noise / silence, other windows)")
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
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.backend == "gloo":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    collective = world > 1 or args.debug_collective
    if collective:
        # control plane (barriers, the communicator id, the max over ranks): a CPU group.  The data-path exchange is
        # RCCL inside the library.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    fx = importlib.import_module("feature-extractor_amd")
    sharded = importlib.import_module("feature-extractor_amd.sharded")
    N = args.window
    C = args.channels_per_gpu if args.channels_per_gpu is not None else (1024 if world == 1 else 8192)
    T = args.frames if args.frames is not None else (512 if world == 1 else 128)
    if args.stream:
        return stream_bench(fx, args, C, T, N, local_rank)
    total_channels = C * world
    first, count = sharded.my_shard(total_channels, rank, world)

    host_frames = fx.synth.frames(count, T, N, first_channel=first)
    if args.signal == "noise":
        host_frames = np.random.default_rng(first).normal(0, 0.1, host_frames.shape).astype(np.float32)
    elif args.signal == "silence":
        host_frames = np.zeros_like(host_frames)
    frames = torch.from_numpy(host_frames).cuda(local_rank)
    del host_frames
    an = fx.BatchAnalyser(count, N, device=local_rank)
    raw = torch.empty((count, T, 12), dtype=torch.float32, device=frames.device)
    sm = torch.empty((count, T, 12), dtype=torch.float32, device=frames.device)

    # What travels between GPUs is what the OSC sink samples (ref OSCFeatureAnalysisOutput.h:89-113): the latest
    # smoothed vector of every channel, [C][12] per rank per step, gathered to rank 0.
    rccl = collective and args.backend == "nccl"
    gathered = None
    if rccl:
        ident = [fx.BatchAnalyser.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0)
        an.comm_create(rank, world, ident[0])
        total, firsts = an.comm_layout()
        assert total == total_channels and firsts[rank] == first, (total, firsts, first)
        if rank == 0:      # two destination buffers: a consumer may read one while the next gather fills the other
            gathered = [torch.empty((total_channels, 12), dtype=torch.float32, device=frames.device) for _ in range(2)]
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
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
            torch.cuda.synchronize()

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
        step()                  # one untimed exchange so that the communicator's channels exist even with --warmup 0
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

    if collective:
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
    if rank == 0:
        frames_total = total_channels * T * args.steps
        value = frames_total / dt
        bytes_per_frame = 4 * N + 48                     # sample bytes + 12 floats out
        launch_bytes = bytes_per_frame * count * T
        avg_launch_s = frame_ms / 1e3 / max(calls, 1)
        achieved = launch_bytes / avg_launch_s / 1e9
        cfg = "configs[1]" if world == 1 else "configs[3]"
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": load_traffic(N, count, T),
                "kernel": "fx_frame_kernel<%d>" % N, "avg_launch_ms": avg_launch_s * 1e3,
                "algorithmic_bytes_per_launch": launch_bytes, "launches_timed": calls,
                "epilogue_ms_per_step": epi_ms / max(calls, 1),
                "limiter": "valu",
                "note": "algorithmic bytes = (4*N + 48) B/frame x frames per launch, against the HBM peak as SURVEY 8(d) defines the "
                        "roofline; the kernel's binding unit is VALU issue (see valu_issue_frac and DESIGN.md 3.3)"}
        vm = valu_model(N)
        if vm:
            # time the kernel's VALU instructions need at the measured per-class issue costs (tools/ubench), as a fraction
            # of the kernel's duration: the binding unit's utilisation
            simds = 1024
            need_s = vm["valu_per_frame"] * vm["mean_issue_ns"] * 1e-9 * count * T / simds
            roof["valu_issue_frac"] = need_s / avg_launch_s
            roof["valu_insts_per_frame"] = vm["valu_per_frame"]
            roof["valu_model_source"] = vm.get("source")
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
        if world == 1 and not args.no_extra and not args.debug_collective:
            extra_steps = 10
            # BASELINE configs[1] read literally is the spectral analyser alone ("fused window + FFT +
            # magnitude + SpectralCharacteristics reductions in one kernel"); the same workload with only the
            # RealTimeSpectralAnalyser constructed, as an extra (never `value`)
            an_s = fx.BatchAnalyser(count, N, device=local_rank, analysers="spectral")
            fps, fms = time_steps(an_s, frames, raw, sm, extra_steps, warmup=3)
            an_s.close()
            ach = launch_bytes / (fms / 1e3) / 1e9
            out["spectral_only"] = {"value": fps, "unit": "frames/s",
                                    "kernel": "fx_frame_kernel<%d, spectral>" % N, "avg_launch_ms": fms,
                                    "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS},
                                    "note": "FX_SPECTRAL_ONLY: RMS, centroid, spread, flatness, LER, flux, slope, onset (8 of the 12 slots)"}
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
                dd[name] = {"value": fps, "unit": "frames/s", "frame_kernel_ms": fms,
                            "relative_to_synth": (avg_launch_s * 1e3) / fms}
                del other
            dd["note"] = "same shape as the headline run; noise = N(0, 0.1^2) white, silence = zeros; relative_to_synth = synth kernel time / this kernel time"
            out["data_dependence"] = dd
            an.reset_state()
            # the reference application's default window (AnalyserTrackController.h:20-21) and the streaming config's window
            others = {}
            for (n2, c2, t2, label) in ((2048, 4096, 64, "configs[2] shape: 4096 channels x 2048-pt"), (4096, 1024, 64, "configs[4] window: 1024 channels x 4096-pt")):
                if n2 == N:
                    continue
                fr2 = torch.from_numpy(fx.synth.frames(c2, t2, n2)).cuda(local_rank)
                an2 = fx.BatchAnalyser(c2, n2, device=local_rank)
                # (best of two passes: the first launches on a fresh context run up to 8 % slow -- clocks, first touch of the
                # scratch buffers -- and this is an extra, not the timed region)
                fps, fms = max(time_steps(an2, fr2, None, None, 2 * extra_steps, warmup=5) for _ in range(2))
                an2.close()
                b2 = (4 * n2 + 48) * c2 * t2
                others[str(n2)] = {"value": fps, "unit": "frames/s", "frame_kernel_ms": fms, "workload": "%s, %d frames per step" % (label, t2),
                                   "hbm_frac": b2 / (fms / 1e3) / 1e9 / HBM_PEAK_GBPS}
                del fr2
            out["other_windows"] = others
            # BASELINE configs[4]: 1 channel, 4096-pt windows, fp16 samples, ONE hop per call through the pinned ring
            # (fx_hop_kernel: the whole step in one launch, the host polls a flag) -- per-hop round trip as this
            # interpreter sees it (tools/stream_latency.cpp measures the same from C++, a few microseconds less)
            try:
                an4 = fx.BatchAnalyser(1, 4096, device=local_rank)
                st4 = fx.HopStream(an4, 1, slots=3, dtype=np.float16)
                h4 = fx.synth.hops(1, 64, 4096, first_channel=24).astype(np.float16)
                n_calls = 2000
                for k in range(n_calls + 200):
                    if k == 200:
                        t_s = time.perf_counter()
                    st4.slot()[...] = h4[:, k % 64:k % 64 + 1]
                    st4.submit()
                    st4.collect(want_raw=False)
                us = (time.perf_counter() - t_s) / n_calls * 1e6
                st4.close()
                an4.close()
                out["streaming_hop"] = {"round_trip_us": us, "unit": "us per 2048-sample hop", "calls": n_calls,
                                        "workload": "configs[4]: 1 channel x 4096-pt windows, fp16 samples, one hop per call, submit + collect from Python"}
            except Exception as e:          # never let an extra take the headline line down
                out["streaming_hop"] = {"error": str(e)}
        if not args.no_cpu_baseline and world == 1 and not args.debug_collective:
            out["cpu_baseline"] = cpu_baseline(fx, N, T)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    if collective:
        dist.barrier()
        if rccl:
            an.comm_destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
