#!/usr/bin/env python3
"""bench.py -- frames/sec of the full 12-feature bundle on synthetic audio, per BASELINE.json.

Workload at N=1 GPU: BASELINE.json configs[1] -- 1024 synthetic 48 kHz channels x 1024-pt fp32
frames (50 % overlap, pre-assembled [C][T][1024], resident in HBM), the whole RealTimeAnalyser
bundle (spectral + pitch + harmonic + RMS + smoothing + onset).  A "step" is one pass of the hot
path over one batch of T consecutive frames per channel.  With N GPUs every rank analyses its own
1024-channel shard (weak scaling; channels are independent, so there is no data-path collective)
and the smoothed feature vectors are gathered to rank 0, the OSC sink, over RCCL.

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--channels-per-gpu", type=int, default=1024)
    ap.add_argument("--frames", type=int, default=512, help="consecutive frames per channel per step (SURVEY 8d: T >= 64)")
    ap.add_argument("--window", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the spectral-only extra measurement")
    ap.add_argument("--signal", default="synth", choices=["synth", "noise", "silence"],
                    help="synth = the BASELINE synthetic mix (default); noise / silence probe data-dependent paths")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: debugging aid for boxes with fewer GPUs than ranks (ranks share devices, features are gathered through host memory)")
    ap.add_argument("--stream", action="store_true",
                    help="host-resident hops through the pinned ring (fx_stream_*): PCIe-inclusive rate, reported as an extra line")
    ap.add_argument("--fp16", action="store_true", help="--stream only: fp16 samples")
    ap.add_argument("--debug-collective", action="store_true",
                    help="with --gpus 1: create a one-rank RCCL group and run the N>1 code path (gather included), "
                         "then check the gathered block against the local one")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    collective = world > 1 or args.debug_collective
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    fx = importlib.import_module("feature-extractor_amd")
    sharded = importlib.import_module("feature-extractor_amd.sharded")
    C, T, N = args.channels_per_gpu, args.frames, args.window
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
    an = fx.BatchAnalyser(count, N, device=local_rank)
    raw = torch.empty((count, T, 12), dtype=torch.float32, device=frames.device)
    sm = torch.empty((count, T, 12), dtype=torch.float32, device=frames.device)
    # What travels between GPUs is what the OSC sink samples (ref OSCFeatureAnalysisOutput.h:89-113): the latest
    # smoothed vector of every channel, [C][12] per rank per step.  Two buffers: the gather of step i
    # (RCCL, its own stream) reads one while step i+1 fills the other.
    latest_bufs = [torch.empty((count, 12), dtype=torch.float32, device=frames.device) for _ in range(2)]

    def barrier():
        an.sync()
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
            torch.cuda.synchronize()

    pending = [None, None]
    counter = [0]
    last = [None]
    # RCCL path: everything is ordered on the device.  `lib_stream` wraps the library's hipStream_t, the gather runs
    # on `side`; events make the gather wait for the step's kernels and make the kernels that next overwrite a
    # feature buffer wait for the gather that read it.  The host never blocks inside the timed loop.
    device_ordered = collective and args.backend == "nccl"
    if device_ordered:
        lib_stream = torch.cuda.ExternalStream(an.stream(), device=frames.device)
        side = torch.cuda.Stream(device=frames.device)

    def drain(slot):
        if pending[slot] is None:
            return
        work = pending[slot][1]
        if device_ordered:
            with torch.cuda.stream(side):
                if work is not None:
                    work.wait()                          # `side` waits for the collective
            lib_stream.wait_stream(side)
        elif work is not None:
            work.wait()
        pending[slot] = None

    def step():
        slot = counter[0] & 1
        counter[0] += 1
        if collective:
            drain(slot)                                  # the gather that last read this buffer is ordered before us
        an.process_frames(frames, out_raw=raw, out_smoothed=sm)
        if collective:
            an.get_features(out=latest_bufs[slot])       # async device copy on the library's stream
        if device_ordered:
            side.wait_stream(lib_stream)                 # features ready before RCCL reads them
            with torch.cuda.stream(side):
                pending[slot] = sharded.gather_features(latest_bufs[slot], total_channels, dst=0, async_op=True, single_rank_collective=True)
            last[0] = (slot, pending[slot][0])
        elif collective:
            an.sync()
            pending[slot] = sharded.gather_features(latest_bufs[slot].cpu(), total_channels, dst=0, async_op=True, single_rank_collective=True)
            last[0] = (slot, pending[slot][0])

    if collective:
        # one untimed exchange so that the RCCL communicator and the gather buffers exist even with --warmup 0
        step()
        drain(0)
        drain(1)
    for _ in range(args.warmup):
        step()
    drain(0)
    drain(1)
    barrier()
    an.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain(0)
    drain(1)
    barrier()
    dt = time.perf_counter() - t0
    frame_ms, epi_ms, calls = an.profile_end()

    t = torch.tensor([dt], dtype=torch.float64, device=frames.device)
    if collective:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    if args.debug_collective and rank == 0:
        slot, gathered = last[0]
        got = sharded.resolve(gathered)[first:first + count].cpu()
        want = latest_bufs[slot].cpu()
        if not torch.equal(torch.nan_to_num(want), torch.nan_to_num(sm[:, -1, :].cpu())):
            raise SystemExit("latest vectors differ from the last frame's smoothed vectors")
        same = bool(torch.equal(torch.nan_to_num(got), torch.nan_to_num(want)))
        print("debug-collective: gathered block %s the local features" % ("equals" if same else "DIFFERS FROM"), file=sys.stderr, flush=True)
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
        out = {
            "metric": "frames/sec (1024-pt FFT, 10-feature bundle)" if N == 1024 else "frames/sec (%d-pt FFT, 10-feature bundle)" % N,
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: %d channels/GPU x %d-pt fp32 frames, %d consecutive frames per channel per step, "
                                   "full 12-feature RealTimeAnalyser bundle (spectral+pitch+harmonic+RMS+smoothing+onset), "
                                   "frames pre-assembled and resident in HBM" % (C, N, T),
                       "channels_per_gpu": C, "frames_per_step": T, "window": N, "sample_rate": 48000,
                       "sharding": "channels, contiguous blocks; per step an RCCL gather of the latest smoothed vectors [C][12] of every rank to rank 0 (the OSC sink)" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": load_traffic(N, count, T),
                         "kernel": "fx_frame_kernel<%d>" % N, "avg_launch_ms": avg_launch_s * 1e3,
                         "algorithmic_bytes_per_launch": launch_bytes, "launches_timed": calls,
                         "epilogue_ms_per_step": epi_ms / max(calls, 1),
                         "note": "algorithmic bytes = (4*N + 48) B/frame x frames per launch; this kernel is VALU/LDS-bound, not HBM-bound (see DESIGN.md)"},
        }
        if world == 1 and not args.no_extra:
            # BASELINE configs[1] read literally is the spectral analyser alone ("fused window + FFT +
            # magnitude + SpectralCharacteristics reductions in one kernel"); the same workload with only the
            # RealTimeSpectralAnalyser constructed, as an extra (never `value`)
            an_s = fx.BatchAnalyser(count, N, device=local_rank, analysers="spectral")
            for _ in range(3):
                an_s.process_frames(frames, out_raw=raw, out_smoothed=sm)
            an_s.sync()
            an_s.profile_begin()
            t1 = time.perf_counter()
            for _ in range(10):
                an_s.process_frames(frames, out_raw=raw, out_smoothed=sm)
            an_s.sync()
            dts = time.perf_counter() - t1
            fms, _, calls_s = an_s.profile_end()
            ach = launch_bytes / (fms / 1e3 / max(calls_s, 1)) / 1e9
            out["spectral_only"] = {"value": total_channels * T * 10 / dts, "unit": "frames/s",
                                    "kernel": "fx_frame_kernel<%d, spectral>" % N, "avg_launch_ms": fms / max(calls_s, 1),
                                    "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS},
                                    "note": "FX_SPECTRAL_ONLY: RMS, centroid, spread, flatness, LER, flux, slope, onset (8 of the 12 slots)"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(fx, N, T)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    if collective:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
