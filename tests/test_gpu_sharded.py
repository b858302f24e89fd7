"""The multi-GPU path on the GPU box: the BASELINE configs[3] shard shape (8192 channels x 1024-pt on one GPU), the
RCCL gather behind the C ABI (fx_comm_* / fx_gather_smoothed) on a one-rank communicator and -- when the box has
two GPUs -- between two processes, and bench.py's launch forms."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def test_configs3_shard_8192_channels(gpu_fx, oracle):
    """One rank's share of BASELINE configs[3]: 8192 channels x 1024-pt.  24 random channels against the oracle;
    the two 4096-channel halves analysed in their own contexts give the same bits (a channel's result does not
    depend on the shard it lands in, which is what makes contiguous-block sharding exact)."""
    import signals
    N, C, T = 1024, 8192, 16
    frames = gpu_fx.synth.frames(C, T, N)
    an = gpu_fx.BatchAnalyser(C, N)
    raw, sm = an.process_frames(frames)
    lo = gpu_fx.BatchAnalyser(C // 2, N).process_frames(frames[: C // 2])
    hi = gpu_fx.BatchAnalyser(C // 2, N).process_frames(frames[C // 2:])
    assert np.array_equal(np.concatenate([lo[0], hi[0]]), raw, equal_nan=True)
    assert np.array_equal(np.concatenate([lo[1], hi[1]]), sm, equal_nan=True)
    assert np.array_equal(an.get_features(), sm[:, -1], equal_nan=True)
    pick = np.random.default_rng(3).choice(C, 24, replace=False)
    oraw, osm = oracle.process_frames(frames[pick], N)
    signals.assert_features_close(raw[pick], oraw, 1e-5, oracle.FEATURE_NAMES, "8192-channel shard raw")
    signals.assert_features_close(sm[pick], osm, 1e-5, oracle.FEATURE_NAMES, "8192-channel shard smoothed")


def _device_signal(torch, C, samples, seed, dtype):
    """[C][samples] on the GPU, made there (the whole of configs[3] is gigabytes): three harmonics of a per-channel pitch + noise"""
    g = torch.Generator(device="cuda").manual_seed(seed)
    out = torch.empty((C, samples), dtype=dtype, device="cuda")
    t = torch.arange(samples, device="cuda", dtype=torch.float32)
    for c0 in range(0, C, 4096):
        c1 = min(C, c0 + 4096)
        f = 55.0 * 2.0 ** ((torch.arange(c0, c1, device="cuda") % 72).to(torch.float32) / 12.0)
        ph = (2.0 * np.pi / 48000.0) * f[:, None] * t[None, :]
        x = 0.4 * torch.sin(ph) + 0.2 * torch.sin(2 * ph) + 0.1 * torch.sin(3 * ph)
        x += 0.05 * (2.0 * torch.rand(x.shape, generator=g, device="cuda") - 1.0)
        out[c0:c1] = x.to(dtype)
    return out


def test_configs3_whole_65536_channels_in_one_context_equal_the_eight_shards(gpu_fx, oracle):
    """BASELINE configs[3] at full size on the one GPU there is: ONE context of 65 536 channels x 1024-pt x 16 hops (what eight ranks of
    8192 channels hold between them) gives, bit for bit, what eight contexts of 8192 channels give on their blocks -- a channel's result does
    not depend on the shard it lands in, at the sizes the 8-GPU run uses -- and 24 channels of it meet the oracle.  Then the index space
    beyond 2^32: 65 536 channels x 72 assembled fp16 windows of 1024 points = 4.8e9 samples in one call; and a device block of 700 samples
    per channel through fx_push_samples (the re-blocking kernel walks the channels in slices of 65 535).  xGMI is the part this cannot
    reach: the gather of the eight blocks stays unmeasured."""
    import signals
    import torch
    N, C, T, S = 1024, 65536, 16, 8192
    hops = _device_signal(torch, C, T * (N // 2), 11, torch.float32).reshape(C, T, N // 2)
    whole = gpu_fx.BatchAnalyser(C, N)
    raw, sm = whole.push_hops(hops)
    whole.sync()
    for r in range(C // S):
        shard = gpu_fx.BatchAnalyser(S, N)
        a, b = shard.push_hops(hops[r * S:(r + 1) * S].contiguous())
        shard.sync()
        assert torch.equal(a.view(torch.int32), raw[r * S:(r + 1) * S].view(torch.int32)), r
        assert torch.equal(b.view(torch.int32), sm[r * S:(r + 1) * S].view(torch.int32)), r
        assert np.array_equal(shard.get_features(), sm[r * S:(r + 1) * S, -1].cpu().numpy(), equal_nan=True)
        shard.close()
    pick = np.sort(np.concatenate([np.random.default_rng(5).choice(C, 21, replace=False), [0, C - 1, S]]))
    idx = torch.from_numpy(pick).cuda()
    oraw, osm = oracle.push_hops(hops[idx].cpu().numpy(), N)
    signals.assert_features_close(raw[idx].cpu().numpy(), oraw, 1e-5, oracle.FEATURE_NAMES, "65536-channel context raw")
    signals.assert_features_close(sm[idx].cpu().numpy(), osm, 1e-5, oracle.FEATURE_NAMES, "65536-channel context smoothed")
    # a device block of 700 samples per channel on top: 1 hop + 188 pending, the same bits as the hop itself
    more = _device_signal(torch, C, 700, 12, torch.float32)
    r2, s2 = whole.push_samples(more)
    whole.sync()
    assert r2.shape == (C, 1, 12) and whole.pending_samples() == 188
    ref = gpu_fx.BatchAnalyser(S, N)
    ref.push_hops(hops[C - S:].contiguous())
    a, b = ref.push_hops(more[C - S:, :512].contiguous().reshape(S, 1, 512))
    ref.sync()
    assert torch.equal(a.view(torch.int32), r2[C - S:].view(torch.int32)) and torch.equal(b.view(torch.int32), s2[C - S:].view(torch.int32))
    del hops, raw, sm, whole, ref, more
    torch.cuda.empty_cache()
    # more than 2^32 samples in one call: assembled fp16 windows, 65 536 x 72 x 1024
    T2 = 72
    frames = _device_signal(torch, C, T2 * N, 13, torch.float16).reshape(C, T2, N)
    assert frames.numel() > 2 ** 32
    big = gpu_fx.BatchAnalyser(C, N)
    raw, sm = big.process_frames(frames)
    big.sync()
    for r in (0, 3, 7):
        shard = gpu_fx.BatchAnalyser(S, N)
        a, b = shard.process_frames(frames[r * S:(r + 1) * S].contiguous())
        shard.sync()
        assert torch.equal(a.view(torch.int32), raw[r * S:(r + 1) * S].view(torch.int32)), r
        assert torch.equal(b.view(torch.int32), sm[r * S:(r + 1) * S].view(torch.int32)), r
        shard.close()
    tail = torch.tensor([C - 1, C - 2, C - 4097], device="cuda")
    oraw, osm = oracle.process_frames(frames[tail].float().cpu().numpy(), N)
    signals.assert_features_close(raw[tail].cpu().numpy(), oraw, 1e-5, oracle.FEATURE_NAMES, "windows beyond 2^32 samples raw")
    signals.assert_features_close(sm[tail].cpu().numpy(), osm, 1e-5, oracle.FEATURE_NAMES, "windows beyond 2^32 samples smoothed")


def test_one_rank_rccl_gather_through_the_c_abi(gpu_fx):
    """fx_comm_create / fx_gather_smoothed / fx_comm_sync on a one-rank communicator: host and device destinations,
    several gathers in flight behind analysis calls."""
    import torch
    N, C, T = 1024, 37, 11
    an = gpu_fx.BatchAnalyser(C, N)
    an.comm_create(0, 1, gpu_fx.BatchAnalyser.comm_unique_id())
    total, first = an.comm_layout()
    assert (total, first) == (C, [0])
    hops = gpu_fx.synth.hops(C, 3 * T, N)
    host_out = np.full((C, 12), -1.0, np.float32)
    dev_out = [torch.full((C, 12), -1.0, dtype=torch.float32, device="cuda:0") for _ in range(2)]
    for i in range(3):
        _, sm = an.push_hops(hops[:, i * T:(i + 1) * T])
        an.gather_features(dst=0, out=dev_out[i & 1])
        an.gather_features(dst=0, out=host_out)
        an.comm_sync()
        assert np.array_equal(host_out, sm[:, -1], equal_nan=True)
        assert np.array_equal(dev_out[i & 1].cpu().numpy(), sm[:, -1], equal_nan=True)
    with pytest.raises(gpu_fx.FxError):
        an.comm_create(0, 1, gpu_fx.BatchAnalyser.comm_unique_id())      # one communicator per context
    an.comm_destroy()
    with pytest.raises(gpu_fx.FxError):
        an.gather_features(dst=0, out=host_out)                          # no communicator any more


def _build_comm_ranks(fx, tmp_path):
    exe = str(tmp_path / "comm_ranks")
    lib_dir = os.path.dirname(fx.library_path())
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    # (the HIP runtime only for the sink's device-resident destination tables: hipMalloc / hipMemcpy / hipFree)
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(rocm, "include"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "comm_ranks.cpp"), "-o", exe,
                           "-L", lib_dir, "-lfx_hip", "-Wl,-rpath," + lib_dir, "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-ldl"])
    return exe


def test_cpp_one_rank_comm(gpu_fx, tmp_path):
    """The C++ host program (no Python, no torch in the process) on one rank."""
    out = subprocess.run([_build_comm_ranks(gpu_fx, tmp_path), "1"], capture_output=True, text=True, env=_env(), timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr


def test_cpp_two_rank_comm(gpu_fx, tmp_path):
    """Two processes, two GPUs, ragged shards; skipped on a one-GPU box."""
    out = subprocess.run([_build_comm_ranks(gpu_fx, tmp_path), "2", "shards=8192,37", "sinks=0,0,0,1,0"], capture_output=True, text=True, env=_env(), timeout=300)
    if out.returncode == 77:
        pytest.skip("one GPU visible")
    assert out.returncode == 0, out.stdout + out.stderr


def _bench(args, timeout=600):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=_env(), timeout=timeout)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line), out.stderr


def test_bench_debug_collective_runs_the_rccl_path_on_one_rank():
    d, err = _bench(["--gpus", "1", "--debug-collective", "--steps", "3", "--warmup", "1", "--frames", "32", "--no-cpu-baseline", "--no-extra"])
    assert d["n_gpus"] == 1 and "RCCL" in d["config"]["sharding"]
    assert "gathered block equals the local features" in err


def test_bench_four_ranks_of_configs3_shards_share_the_one_gpu():
    """`bench.py --gpus 4 --backend gloo --channels-per-gpu 8192 --frames 16`: the real shard plan of configs[3] -- 8192 channels x 1024-pt
    per rank -- as FOUR rank processes on the one GPU of this box (the GPU box's process guard allows six processes on a card, the test
    runner being one; eight ranks on one card is not something this pool lets a test do), the control group, the gather of every step to the
    sink through host memory, the sink's check of its own block, per-rank records.  What it cannot show is RCCL between GPUs."""
    d, err = _bench(["--gpus", "4", "--steps", "3", "--warmup", "1", "--backend", "gloo", "--channels-per-gpu", "8192", "--frames", "16"], timeout=1100)
    assert d["n_gpus"] == 4 and d["config"]["workload"].startswith("configs[3]") and d["config"]["total_channels"] == 4 * 8192
    assert d["value"] > 0 and d["scaling"] == "weak"
    assert len(d["per_rank"]) == 4 and all(r["frames_per_s"] > 0 for r in d["per_rank"])
    assert "gathered block equals the local features" in err


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` from a bare shell (no torchrun): rc 0 and n_gpus 2.  On a one-GPU box the two
    ranks share the device and exchange through host memory (--backend gloo); the shard is configs[3]'s."""
    import torch
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    d, err = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", backend, "--channels-per-gpu", "2048", "--frames", "16"])
    assert d["n_gpus"] == 2 and d["config"]["workload"].startswith("configs[3]") and d["config"]["total_channels"] == 4096
    assert d["cpu_baseline"] is None and d["value"] > 0
    assert "gathered block equals the local features" in err
