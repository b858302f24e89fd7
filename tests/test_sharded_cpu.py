"""N>1 path on CPU: two processes, gloo, world_size 2.  The sharding/gather code is the one bench.py
uses with RCCL; here each rank's local analysis is done by the CPU oracle (tests may use it) so the
check is: sharded + gathered result == unsharded result, bitwise, and the sink encodes per-channel OSC."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total_channels, T, N, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fx = importlib.import_module("feature-extractor_amd")
    sharded = importlib.import_module("feature-extractor_amd.sharded")
    from oracle import fx_oracle as fo
    first, count = sharded.my_shard(total_channels, rank, world)
    hops = fx.synth.hops(count, T, N, first_channel=first)
    raw, sm = fo.push_hops(hops, N)
    res, work = sharded.gather_features(torch.from_numpy(sm), total_channels, dst=0, async_op=True)
    if work is not None:
        work.wait()
    if rank == 0:
        np.save(out_path, sharded.resolve(res).numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total_channels", [6, 7])
def test_two_rank_gather_equals_unsharded(tmp_path, fx, oracle, total_channels):
    T, N = 6, 1024
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, _free_port(), total_channels, T, N, out), nprocs=2, join=True)
    got = np.load(out)
    _, want = oracle.push_hops(fx.synth.hops(total_channels, T, N), N)
    assert got.shape == (total_channels, T, 12)
    assert np.array_equal(got, want, equal_nan=True)


def test_shard_bounds_cover_all_channels():
    sharded = importlib.import_module("feature-extractor_amd.sharded")
    for C in (1, 7, 8, 1024, 65536, 65537):
        for G in (1, 2, 3, 8):
            b = sharded.shard_bounds(C, G)
            assert sum(n for _, n in b) == C
            pos = 0
            for first, n in b:
                assert first == pos or n == 0
                pos += n
    assert sharded.shard_bounds(65536, 8)[3] == (3 * 8192, 8192)        # BASELINE config 4: 8192 channels per GPU


def test_sink_encodes_one_datagram_per_channel(fx, oracle):
    sharded = importlib.import_module("feature-extractor_amd.sharded")
    sm = np.random.default_rng(0).standard_normal((12, 12)).astype(np.float32)
    sink = sharded.OscSink(fx.osc_encode)
    msgs = sink.datagrams(sm)
    assert len(msgs) == 12 and len(msgs[0]) == 76 and len(msgs[11]) == 76
    assert msgs[10] == oracle.osc_message("/Audio/A10", sm[10])
    sink.close()


def test_osc_target_parsing_follows_the_reference():
    sharded = importlib.import_module("feature-extractor_amd.sharded")
    assert sharded.parse_osc_target("127.0.0.1") == ("127.0.0.1", 9000)        # default port, ref OSCFeatureAnalysisOutput.h:117
    assert sharded.parse_osc_target("10.0.0.7:8001") == ("10.0.0.7", 8001)


def test_sink_sends_udp_datagrams_at_60hz_to_both_targets(fx, oracle):
    """Loopback UDP: primary + secondary receiver each get byte-exact 76-byte messages, paced ~60 Hz."""
    import time
    sharded = importlib.import_module("feature-extractor_amd.sharded")
    rx = []
    for _ in range(2):
        r = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        r.bind(("127.0.0.1", 0))
        r.settimeout(2.0)
        rx.append(r)
    sm = np.random.default_rng(1).standard_normal((3, 12)).astype(np.float32)
    sink = sharded.OscSink(fx.osc_encode, "127.0.0.1:%d" % rx[0].getsockname()[1], "127.0.0.1:%d" % rx[1].getsockname()[1])
    sink.update(sm)
    t0 = time.perf_counter()
    sink.start()
    got = [[rx[k].recv(256) for _ in range(3 * 12)] for k in range(2)]        # 12 ticks x 3 channels
    dt = time.perf_counter() - t0
    sink.close()
    want = [oracle.osc_message("/Audio/A%d" % c, sm[c]) for c in range(3)]
    for k in range(2):
        assert got[k][:3] == want and all(len(m) == 76 for m in got[k])
    assert 0.12 < dt < 1.0                         # 11 periods of 1/60 s = 0.18 s, generous bounds for CI
    for r in rx:
        r.close()


# ---- bench.py's own rank path (rank_main) under world_size 2, gloo, CPU: the oracle stands in for the GPU analyser ----
class _OracleAnalyser:
    """The slice of BatchAnalyser's interface bench.rank_main uses, backed by the CPU oracle (test infrastructure)."""

    def __init__(self, count, window):
        from oracle import fx_oracle as fo
        self.ch = [fo.Channel(window) for _ in range(count)]
        self.latest = np.zeros((count, 12), np.float32)
        self.calls = 0

    def process_frames(self, frames, out_raw=None, out_smoothed=None):
        for c, ch in enumerate(self.ch):
            raw, sm = ch.process_frames(frames[c].numpy())
            out_raw[c] = torch.from_numpy(raw)
            out_smoothed[c] = torch.from_numpy(sm)
            self.latest[c] = sm[-1]
        self.calls += 1

    def get_features(self):
        return self.latest.copy()

    def sync(self):
        pass

    def profile_begin(self):
        self.calls = 0

    def profile_end(self):
        return 1.0 * self.calls, 0.1 * self.calls, self.calls

    def reset_state(self):
        [ch.reset() for ch in self.ch]

    def close(self):
        pass


class _OracleEngine:
    name = "cpu-oracle"
    device = None
    rccl_capable = False

    def frames(self, host):
        return torch.from_numpy(host)

    def empty(self, shape):
        return torch.zeros(shape, dtype=torch.float32)

    def analyser(self, count, window, **kw):
        return _OracleAnalyser(count, window)

    def synchronize(self):
        pass


def _bench_rank(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"] = str(rank)
    import bench
    args = bench.build_parser().parse_args(["--gpus", str(world), "--backend", "gloo", "--channels-per-gpu", "3", "--frames", "4",
                                            "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra", "--rank-timeout", "120"])
    out = bench.rank_main(args, _OracleEngine(), rank, world)
    if rank == 0:
        import json
        json.dump(out, open(out_path, "w"))
    else:
        assert out is None


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_rank_path_two_ranks_gloo_cpu(tmp_path, world):
    """bench.py's N>1 entry (rank_main: shard plan, control group, per-step gather to the sink, the max over ranks, the
    sink's check of its block, one line from rank 0) with world_size 2, 4 and 8 -- the driver's 1/2/4/8 ladder -- on CPU: a
    stand-in engine analyses each rank's shard with the oracle, everything else is the code `bench.py --gpus N` runs.  A wrong
    shard offset or gather order fails the run's own collective check (exit 3)."""
    import json
    out = str(tmp_path / "line.json")
    mp.spawn(_bench_rank, args=(world, _free_port(), out), nprocs=world, join=True)
    d = json.load(open(out))
    assert d["n_gpus"] == world and d["scaling"] == "weak" and d["config"]["total_channels"] == 3 * world and d["config"]["channels_per_gpu"] == 3
    assert [r["rank"] for r in d["per_rank"]] == list(range(world))
    assert d["value"] > 0 and d["steps"] == 2 and "gloo" in d["config"]["sharding"]
    assert d["roofline"]["bound"] == "valu" and d["roofline"]["compute"]["flops_per_frame"] > 2e5


def test_first_multigpu_report_shape_from_a_two_rank_gloo_run(tmp_path):
    """tools/first_multigpu_run.py (the one-command kit for the day a multi-GPU node runs this): its report is assembled from bench
    lines with per-rank records.  Dry run on CPU: a real two-rank gloo line from bench.rank_main (oracle stand-in engine) plus a
    made-up one-rank line go through assemble(); the JSON has every rank's rate and kernel times, the scaling table and findings."""
    import importlib.util
    import json
    out = str(tmp_path / "line.json")
    mp.spawn(_bench_rank, args=(2, _free_port(), out), nprocs=2, join=True)
    two = json.load(open(out))
    assert [r["rank"] for r in two["per_rank"]] == [0, 1]
    for r in two["per_rank"]:
        assert r["channels"] == 3 and r["frames_per_s"] > 0 and r["frame_kernel_ms_per_step"] > 0 and "seconds" in r
    spec = importlib.util.spec_from_file_location("first_multigpu_run", os.path.join(ROOT, "tools", "first_multigpu_run.py"))
    kit = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kit)
    one = dict(two, n_gpus=1, value=two["value"] / 1.9, per_rank=None)
    rep = kit.assemble(2, {2: {"rc": 0, "skipped": False, "tail": "ok"}}, {1: one, 2: two})
    assert set(rep) >= {"visible_gpus", "comm_ranks", "bench", "scaling", "findings"}
    assert abs(rep["scaling"]["2"]["speedup_vs_1"] - 1.9) < 1e-9 and abs(rep["scaling"]["2"]["efficiency"] - 0.95) < 1e-9
    assert rep["bench"]["2"]["per_rank"][1]["rank"] == 1 and isinstance(rep["findings"], list) and rep["findings"]
    # what the findings are for: a slow rank, a communicator of the wrong size, a gather that takes a third of the step
    bad = json.loads(json.dumps(two))
    bad["per_rank"][1]["frames_per_s"] *= 0.5
    bad["per_rank"][0]["exchange"] = {"rccl_ranks": 1, "gathers": 3, "gathers_timed": 3, "gather_ms_total": 3.0, "gather_ms_max": 1.5,
                                      "gather_ms_mean": 0.4 * bad["ms_per_step"]}
    text = " ".join(kit.findings(2, bad))
    assert "slower than the fastest rank" in text and "RCCL counts 1 ranks" in text and "may no longer hide" in text
    json.dumps(rep)
