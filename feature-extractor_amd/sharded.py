"""Channel sharding across the GPUs of one node (one process per GPU, torch.distributed).

Channels are independent (one AnalyserTrackController each, ref AnalyserTrackController.h:199-210)
and frames within a channel are sequential, so the path shards by channel and never by time:
rank r owns the contiguous block [r*ceil(C/G), ...).  There is no data-path collective; the only
exchange is the gather of the 12-float feature vectors to the rank that owns the OSC sink
(ref OSCFeatureAnalysisOutput.h:89-113), which on GPUs is RCCL over xGMI.
"""
import numpy as np


def shard_bounds(num_channels, world_size):
    """[(first, count)] per rank: contiguous blocks, sizes differ by at most one block remainder."""
    per = -(-num_channels // world_size)
    out = []
    for r in range(world_size):
        first = min(r * per, num_channels)
        out.append((first, max(0, min(per, num_channels - first))))
    return out


def my_shard(num_channels, rank, world_size):
    return shard_bounds(num_channels, world_size)[rank]


def gather_features(local, num_channels, dst=0, group=None, async_op=False, single_rank_collective=False):
    """Gather per-rank feature blocks [C_local][...][12] (torch tensors, CPU for gloo / CUDA for
    RCCL) to rank `dst`.  Returns (result, work): on dst `result` is the [num_channels][...][12]
    tensor in channel order (valid once work has completed), elsewhere None.  Ranks may own
    different channel counts; blocks are padded to the largest for the collective."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    bounds = shard_bounds(num_channels, world)
    per = max(b[1] for b in bounds)
    if local.shape[0] != bounds[rank][1]:
        raise ValueError("rank %d owns %d channels, got %d" % (rank, bounds[rank][1], local.shape[0]))
    if local.shape[0] < per:
        pad = torch.zeros((per - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], 0)
    local = local.contiguous()
    if world == 1 and not single_rank_collective:     # (a one-rank collective is only useful to exercise the backend)
        return local[:num_channels], None
    if rank == dst:
        out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        parts = list(out.view((world, per) + tuple(local.shape[1:])).unbind(0))
        work = dist.gather(local, gather_list=parts, dst=dst, group=group, async_op=async_op)
        keep = np.concatenate([np.arange(r * per, r * per + bounds[r][1]) for r in range(world)])
        if len(keep) == world * per:
            return out, work
        return _LazySelect(out, torch.as_tensor(keep, device=out.device)), work
    work = dist.gather(local, gather_list=None, dst=dst, group=group, async_op=async_op)
    return None, work


class _LazySelect:
    """Index the padded gather buffer only after the collective has finished."""

    def __init__(self, buf, keep):
        self.buf, self.keep = buf, keep

    def resolve(self):
        return self.buf.index_select(0, self.keep)


def resolve(result):
    return result.resolve() if isinstance(result, _LazySelect) else result


def parse_osc_target(address, default_port=9000):
    """'ip[:port]' -> (ip, port), as OSCFeatureAnalysisOutput::connectToAddress parses it
    (ref OSCFeatureAnalysisOutput.h:115-123: text after the last ':' is the port, default 9000)."""
    address = address.strip()
    if ":" in address:
        host, _, port = address.rpartition(":")
        try:
            return host.split(":")[0], int(port)
        except ValueError:
            return host.split(":")[0], 0
    return address, default_port


class OscSink:
    """The sink side of the path (ref OSCFeatureAnalysisOutput.h:84-136): holds the latest smoothed
    vectors [C][12] and, paced by a 60 Hz timer (ref :133 startTimerHz (60)), sends one OSC message per
    channel -- address /Audio/A<channel> (ref MainComponent.cpp:170), 12 big-endian floats in the order of
    ref :107 -- to a primary and an optional secondary target (ref AnalyserTrackController.h:22-23).
    Sampling is asynchronous to frame production, as in the reference: a value may be sent twice.

    Scale-shaped since round 6: the messages of a tick are formed in ONE call (fx_osc_encode_batch on the host, or handed over
    ready-made by BatchAnalyser.osc_datagrams, which writes them on the GPU) and go to the kernel in sendmmsg batches from the
    library's sender threads (fx_osc_sender, csrc/fx_osc_sender.cpp), which also run the timer: no Python in a tick.
    `encode` (the per-message encoder of earlier rounds) is accepted and ignored."""

    def __init__(self, encode=None, target="127.0.0.1:9000", secondary=None, bundle_prefix="/Audio/A", rate_hz=60.0, first_channel=0, threads=1, gso=False):
        from . import capi
        self._capi = capi
        self.targets = [parse_osc_target(target)] + ([parse_osc_target(secondary)] if secondary else [])
        self.prefix = bundle_prefix
        self.first_channel = first_channel
        self.rate_hz = rate_hz
        self.sender = capi.OscSender(target, secondary, threads=threads, gso=gso)
        self._have = False

    def update(self, smoothed):
        """Publish the newest AudioFeatures::getValue vectors [C][12] (what the timer will sample)."""
        d, n = self._capi.osc_encode_batch(self.prefix, self.first_channel, np.asarray(smoothed, np.float32).reshape(-1, 12))
        self.update_datagrams(d, n)

    def update_datagrams(self, datagrams, lengths):
        """Publish messages that are already formed ([C][stride] bytes + lengths: BatchAnalyser.osc_datagrams)."""
        self.sender.update(datagrams, lengths)
        self._have = True

    def datagrams(self, smoothed):
        d, n = self._capi.osc_encode_batch(self.prefix, self.first_channel, np.asarray(smoothed, np.float32).reshape(-1, 12))
        return [bytes(d[c, :n[c]]) for c in range(d.shape[0])]

    def send(self, smoothed=None):
        """One timer tick now: sendSpectralFeaturesViaOSC for every channel (ref :89-113).  Returns the datagrams handed to the kernel."""
        if smoothed is not None:
            self.update(smoothed)
        return self.sender.send() if self._have else 0

    @property
    def sent(self):
        return self.sender.stats()["datagrams"]

    def stats(self):
        return self.sender.stats()

    def start(self):
        self.sender.start(self.rate_hz)

    def stop(self):
        self.sender.stop()

    def close(self):
        self.sender.close()
