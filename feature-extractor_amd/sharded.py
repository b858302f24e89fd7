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


def gather_features(local, num_channels, dst=0, group=None, async_op=False):
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
    if world == 1:
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


class OscSink:
    """The sink side of the path: turns gathered smoothed vectors [C][12] into the datagrams
    OSCFeatureAnalysisOutput would send, one per channel, address /Audio/A<channel>
    (ref MainComponent.cpp:170, OSCFeatureAnalysisOutput.h:107).  `sock` is optional: tests and
    the bench only encode."""

    def __init__(self, encode, host="127.0.0.1", port=9000, sock=None):
        self.encode, self.addr, self.sock = encode, (host, port), sock

    def datagrams(self, smoothed):
        smoothed = np.asarray(smoothed, np.float32).reshape(-1, 12)
        return [self.encode("/Audio/A%d" % c, smoothed[c]) for c in range(smoothed.shape[0])]

    def send(self, smoothed):
        msgs = self.datagrams(smoothed)
        if self.sock is not None:
            for m in msgs:
                self.sock.sendto(m, self.addr)
        return len(msgs)
