"""Channel sharding across the GPUs of one node.

Receiver channels are independent (rxa[channel], wdsp/RXA.c:29; ch[channel], wdsp/channel.c:29), so the
multi-GPU form is a partition of the channel index range, one process per GPU, with NO collective on the
data path.  The only cross-rank step is agreeing on the wall time of the timed region (max over ranks).
"""


def channel_range(rank, world, channels_per_rank):
    """Weak-scaling partition used by bench.py: rank r owns global channels [r*cpr, (r+1)*cpr)."""
    if not (0 <= rank < world) or channels_per_rank <= 0:
        raise ValueError("bad rank/world/channels")
    return range(rank * channels_per_rank, (rank + 1) * channels_per_rank)


def split_channels(total_channels, world):
    """Strong-scaling partition: contiguous ranges, ceil(total/world) per rank."""
    per = -(-total_channels // world)
    return [range(min(r * per, total_channels), min((r + 1) * per, total_channels)) for r in range(world)]


def max_over_ranks(seconds, device=None):
    """Max of a per-rank duration over the process group (identity when not initialised)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_throughput(samples_per_rank_per_step, world, steps, seconds_max):
    """Whole-job input samples per second."""
    return samples_per_rank_per_step * world * steps / seconds_max
