"""The N > 1 path of bench.py on CPU: two processes, gloo backend."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from quisk_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    chans = shard.channel_range(rank, world, 256)
    # every rank reports its own channels and a rank-dependent duration; the job time is the max
    mine = torch.zeros(world * 256, dtype=torch.int64)
    mine[chans.start:chans.stop] = 1
    dist.all_reduce(mine)
    dt = shard.max_over_ranks(0.010 * (rank + 1))
    thr = shard.job_throughput(256 * 1024.0, world, 5, dt)
    q.put((rank, int(mine.min()), int(mine.max()), dt, thr))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_partition_and_timing():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, lo, hi, dt, thr in res:
        assert lo == 1 and hi == 1                 # the ranges tile [0, 512) exactly once
        assert abs(dt - 0.020) < 1e-12             # max over ranks
        assert abs(thr - 256 * 1024.0 * 2 * 5 / 0.020) < 1e-3


def test_strong_split_covers_everything():
    for total, world in ((2048, 8), (256, 8), (10, 4), (3, 8)):
        parts = shard.split_channels(total, world)
        flat = [c for r in parts for c in r]
        assert flat == list(range(total)) and len(parts) == world
