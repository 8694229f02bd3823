"""N>1 path on CPU: world_size-2 gloo processes shard the instances contiguously, each 'solves' its shard
and the all-gather reassembles the global dq in instance order (the path bench.py --gpus N takes with RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from opensot_amd.parallel import all_gather_dq, shard_range


def test_shard_range_partitions():
    for total in (0, 1, 7, 4096, 32768, 10):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _worker(rank, world, port, total, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(total, rank, world)
    # stand-in for the per-rank solve: dq[i] depends only on the GLOBAL instance index
    idx = torch.arange(lo, hi, dtype=torch.float64)
    local = idx[:, None] * 10.0 + torch.arange(n, dtype=torch.float64)[None, :]
    full = all_gather_dq(local, total)
    q.put((rank, full.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 7])
def test_two_rank_allgather_reassembles_global_order(total):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n = 5
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.arange(total)[:, None] * 10.0 + np.arange(n)[None, :]
    for r in range(2):
        np.testing.assert_array_equal(res[r], want)
