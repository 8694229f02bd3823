"""Instance sharding over ranks: contiguous block partition, static plan replicated, no data-path
collective; the only exchange is the all-gather of the solved dq (and status) shards.

bench.py and tests/test_distributed_cpu.py drive THE SAME objects: `ShardGather` (the collective) and `ShardedCycle` +
`timed_steps` (the per-rank step loop with its barrier / synchronise bracket).  On GPUs the backend is RCCL over xGMI, on
CPU tensors gloo; the code path is one.
"""
import time


def shard_range(total, rank, world):
    """[lo, hi) of the instances owned by `rank`; the first (total % world) ranks get one extra."""
    if world < 1 or rank < 0 or rank >= world or total < 0:
        raise ValueError("bad shard arguments")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


class ShardGather:
    """all-gather of the per-rank result shards into global instance order: dq [total][n] fp64 and status [total] int32
    (`ncclAllGather(dq_shard) (+ status)`).  Shards may be uneven (total % world != 0): every rank contributes a
    block padded to the largest shard, ONE `all_gather_into_tensor` per array moves them, and `dq` / `status` return
    views (even shards) or compacted copies (uneven).  Buffers are allocated once; call = enqueue on the current stream."""

    def __init__(self, total, n, device, dtype, group=None):
        import torch
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.total, self.n = total, n
        self.sizes = [shard_range(total, r, self.world)[1] - shard_range(total, r, self.world)[0] for r in range(self.world)]
        self.mx = max(self.sizes) if self.sizes else 0
        self.even = all(s == self.mx for s in self.sizes)
        self.recv_dq = torch.empty((self.world * self.mx, n), dtype=dtype, device=device)
        self.recv_status = torch.empty((self.world * self.mx,), dtype=torch.int32, device=device)
        self.pad_dq = None if self.even else torch.zeros((self.mx, n), dtype=dtype, device=device)
        self.pad_status = None if self.even else torch.zeros((self.mx,), dtype=torch.int32, device=device)
        self.has_status = False

    def _gather(self, recv, send):
        if hasattr(self.dist, "all_gather_into_tensor"):
            self.dist.all_gather_into_tensor(recv, send, group=self.group)
        else:   # (older torch: the list form on views of the same receive buffer)
            self.dist.all_gather(list(recv.chunk(self.world)), send, group=self.group)

    def __call__(self, dq_shard, status_shard=None):
        mine = self.sizes[self.rank]
        if self.even:
            send = dq_shard[:mine]
        else:
            self.pad_dq[:mine].copy_(dq_shard[:mine])
            send = self.pad_dq
        self._gather(self.recv_dq, send)
        self.has_status = status_shard is not None
        if self.has_status:
            if self.even:
                sst = status_shard[:mine]
            else:
                self.pad_status[:mine].copy_(status_shard[:mine])
                sst = self.pad_status
            self._gather(self.recv_status, sst)

    def _compact(self, buf):
        import torch
        if self.even:
            return buf
        return torch.cat([buf[r * self.mx: r * self.mx + self.sizes[r]] for r in range(self.world)], dim=0)

    @property
    def dq(self):
        return self._compact(self.recv_dq)

    @property
    def status(self):
        return self._compact(self.recv_status) if self.has_status else None


class ShardedCycle:
    """one step of the data-parallel control loop on this rank: AutoStack::update + Solver::solve of the rank's shard
    (any object with .update(dev_leaf), .solve(B), .dq, .status, .A -- a BatchedStack, or a stub in the CPU tests), then
    the gather of the solved shards.  The steps rotate through K temporally coherent cycles, each with its own leaf
    inputs and stacked Jacobians (nothing is copied inside a step: the stack just points at the cycle's buffers)."""

    def __init__(self, stack, dev_leaves, A_sets, B_local, gather=None):
        self.stack, self.dev_leaves, self.A_sets, self.B = stack, dev_leaves, A_sets, B_local
        self.gather = gather
        self.i = 0

    def step(self):
        k = self.i % len(self.dev_leaves)
        self.i += 1
        if self.A_sets is not None:
            self.stack.A = self.A_sets[k]
        if hasattr(self.stack, "cycle"):     # update + solve in one launch (BatchedStack.cycle), same results
            self.stack.cycle(self.dev_leaves[k])
        else:
            self.stack.update(self.dev_leaves[k])
            self.stack.solve(self.B)
        if self.gather is not None:
            self.gather(self.stack.dq[:self.B], self.stack.status[:self.B])


def timed_steps(step, steps, warmup, sync, dist=None, device=None):
    """the benchmark contract's bracket: `warmup` untimed steps, then EXACTLY `steps` steps between barrier +
    synchronise on both sides; returns the elapsed seconds, MAX over ranks."""
    import torch
    for _ in range(warmup):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def all_gather_dq(local_dq, total, group=None):
    """convenience form of ShardGather for one call: [total][n] on every rank"""
    g = ShardGather(total, local_dq.shape[1], local_dq.device, local_dq.dtype, group)
    g(local_dq)
    return g.dq.clone()
