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
    (`ncclAllGather(dq_shard) (+ status)`) in ONE collective per step: a rank's block is a flat fp64 buffer, its
    dq rows first and its int32 status words behind them (viewed as int32: the collective moves bytes) -- a small collective
    over xGMI is latency-bound, two of them cost twice.  `bind(stack)` makes the solver write its dq and status STRAIGHT
    into that block (the stack's output tensors become views of it), so a step is solve + one collective with no copy
    kernel in between; shards handed in from other tensors are copied.  Shards may be uneven (total % world != 0): every
    rank's block is sized for the largest shard; `dq` / `status` return the rows in global instance order.

    `overlap` (None = on for a world of more than one rank on GPUs): the collective is issued with async_op, so it runs on the
    process group's stream while the rank solves the next step (two blocks, alternating; with `bind` the caller re-binds every
    step, `ShardedCycle` does): step t + 1's solve is ordered behind step t's SOLVE only -- the gather of step t reads a block
    the solver does not write before step t + 2, whose `bind` waits for it.  With a world of one the collective is a 1 MB
    device copy and the in-stream form is as fast (round 3: 0.194 against 0.210 ms per step), hence the default."""

    def __init__(self, total, n, device, dtype, group=None, overlap=None, sizes=None):
        import torch
        import torch.distributed as dist
        self.torch = torch
        self.dist, self.group = dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.total, self.n = total, n
        # rows per rank: the block partition of `total`, or given explicitly (a LANE of a pipelined rank gathers the
        # matching sub-batches of all ranks: sizes[r] = rows of that lane on rank r, total = their sum)
        if sizes is not None:
            if len(sizes) != self.world or sum(sizes) != total:
                raise ValueError("sizes must list one row count per rank and add up to total")
            self.sizes = [int(v) for v in sizes]
        else:
            self.sizes = [shard_range(total, r, self.world)[1] - shard_range(total, r, self.world)[0] for r in range(self.world)]
        self.mx = max(self.sizes) if self.sizes else 0
        self.even = all(s == self.mx for s in self.sizes)
        self.overlap = (self.world > 1 if overlap is None else bool(overlap)) and torch.device(device).type == "cuda"
        nb = 2 if self.overlap else 1
        self.ndq = self.mx * n                       # doubles of dq in a block
        self.blk = self.ndq + (self.mx + 1) // 2     # + the status words, two per double
        self.send_b = [torch.zeros((self.blk,), dtype=dtype, device=device) for _ in range(nb)]
        self.recv_b = [torch.empty((self.world * self.blk,), dtype=dtype, device=device) for _ in range(nb)]
        self.work = [None] * nb
        self._send_views = [self._views(b) for b in self.send_b]     # (dq rows, status words) of each send block, made once
        self.i = 0
        self.last = 0
        self.has_status = False

    def _views(self, block):
        return block[:self.ndq].view(self.mx, self.n), block[self.ndq:].view(self.torch.int32)[:self.mx]

    def bind(self, stack):
        """the stack's dq / status outputs become views of the NEXT step's send block"""
        b = self.i % len(self.send_b)
        self._wait(b)   # (overlap: the asynchronous gather that last READ this block is over before the solver rewrites it)
        stack.dq, stack.status = self._send_views[b]

    def _gather(self, recv, send, async_op=False):
        if hasattr(self.dist, "all_gather_into_tensor"):
            return self.dist.all_gather_into_tensor(recv, send, group=self.group, async_op=async_op)
        # (older torch: the list form on views of the same receive buffer)
        return self.dist.all_gather(list(recv.chunk(self.world)), send, group=self.group, async_op=async_op)

    def _wait(self, b):
        if self.work[b] is not None:
            self.work[b].wait()   # (stream-level for RCCL: the current stream waits, the host does not)
            self.work[b] = None

    def __call__(self, dq_shard, status_shard=None):
        mine = self.sizes[self.rank]
        b = self.i % len(self.recv_b)
        self.i += 1
        self.last = b
        self.has_status = status_shard is not None
        self._wait(b)             # (overlap: the gather that last used this pair of blocks is over)
        sdq, sst = self._send_views[b]
        if dq_shard.data_ptr() != sdq.data_ptr():
            sdq[:mine].copy_(dq_shard[:mine])
        if self.has_status and status_shard.data_ptr() != sst.data_ptr():
            sst[:mine].copy_(status_shard[:mine])
        w = self._gather(self.recv_b[b], self.send_b[b], async_op=self.overlap)
        if self.overlap:
            self.work[b] = w

    def _collect(self, which):
        self._wait(self.last)
        buf = self.recv_b[self.last]
        parts = [self._views(buf[r * self.blk:(r + 1) * self.blk])[which][:self.sizes[r]] for r in range(self.world)]
        return parts[0] if self.world == 1 else self.torch.cat(parts, dim=0)

    @property
    def dq(self):
        """the gathered dq of the LAST call, [total][n]"""
        return self._collect(0)

    @property
    def status(self):
        return self._collect(1) if self.has_status else None


class ShardedCycle:
    """one step of the data-parallel control loop on this rank: AutoStack::update + Solver::solve of the rank's shard
    (any object with .update(dev_leaf), .solve(B), .dq, .status, .A -- a BatchedStack, or a stub in the CPU tests), then
    the gather of the solved shards.  The steps rotate through K temporally coherent cycles, each with its own leaf
    inputs and stacked Jacobians (nothing is copied inside a step: the stack just points at the cycle's buffers)."""

    def __init__(self, stack, dev_leaves, A_sets, B_local, gather=None, bind=True):
        self.stack, self.dev_leaves, self.A_sets, self.B = stack, dev_leaves, A_sets, B_local
        self.gather = gather
        self.bind = bind and gather is not None and hasattr(stack, "cycle")   # (a real BatchedStack: outputs are re-pointable)
        self.i = 0
        # a real BatchedStack can keep the argument structs of a (cycle inputs, Jacobian set, output block) combination:
        # the dev_leaves / A_sets handed in here are never replaced, only written into
        import inspect
        self._cached = hasattr(stack, "cycle") and "cached" in inspect.signature(stack.cycle).parameters

    def step(self):
        k = self.i % len(self.dev_leaves)
        self.i += 1
        if self.A_sets is not None:
            self.stack.A = self.A_sets[k]
        if self.gather is not None and self.bind:
            self.gather.bind(self.stack)     # the solver writes dq / status straight into the collective's send block
        if hasattr(self.stack, "cycle"):     # update + solve in one launch (BatchedStack.cycle), same results
            if self._cached:
                self.stack.cycle(self.dev_leaves[k], cached=True)
            else:
                self.stack.cycle(self.dev_leaves[k])
        else:
            self.stack.update(self.dev_leaves[k])
            self.stack.solve(self.B)
        if self.gather is not None:
            self.gather(self.stack.dq[:self.B], self.stack.status[:self.B])


def lane_ranges(B, lanes):
    """[lo, hi) of each of the `lanes` contiguous sub-batches of a rank's B instances (same rule as shard_range)"""
    return [shard_range(B, i, lanes) for i in range(lanes)]


def suggest_lanes(B, resident, min_lanes=2, max_lanes=4):
    """How many sub-batches ("lanes") a rank's B instances are driven as (PipelinedCycle), from what the chip holds at once.

    One wavefront solves one instance and the device keeps `resident` of them in flight for the plan's kernel
    (BatchedStack.resident_waves() = osot_solver_resident_waves: CUs x wavefronts per CU from the kernel's register and LDS
    footprint; resident_waves_nhqp() for the null-space front-end), so a lane's launch of b = B / S instances runs as
    ceil(b / resident) ROUNDS, and the part of its last round that stays empty is slots nothing fills until the launch's longest
    instance ends.  The rule: among min_lanes .. max_lanes take the count whose launches waste the smallest fraction of their rounds,
    1 - b / (ceil(b / resident) resident); ties go to the fewer lanes (fewer, larger launches: less fill and drain per timed region).
    At least two lanes so that one lane's tail is covered by the other's next launch (24.3 -> 27.1 M on one stream
    against two).  What it reproduces (rounds 4-5 found these by hand, tools/exp_*lanes*.py): BASELINE config 3, 4096 instances on
    2048 slots -> 2 (two launches of exactly one round); the 35-coordinate COMAN stacks on the 40-lane kernel's 1792 slots -> 3
    (1365 = one round; 2048 would be a round and a seventh); the null-space front-end at config 3 on 1536 slots -> 3; its 64-column
    preparation on 1024 slots -> 2 (2048 = exactly two rounds)."""
    B, resident = int(B), max(1, int(resident))
    if B < 2 * min_lanes:
        return 1
    best, best_waste = min_lanes, None
    for S in range(min_lanes, max_lanes + 1):
        b = -(-B // S)                               # the largest lane (shard_range hands the remainder to the first lanes)
        rounds = -(-b // resident)
        waste = 1.0 - b / float(rounds * resident)
        if best_waste is None or waste < best_waste - 1e-9:
            best, best_waste = S, waste
    return best


class PipelinedCycle:
    """A rank's shard as S contiguous SUB-BATCHES ("lanes"), each with its own solver state and its own stream, and NO
    join between steps: lane j's step t+1 is enqueued behind lane j's step t only.  That is exactly the dependency the
    control loop has -- instance i's cycle t+1 needs instance i's cycle t and nothing else (instances are
    independent) -- while a single launch per step also makes every instance wait for the slowest instance of the whole
    batch.  One wavefront solves one instance and the chip holds a fixed number of them, so a 4096-instance launch is two
    rounds whose tail (a few wavefronts with twice the average iteration count) leaves most of the chip idle; with two
    lanes the other lane's next launch fills those slots.  Measured at BASELINE config 3, B = 4096: 24.2 -> 27.3 M solves/s
    with 2 lanes (4 lanes: the same; the floor is the longest single instance, which every step has to wait for in ITS
    lane).  Results are identical to the single-launch form (instances do not interact).

    lanes: a list of ShardedCycle (each over its own stack / sub-batch / gather); streams: one per lane (torch.cuda.Stream)
    or None to run the lanes in turn on the current stream (CPU tests).  `step()` enqueues one step of every lane;
    completion is the caller's synchronise (timed_steps brackets K steps with one on each side)."""

    def __init__(self, lanes, streams=None):
        self.lanes = list(lanes)
        self.streams = list(streams) if streams is not None else [None] * len(self.lanes)
        if len(self.streams) != len(self.lanes):
            raise ValueError("one stream per lane")
        self._torch = None
        if any(st is not None for st in self.streams):
            import torch
            self._torch = torch
        # a lane whose only device work is the solver's own launches takes its stream as an argument (BatchedStack.stream);
        # only a lane with a collective behind the solve needs torch's stream context (the gather runs on the current stream)
        self._ctx = []
        for lane, st in zip(self.lanes, self.streams):
            direct = st is not None and lane.gather is None and hasattr(lane.stack, "stream")
            if direct:
                lane.stack.stream = st
            self._ctx.append(st is not None and not direct)

    def step(self):
        for lane, st, ctx in zip(self.lanes, self.streams, self._ctx):
            if not ctx:
                lane.step()
            else:
                with self._torch.cuda.stream(st):
                    lane.step()

    def capture(self, steps):
        """`steps` steps of every lane as HIP graphs (torch.cuda.CUDAGraph around the C-ABI launches; call after warm-up steps).
        `replay()` then enqueues steps x lanes launches with one graph launch per lane: the dependent launches of a lane follow
        each other ~4 us sooner than the same launches submitted one by one (measured: 27.7 -> 28.5 M solves/s at BASELINE
        config 3 with two lanes, 22.5 -> 25.1 M with one), and the host does almost nothing.  Only for lanes whose step is the
        solver's own launch (no collective behind it).  `steps` must be even: the solver alternates two sets of dispatch-order
        buffers from launch to launch, and a replay has to leave them where the host believes they are.
        A graph starts on the cycle of the rotation it was captured on, so ONE GRAPH PER STARTING CYCLE that the replays can
        meet is captured ((start + r steps) mod K, r = 0, 1, ...: a single graph when steps is a multiple of the K cycles a lane
        rotates through): replays follow the rotation exactly like the same steps submitted one by one, and the state carried
        from cycle to cycle (cost estimates, hot-start sets) never sees a jump at a replay boundary."""
        if self._torch is None or any(st is None for st in self.streams):
            raise RuntimeError("graph capture needs one stream per lane")
        # (a lane with a collective behind its solve is captured with it -- RCCL collectives can be recorded into a HIP graph -- as
        #  long as the collective is issued in the lane's own stream: the asynchronous two-block form keeps host-side work handles)
        if any(getattr(lane.gather, "overlap", False) for lane in self.lanes):
            raise RuntimeError("graph capture needs in-stream collectives (ShardGather(overlap=False))")
        if steps < 2 or steps % 2:
            raise ValueError("an even number of steps per graph")
        K = len(getattr(self.lanes[0], "dev_leaves", ())) or 1
        if any((len(getattr(lane, "dev_leaves", ())) or 1) != K or lane.i != self.lanes[0].i for lane in self.lanes):
            raise ValueError("lanes must rotate through the same number of cycles, in step")
        i0 = self.lanes[0].i
        phases = sorted({(i0 + r * steps) % K for r in range(K)})
        # the solver's host-side launch state (which half of the order buffers the next launch reads) advances during capture
        # although nothing runs (an even number of launches per graph leaves it where it was); if the capture fails the dispatch
        # order is forgotten (the next plain launch runs in order and builds a fresh one: osot_solver_set_schedule)
        graphs = {}
        try:
            for ph in phases:
                per_lane = []
                for lane, st in zip(self.lanes, self.streams):
                    lane.i = ph
                    if lane.gather is not None and (steps % len(lane.gather.send_b)):
                        raise ValueError("steps per graph must be a multiple of the gather's blocks")
                    g = self._torch.cuda.CUDAGraph()
                    with self._torch.cuda.graph(g, stream=st):
                        for _ in range(steps):
                            lane.step()
                    per_lane.append(g)
                graphs[ph] = per_lane
        except Exception:
            for lane in self.lanes:
                if hasattr(lane.stack, "set_schedule"):
                    lane.stack.set_schedule(True)
            raise
        finally:
            for lane in self.lanes:
                lane.i = i0
        self._graphs, self.graph_steps, self._K = graphs, steps, K
        self._replay_events = []

    def replay(self, timed=False):
        """one replay of every lane's graph: `graph_steps` steps of the whole shard, continuing the rotation.  timed: a pair of
        events on the lane's stream around its graph (replay_launch_ms() turns them into the average duration of one launch of the
        lane INSIDE the replays, the gap between consecutive launches of a graph included)"""
        ph = self.lanes[0].i % self._K
        if ph not in self._graphs:
            # (capture() records the phases whole replays can reach from where it was called; plain step() calls in between that
            #  are not a multiple of the rotation land elsewhere)
            raise RuntimeError(f"no graph was captured for cycle {ph} of the rotation (captured: {sorted(self._graphs)}): "
                               "call capture() again after stepping, or step a multiple of the rotation between replays")
        for j, (lane, g, st) in enumerate(zip(self.lanes, self._graphs[ph], self.streams)):
            with self._torch.cuda.stream(st):
                if timed:
                    e0, e1 = self._torch.cuda.Event(enable_timing=True), self._torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                    g.replay()
                    e1.record(st)
                    self._replay_events.append((e0, e1))
                else:
                    g.replay()
            lane.i += self.graph_steps

    def replay_launch_ms(self):
        """average over the timed replays (all lanes) of graph time / steps per graph, in ms; None without timed replays.  Call after
        a synchronise."""
        ev, self._replay_events = self._replay_events, []
        if not ev:
            return None
        return sum(a.elapsed_time(b) for a, b in ev) / (len(ev) * self.graph_steps)


class StubStack:
    """NOT a solver: a stand-in with BatchedStack's per-step surface on CPU tensors, so that bench.py's own launcher,
    sharding, lanes and gather can be run end to end where there is no GPU (`bench.py --backend stub`, world of two on gloo
    in tests/test_distributed_cpu.py).  Its "dq" is a checksum of the cycle's inputs per instance; bench.py labels every line
    it produces as a stub line."""

    def __init__(self, plan, max_batch, device=0, want_levels=False):
        import torch
        self.torch, self.plan, self.max_batch = torch, plan, int(max_batch)
        self.device = torch.device("cpu")
        n, L, B = plan.n, plan.L, self.max_batch
        self.A = [torch.zeros((B, plan.ma(k), n), dtype=torch.float64) if plan.ma(k) else None for k in range(L)]
        self.dq = torch.zeros((B, n), dtype=torch.float64)
        self.status = torch.full((B,), -1, dtype=torch.int32)
        self.launches = 0

    def load_leaf(self, leaf):
        import numpy as np
        B = leaf["B"]
        for k in range(self.plan.L):
            if self.A[k] is not None:
                self.A[k][:B].copy_(self.torch.as_tensor(np.ascontiguousarray(leaf["A"][k])))
        return {"B": B}

    def update(self, dev_leaf):
        self._B = dev_leaf["B"]

    def solve(self, B):
        acc = self.torch.zeros((B, self.plan.n), dtype=self.torch.float64)
        for a in self.A:
            if a is not None:
                acc += a[:B].sum(dim=1)
        self.dq[:B] = acc
        self.status[:B] = 0
        self.launches += 1

    def cycle(self, dev_leaf, write_weights=True, cached=False):
        self.update(dev_leaf)
        self.solve(dev_leaf["B"])

    def set_timing(self, on, stride=1):
        self.launches = 0

    def set_schedule(self, longest_first=True):
        pass

    def kernel_time_ms(self, reset=True):
        c = self.launches
        if reset:
            self.launches = 0
        return (1.0e-3, c)


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
    timed_steps.host_enqueue_s = time.perf_counter() - t0    # this rank's time to ENQUEUE the steps (diagnostic: host-bound if ~ elapsed)
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
