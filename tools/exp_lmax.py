#!/usr/bin/env python3
"""developer experiment (GPU box): is the headline bound by its LONGEST instance?  The headline batch (two lanes of 2048, stream
launches, the bench's four drifting cycles) with the active-set iteration cap of the plan lowered: the instances above the cap end
early (status MAX_ITER: their answers are void -- this is a timing probe, not a product mode), so the launch's longest job shrinks
while the mean job hardly moves.  If the rate follows the cap, the launch IS its longest instance (rate <= B / L_max).
usage: python tools/exp_lmax.py [cap ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from opensot_amd import synth
from opensot_amd.solver import BatchedStack
sys.path.insert(0, os.path.join(ROOT, "tools"))

K, steps = 4, 50


def sub(lf, lo, hi):
    cut = lambda a: None if a is None else a[lo:hi]
    return {"B": hi - lo, "A": [cut(a) for a in lf["A"]],
            "task": [[tuple(cut(x) for x in t) for t in lev] for lev in lf["task"]],
            "bound": [tuple(cut(x) for x in t) for t in lf["bound"]],
            "rows": [tuple(cut(x) for x in t) for t in lf["rows"]]}


def run(B, S, cap, reps=3):
    plan, leaf = synth.make_velocity_stack("C3", B, seed=3000)
    plan.max_iter = cap
    rng = np.random.default_rng(77)
    leaves = [leaf]
    for _ in range(K - 1):
        leaves.append(synth.perturb(leaves[-1], rng, 0.01))
    streams = [torch.cuda.Stream() for _ in range(S)]
    lanes = []
    for s in range(S):
        lo, hi = s * B // S, (s + 1) * B // S
        st = BatchedStack(plan, hi - lo, device=0, want_levels=False)
        devs, As = [], []
        for lf in leaves:
            st.A = [None if a is None else torch.empty_like(a) for a in st.A]
            devs.append(st.load_leaf(sub(lf, lo, hi)))
            As.append(st.A)
        lanes.append((st, devs, As))
    torch.cuda.synchronize()

    def step(i):
        k = i % K
        for s, (st, devs, As) in enumerate(lanes):
            with torch.cuda.stream(streams[s]):
                st.A = As[k]
                st.cycle(devs[k])
    for i in range(8):
        step(i)
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(reps):
        t0 = time.perf_counter()
        for i in range(steps):
            step(8 + i)
        torch.cuda.synchronize()
        best = max(best, B * steps / (time.perf_counter() - t0) / 1e6)
    ok = sum(int((st.status == 0).sum().item()) for st, _, _ in lanes)
    it = torch.cat([st.iterations for st, _, _ in lanes]).double()
    return best, ok, float(it.mean()), float(it.max())


caps = [int(a) for a in sys.argv[1:]] or [0, 60, 50, 45, 40]
for B, S in ((4096, 2), (32768, 1)):
    for cap in caps:
        v, ok, im, ix = run(B, S, cap)
        print(f"B={B} S={S} cap={cap}: {v:.3f} M/s ok {ok}/{B} iterations mean {im:.1f} max {ix:.0f}", flush=True)
