#!/usr/bin/env python3
"""developer experiment (GPU box): the headline batch as S sub-batches on S streams, pipelined across steps
(no join between steps: instance i's cycle t+1 only waits for instance i's cycle t), against the single launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from opensot_amd import synth
from opensot_amd.solver import BatchedStack

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = 4
steps = 50
plan, leaf = synth.make_velocity_stack("C3", B, seed=3000)
rng = np.random.default_rng(77)
leaves = [leaf]
for _ in range(K - 1):
    leaves.append(synth.perturb(leaves[-1], rng, 0.01))


def sub(lf, lo, hi):
    cut = lambda a: None if a is None else a[lo:hi]
    return {"B": hi - lo, "A": [cut(a) for a in lf["A"]],
            "task": [[tuple(cut(x) for x in t) for t in lev] for lev in lf["task"]],
            "bound": [tuple(cut(x) for x in t) for t in lf["bound"]],
            "rows": [tuple(cut(x) for x in t) for t in lf["rows"]]}


def run(S, hot=False):
    streams = [torch.cuda.Stream() for _ in range(S)]
    lanes = []
    for s in range(S):
        lo, hi = s * B // S, (s + 1) * B // S
        st = BatchedStack(plan, hi - lo, device=0, want_levels=False)
        if hot:
            st.set_hotstart(True)
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
    t0 = time.perf_counter()
    for i in range(steps):
        step(8 + i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ok = sum(int((st.status == 0).sum().item()) for st, _, _ in lanes)
    dq = torch.cat([st.dq for st, _, _ in lanes]).cpu().numpy()
    it = torch.cat([st.iterations for st, _, _ in lanes]).cpu().numpy()
    print(f"S={S} hot={hot}: {B * steps / el / 1e6:.2f} M solves/s, {1e3 * el / steps:.4f} ms/step, ok {ok}/{B}, iters mean {it.mean():.2f} max {it.max()}", flush=True)
    return dq


d1 = run(1)
for S in (2, 4, 8):
    d = run(S)
    print("   max |dq - dq(S=1)| =", np.abs(d - d1).max())
dh = run(1, hot=True)
print("   hot vs cold max |ddq| =", np.abs(dh - d1).max())
dh = run(2, hot=True)
print("   hot vs cold max |ddq| =", np.abs(dh - d1).max())
