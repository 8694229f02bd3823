#!/usr/bin/env python3
"""developer experiment (GPU box): the steps of a lane captured in a HIP graph (torch.cuda.CUDAGraph around the C-ABI launches)
against plain stream launches -- does a graph shorten the hand-over between two dependent launches of a lane?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from opensot_amd import synth
from opensot_amd.solver import BatchedStack

B, K = 4096, 4
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4      # steps of a lane per graph
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


def lane_steps(s, n):
    st, devs, As = lanes[s]
    for i in range(n):
        k = i % K
        st.A = As[k]
        st.cycle(devs[k])


steps = 48
# plain
for s in range(S):
    with torch.cuda.stream(streams[s]):
        lane_steps(s, 8)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps // K):
    for s in range(S):
        with torch.cuda.stream(streams[s]):
            lane_steps(s, K)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"plain launches : {B * steps / el / 1e6:.2f} M solves/s, {1e3 * el / steps:.4f} ms/step")
# graphs: K steps of a lane per graph (even number of launches: the solver's double-buffered order state comes back to where it was)
graphs = []
try:
    for s in range(S):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=streams[s]):
            lane_steps(s, G)
        graphs.append(g)
    torch.cuda.synchronize()
    for _ in range(2):
        for s in range(S):
            with torch.cuda.stream(streams[s]):
                graphs[s].replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps // G):
        for s in range(S):
            with torch.cuda.stream(streams[s]):
                graphs[s].replay()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ok = sum(int((st.status == 0).sum().item()) for st, _, _ in lanes)
    print(f"graph replay S={S} G={G}: {B * steps / el / 1e6:.2f} M solves/s, {1e3 * el / steps:.4f} ms/step, ok {ok}/{B}")
except Exception as e:
    print("graph capture failed:", repr(e)[:300])
