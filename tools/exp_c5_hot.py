#!/usr/bin/env python3
"""developer experiment (GPU box): config 5 at its shard size over temporally coherent cycles (1 % drift), cold start against the
hot start of the working sets (osot_solver_set_hotstart: what the reference's qpOASES back-end does across control cycles) --
the launch is its longest instance, and the longest instances are the ones with thirty-odd active constraints"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from opensot_amd import synth
from opensot_amd.solver import BatchedStack

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
drift = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
cfg = sys.argv[3] if len(sys.argv) > 3 else "C5"      # (C3 / C4: the velocity stacks, for comparison)
K, steps = 4, 40
plan, leaf = synth.make_id_stack(B, seed=5000) if cfg == "C5" else synth.make_velocity_stack(cfg, B, seed=4000)
rng = np.random.default_rng(77)
leaves = [leaf]
for _ in range(K - 1):
    leaves.append(synth.perturb(leaves[-1], rng, drift))


def run(hot):
    st = BatchedStack(plan, B, device=0, want_levels=False)
    if hot:
        st.set_hotstart(True)
    devs = [st.load_leaf(lf) for lf in leaves]
    for i in range(8):
        st.cycle(devs[i % K])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        st.cycle(devs[i % K])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    it = st.iterations[:B].cpu().numpy()
    ok = int((st.status[:B] == 0).sum().item())
    print(f"hot={hot}: {B * steps / el / 1e6:.3f} M solves/s, {1e3 * el / steps:.4f} ms/step, ok {ok}/{B}, iterations mean {it.mean():.1f} p99 {np.percentile(it, 99):.0f} max {it.max()}", flush=True)
    return st.dq[:B].double().cpu().numpy()


a = run(False)
b = run(True)
print("max |dq_hot - dq_cold| =", np.abs(a - b).max())
