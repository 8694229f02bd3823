#!/usr/bin/env python3
"""randomised sweep of the hot start (osot_solver_set_hotstart) on the GPU: closed sequences of temporally coherent cycles with
drifts between 0.1 % and 5 % per cycle, hot against cold on the same inputs -- same status for every instance, same dq (1e-8 x
scale) for every instance both solve; not part of the pytest suite (run on the GPU box: python tests/stress_hotstart.py [seed] [sequences])"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from opensot_amd import synth
from opensot_amd.solver import BatchedStack

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rng = np.random.default_rng(seed)
t0 = time.time()
bad, solves, worst, it_hot, it_cold, max_hot, max_cold = 0, 0, 0.0, 0, 0, 0, 0
for q in range(N):
    cfg = ("C3", "C4", "C5")[q % 3]
    B = 128 if cfg == "C5" else 384
    s = int(rng.integers(1, 1 << 30))
    plan, leaf = synth.make_id_stack(B, seed=s) if cfg == "C5" else synth.make_velocity_stack(cfg, B, seed=s)
    cold, hot = BatchedStack(plan, B, device=0, want_levels=False), BatchedStack(plan, B, device=0, want_levels=False)
    hot.set_hotstart(True)
    lf = leaf
    for cyc in range(10):
        if cyc:
            lf = synth.perturb(lf, rng, float(rng.choice([0.001, 0.003, 0.01, 0.02, 0.05])))
        for st in (cold, hot):
            st.cycle(st.load_leaf(lf))
        torch.cuda.synchronize()
        s0, s1 = cold.status[:B].cpu().numpy(), hot.status[:B].cpu().numpy()
        d0, d1 = cold.dq[:B].double().cpu().numpy(), hot.dq[:B].double().cpu().numpy()
        ok = (s0 == 0) & (s1 == 0)
        scale = max(1.0, np.abs(d0[ok]).max()) if ok.any() else 1.0
        diff = np.abs(d0[ok] - d1[ok]).max() / scale if ok.any() else 0.0
        worst = max(worst, diff)
        if (s0 != s1).any() or diff > 1e-8:
            bad += 1
            print(f"  MISMATCH cfg {cfg} seed {s} cycle {cyc}: status differs for {(s0 != s1).sum()} instances, max scaled |ddq| {diff:.2e}")
        solves += B
        a, b = cold.iterations[:B].cpu().numpy(), hot.iterations[:B].cpu().numpy()
        it_cold += int(a.sum()); it_hot += int(b.sum()); max_cold = max(max_cold, int(a.max())); max_hot = max(max_hot, int(b.max()))
print(f"seed {seed}: {N} sequences x 10 cycles, {solves} solves per mode in {time.time() - t0:.0f} s: {bad} cycles with a mismatch; worst scaled "
      f"|dq_hot - dq_cold| {worst:.2e}; iterations per solve cold {it_cold / solves:.1f} hot {it_hot / solves:.1f}; longest cold {max_cold} hot {max_hot}")
