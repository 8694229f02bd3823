#!/usr/bin/env python3
"""throughput of every BASELINE.json configuration on one GPU (development aid; bench.py is the judged line)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opensot_amd import synth
from opensot_amd.solver import BatchedStack

CASES = (("C2", 1024), ("C3", 4096), ("C4", 4096), ("C5", 1024))
if len(sys.argv) > 1 and sys.argv[1] == "--batch-sweep":      # throughput of config 3 against the batch size
    CASES = tuple(("C3", b) for b in (1024, 2048, 4096, 8192, 16384, 32768))
for cfg, B in CASES:
    plan, leaf = synth.make_id_stack(B, seed=1) if cfg == "C5" else synth.make_velocity_stack(cfg, B, seed=1)
    st = BatchedStack(plan, B, device=0, want_levels=False)
    dev = st.load_leaf(leaf)
    for _ in range(3):
        st.update(dev); st.solve(B)
    torch.cuda.synchronize()
    st.set_timing(True)
    K = 20
    t0 = time.perf_counter()
    for _ in range(K):
        st.update(dev); st.solve(B)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms, cnt = st.kernel_time_ms()
    ok = int((st.status[:B] == 0).sum())
    print(f"{cfg}: B={B} n={plan.n} rows={[plan.m(k) for k in range(plan.L)]} nc={plan.nc}: "
          f"{B*K/el/1e6:.3f} M solves/s end-to-end, cascade kernel {ms*1e3:.1f} us/launch, "
          f"iters/solve {st.iterations[:B].float().mean().item():.1f}, ok {ok}/{B}")
