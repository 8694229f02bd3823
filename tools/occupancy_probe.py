#!/usr/bin/env python3
"""resident wavefronts of the cascade kernel for the configuration given (development aid): what the LDS budget buys."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opensot_amd import synth
from opensot_amd.solver import BatchedStack

cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
for B in [int(a) for a in sys.argv[2:]] or [256, 512, 768, 1024, 2048, 4096]:
    plan, leaf = synth.make_id_stack(B, seed=1) if cfg == "C5" else synth.make_velocity_stack(cfg, B, seed=1)
    st = BatchedStack(plan, B, device=0, want_levels=False)
    dev = st.load_leaf(leaf)
    for _ in range(3):
        st.update(dev); st.solve(B)
    torch.cuda.synchronize()
    st.set_timing(True)
    K = 10
    for _ in range(K):
        st.update(dev); st.solve(B)
    torch.cuda.synchronize()
    ms, cnt = st.kernel_time_ms()
    print(f"{cfg} B={B}: resident waves {st.resident_waves()}, cascade kernel {ms*1e3:.1f} us/launch, {B/ms/1e3:.3f} M solves/s (kernel)")
