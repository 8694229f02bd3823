#!/usr/bin/env python3
"""iteration-count census of a configuration's batch (development aid): the longest instance bounds a one-round launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from opensot_amd import synth
from opensot_amd.solver import BatchedStack
cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
plan, leaf = synth.make_id_stack(B, seed=seed) if cfg == "C5" else synth.make_velocity_stack(cfg, B, seed=seed)
st = BatchedStack(plan, B, device=0, want_levels=False)
st.update(st.load_leaf(leaf)); st.solve(B); torch.cuda.synchronize()
it = st.iterations[:B].cpu().numpy()
print(f"{cfg} B={B} seed={seed}: iterations mean {it.mean():.1f} p50 {np.percentile(it,50):.0f} p90 {np.percentile(it,90):.0f} "
      f"p99 {np.percentile(it,99):.0f} max {it.max()}  top: {sorted(zip(it.tolist(), range(B)))[-6:]}")
cyc = st.profile_phases(B)
t = cyc[:, 7]
print("cycles per instance: mean %.0f p99 %.0f max %.0f; top by cycles: %s" % (t.mean(), np.percentile(t, 99), t.max(), [(int(t[i]), int(i), int(it[i])) for i in np.argsort(t)[-6:]]))
