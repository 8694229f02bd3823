"""developer helper (GPU box): iteration-count tail of config 5 (which instances set the launch time) -- dumps the worst instance's
assembled problem for the emulator (tests/helpers.emu_cascade)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from opensot_amd import synth
from opensot_amd.solver import BatchedStack
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
B = 1024
plan, leaf = synth.make_id_stack(B, seed=seed)
st = BatchedStack(plan, B, device=0, want_levels=False)
st.update(st.load_leaf(leaf)); st.solve(B); torch.cuda.synchronize()
it = st.iterations[:B].cpu().numpy()
order = np.argsort(-it)
print("seed", seed, "iterations: mean %.1f p50 %d p90 %d p99 %d max %d" % (it.mean(), np.percentile(it, 50), np.percentile(it, 90), np.percentile(it, 99), it.max()))
print("top:", [(int(i), int(it[i])) for i in order[:12]])
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "c5_tail_seed%d.npy" % seed), order[:12])
