"""developer helper (GPU box): per-instance timelines of osot_cycle over the bench's drifting control cycles ->
gpurun_out/timeline.npz (dur [steps][B] us, start, iterations of every step): input of the offline packing study"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
buf = torch.zeros((B, 4), dtype=torch.int64, device="cuda")
os.environ["OSOT_DEBUG_CYCLE_PROF"] = hex(buf.data_ptr())
from opensot_amd import synth
from opensot_amd.solver import BatchedStack
plan, leaf = synth.make_velocity_stack("C3", B, seed=3000)
st = BatchedStack(plan, B, device=0, want_levels=False)
rng = np.random.default_rng(77)
leaves = [leaf]
for _ in range(3):
    leaves.append(synth.perturb(leaves[-1], rng, 0.01))
dev_leaves, A_sets = [], []
for lf in leaves:
    st.A = [None if a is None else torch.empty_like(a) for a in st.A]
    dev_leaves.append(st.load_leaf(lf))
    A_sets.append(st.A)
durs, starts, its, casc = [], [], [], []
for step in range(12):
    k = step % 4
    st.A = A_sets[k]
    st.cycle(dev_leaves[k])
    torch.cuda.synchronize()
    c = buf.cpu().numpy()
    t0 = c[:, 2].min()
    starts.append((c[:, 2] - t0) / 100.0); durs.append((c[:, 3] - c[:, 2]) / 100.0); casc.append(c[:, 1].copy())
    its.append(st.iterations[:B].cpu().numpy().copy())
    print("step", step, "span %.1f us" % ((c[:, 3] - t0).max() / 100.0), "mean dur %.1f" % durs[-1].mean(), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/timeline.npz", dur=np.array(durs), start=np.array(starts), iters=np.array(its), casc=np.array(casc), slots=st.resident_waves())
