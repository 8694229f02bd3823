import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from opensot_amd import synth
from opensot_amd.solver import BatchedStack
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
plan, leaf = synth.make_id_stack(B, seed=3000) if cfg == "C5" else synth.make_velocity_stack(cfg, B, seed=3000)
st = BatchedStack(plan, B, device=0, want_levels=False)
st.update(st.load_leaf(leaf)); st.solve(B); torch.cuda.synchronize()
cyc = st.profile_phases(B)
m = cyc.mean(axis=0)
for name, v in zip(st.PHASES, m): print(f"{name:16s} {v:10.0f} cycles  {100*v/m[7]:5.1f}%")
print("iters mean", st.iterations[:B].float().mean().item())
t = cyc[:, 7]
print("total cycles per instance: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f" % (t.mean(), np.percentile(t, 50), np.percentile(t, 90), np.percentile(t, 99), t.max()))
it = st.iterations[:B].cpu().numpy()
print("iterations: mean %.1f p50 %d p90 %d p99 %d max %d" % (it.mean(), np.percentile(it, 50), np.percentile(it, 90), np.percentile(it, 99), it.max()))
heavy = it >= np.percentile(it, 99)
if heavy.any():
    extra = (it[heavy] - 30).mean()
    print("instances with >= p99 iterations (%d of them, mean %.1f iterations): in:* cycles per inequality iteration" % (heavy.sum(), it[heavy].mean()))
    for name, v in zip(st.PHASES[12:], cyc[heavy][:, 12:].mean(axis=0)):
        print(f"   {name:16s} {v / extra:8.0f}")
    print("   inequalities total per iteration %.0f" % (cyc[heavy][:, 5].mean() / extra))
