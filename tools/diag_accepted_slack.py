"""developer diagnostic (GPU): the accepted-slack fixture through the cascade, per-level evidence"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import accepted_slack_instance, answer_is_acceptable
from opensot_amd.solver import BatchedStack
from oracle import lexcheck as lc
np.set_printoptions(precision=3, linewidth=200)
plan, asm, wit = accepted_slack_instance()
st = BatchedStack(plan, 1, device=0, want_levels=True)
st.load_assembled(asm); st.solve(1); torch.cuda.synchronize()
dq = st.dq[:1].cpu().numpy(); xl = st.x_levels[:1].cpu().numpy()
print("status", st.status[:1].cpu().numpy(), "iters", st.iterations[:1].cpu().numpy(), "slack", st.accepted_slack[:1].cpu().numpy())
print("acceptable:", answer_is_acceptable(asm, 0, dq[0], wit))
C, lo, up = asm["C"][0], asm["lo"][0], asm["up"][0]
for k in range(plan.L):
    x = xl[0, k]
    cx = C @ x
    print(f"level {k}: C x = {cx}, lo = {lo}, up = {up}, viol = {np.maximum(lo - cx, cx - up).max():.3e}")
    for j in range(k):
        print(f"   drift of optimality rows of level {j}: {np.abs(asm['A'][j][0] @ (x - xl[0, j])).max():.3e}  |x_k - x_j| = {np.abs(x - xl[0, j]).max():.3e}")
for nm, x, ok in wit:
    print(nm, ok, "dist", np.abs(dq[0] - x).max(), "viol", lc.global_violation(asm, 0, x), "lex", lc.lex_costs(asm, 0, x))
print("device lex", lc.lex_costs(asm, 0, dq[0]), "viol", lc.global_violation(asm, 0, dq[0]))
np.save(os.path.join(ROOT, "gpurun_out", "diag_xl.npy"), xl)
