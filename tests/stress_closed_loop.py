"""TEST INFRASTRUCTURE (run by hand on the GPU box): closed-loop robustness sweep.  Many instances of the 32-DoF humanoid
chase random wrist targets (some through the body) with the self-collision rows on, everything on the device; every
instance the cascade does not solve is re-solved by the witnesses (qpOASES run to the exact optimum, the eiQuadProg
restatement): a failure the witnesses do not share is a product bug and is dumped to gpurun_out/."""
import os, sys, time
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))
import numpy as np, torch
from opensot_amd import abi, kinematics as kin
from opensot_amd.plan import StackPlan, Task, Bound, Rows, subtask, eps_abs_from_factor
from opensot_amd.solver import BatchedStack
from oracle import pyoracle as oracle

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cycles = int(sys.argv[3]) if len(sys.argv) > 3 else 300
eps_factor = float(sys.argv[4]) if len(sys.argv) > 4 else 1e6
# "tasks": the feet are the first level (12 task rows); "ttc": the feet are constraints::TaskToConstraint rows as in
# examples/cpp/coman_ik.cpp:437-442, (com-less variant) wrists / postural << limits << l_sole << r_sole << self-collision
mode = sys.argv[5] if len(sys.argv) > 5 else "tasks"
m = kin.humanoid32_pairs(kin.humanoid32())
n, P = m.n, len(m.pairs)
dev = torch.device("cuda", 0); f64 = dict(dtype=torch.float64, device=dev)
rng = np.random.default_rng(seed)
q0 = np.zeros((B, n))
q0[:, [m.names.index(s + "Elbj") for s in "RL"]] = -0.9
q0[:, m.names.index("RShLat")] = -0.35; q0[:, m.names.index("LShLat")] = 0.35
q0[:, [m.names.index(s + "KneeSag") for s in "RL"]] = 0.4
q0[:, [m.names.index(s + "HipSag") for s in "RL"]] = -0.2
q0[:, [m.names.index(s + "AnkSag") for s in "RL"]] = -0.2
q0 += rng.normal(0.0, 0.03, (B, n))
wrist = lambda nm: subtask(Task(abi.TASK_CARTESIAN, 6, lam=0.1, name=nm), [0, 1, 2])
sc = Rows(abi.ROWS_COLLISION, P, d_threshold=0.02, detection_threshold=0.0, bound_scaling=0.2, name="sc")
if mode == "ttc":
    levels = [[wrist("l_wrist"), wrist("r_wrist")], [Task(abi.TASK_POSTURAL, n, lam=0.01, name="postural")]]
    rowblocks = [Rows(abi.ROWS_TASK_CARTESIAN, 6, lam=0.1, name="l_sole"), Rows(abi.ROWS_TASK_CARTESIAN, 6, lam=0.1, name="r_sole"), sc]
else:
    levels = [[Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="l_sole"), Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="r_sole")],
              [wrist("l_wrist"), wrist("r_wrist")], [Task(abi.TASK_POSTURAL, n, lam=0.01, name="postural")]]
    rowblocks = [sc]
LW = 0 if mode == "ttc" else 1      # level of the wrist tasks
plan = StackPlan(n=n, levels=levels, bounds=[Bound(abi.BOUND_JOINT_LIMITS, scaling=1.0, name="jl"), Bound(abi.BOUND_VELOCITY_LIMITS, dT=0.01, name="vl")],
                 rowblocks=rowblocks, eps_abs=eps_abs_from_factor(eps_factor))
st = BatchedStack(plan, B, device=0, want_levels=False)
K = kin.Kinematics(m, device=0)
q = torch.as_tensor(q0, **f64).contiguous()
pose = [torch.zeros((B, 12), **f64) for _ in range(4)]
Jd = torch.zeros((B, P, n), **f64); dist = torch.zeros((B, P), **f64); Jw = torch.zeros((B, 12, n), **f64)


def fk():
    soles = {2: (st.C, 0), 3: (st.C, 6)} if mode == "ttc" else {2: (st.A[0], 0), 3: (st.A[0], 6)}
    K.forward(q, frame_pose={f: pose[f] for f in range(4)}, frame_J={**soles, 0: (Jw, 0), 1: (Jw, 6)}, pair_dist=dist, pair_J=(Jd, 0))
    st.A[LW][:B, 0:3].copy_(Jw[:, 0:3]); st.A[LW][:B, 3:6].copy_(Jw[:, 6:9])


fk(); torch.cuda.synchronize()
pose_d = [p.clone() for p in pose]
# wrist targets: random points in a box around the chest / pelvis / other hand -- many are inside the body
ctr = 0.5 * (pose[0][:, 9:] + pose[1][:, 9:])
for f in (0, 1):
    pose_d[f][:, 9:] = ctr + torch.as_tensor(rng.uniform([-0.25, -0.2, -0.2], [0.1, 0.2, 0.35], (B, 3)), **f64)
qmin = torch.full((B, n), -2.0, **f64); qmax = torch.full((B, n), 2.0, **f64)
q_ref = q.clone(); qdot_max = torch.full((B, n), 2.0, **f64)
wr = [(pose[0], pose_d[0], None), (pose[1], pose_d[1], None)]
so = [(pose[2], pose_d[2], None), (pose[3], pose_d[3], None)]
leaf = {"B": B, "task": ([wr, [(q, q_ref, None)]] if mode == "ttc" else [so, wr, [(q, q_ref, None)]]),
        "bound": [(q, qmin, qmax), (qdot_max, None, None)], "rows": (so + [(Jd, dist, None)] if mode == "ttc" else [(Jd, dist, None)])}
NL = len(levels)
nc = plan.nc
t0 = time.time()
solves = fails = bugs = shared = 0
bug_instances, shared_instances = set(), set()
dmin = np.inf
for cycle in range(cycles):
    fk(); st.update(leaf); st.solve(B); torch.cuda.synchronize()
    s = st.status[:B].cpu().numpy()
    solves += B
    dmin = min(dmin, float(dist.min()))
    bad = np.nonzero(s)[0]
    for i in bad[:8]:
        fails += 1
        asm = {"n": n, "B": 1, "L": NL, "eps_abs": plan.eps_abs, "m": [plan.m(k) for k in range(NL)], "ma": [plan.ma(k) for k in range(NL)],
               "A": [None if st.A[k] is None else st.A[k][i:i + 1].cpu().numpy() for k in range(NL)],
               "b": [st.b[k][i:i + 1].cpu().numpy() for k in range(NL)], "w": [st.w[k][i:i + 1].cpu().numpy() for k in range(NL)],
               "c": [None] * NL, "nc": nc, "C": st.C[i:i + 1].cpu().numpy(), "lo": st.lo[i:i + 1].cpu().numpy(), "up": st.up[i:i + 1].cpu().numpy(),
               "l": st.l[i:i + 1].cpu().numpy(), "u": st.u[i:i + 1].cpu().numpy()}
        rx = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1, termination_tolerance=10 * 2.221e-16)
        re_ = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
        if rx["status"][0] == 1 or re_["status"][0] == 1:
            first = int(i) not in bug_instances
            bugs += 1; bug_instances.add(int(i))
            print("BUG cycle", cycle, "instance", int(i), "status", int(s[i]), "witnesses", int(rx["status"][0]), int(re_["status"][0]), flush=True)
            if first and len(bug_instances) <= 6:
                np.savez(os.path.join(_ROOT, "gpurun_out", f"closed_loop_bug_{seed}_{len(bug_instances)}.npz"), **{k: v for k, v in asm.items() if isinstance(v, np.ndarray)},
                         **{f"A{k}": a for k, a in enumerate(asm["A"]) if a is not None}, **{f"b{k}": a for k, a in enumerate(asm["b"])},
                         **{f"w{k}": a for k, a in enumerate(asm["w"])})
        else:
            shared += 1; shared_instances.add(int(i))
    q += st.dq[:B]
print(f"seed {seed} eps_factor {eps_factor:g} feet as {mode}: {solves} closed-loop solves in {time.time() - t0:.0f} s, {fails} not solved (checked), "
      f"{shared} of them infeasible for the witnesses too ({len(shared_instances)} distinct instances), {bugs} product-only failures "
      f"({len(bug_instances)} distinct instances; a failed instance does not move, so it meets the same problem again); min pair distance seen {dmin:.4f}")
