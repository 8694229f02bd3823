"""TEST INFRASTRUCTURE (run by hand on the GPU box): closed-loop sweep of the reference's OWN robot and stacks -- COMAN, 35 coordinates,
S1 .. S4 of examples/cpp/coman_ik.cpp:425-449, the feet as TaskToConstraint rows -- i.e. of the 40-lane layout's null-space paths
(osot_qp_core.h: nullspace_equalities_wide, nullspace_dense_wide, lowrank_prepare<40>) under the drift of a real loop.  B robots chase
random wrist goals; (a) every instance the cascade does not solve is re-solved by the witnesses (qpOASES run to its exact optimum, the
eiQuadProg restatement): a failure they do not share is a product bug; (b) every `every` cycles a sample of the solved instances is
judged against the witnesses by the literal parity rule of the parity tests (tests/helpers.py:answer_is_acceptable).
usage: stress_closed_loop_coman.py SEED B CYCLES STACK [GOAL_RADIUS]"""
import os, sys, time
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))
import numpy as np, torch
import bench
from helpers import answer_is_acceptable
from opensot_amd import kinematics as kin
from opensot_amd.solver import BatchedStack
from oracle import pyoracle as oracle

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cycles = int(sys.argv[3]) if len(sys.argv) > 3 else 200
which = sys.argv[4] if len(sys.argv) > 4 else "S3"
radius = float(sys.argv[5]) if len(sys.argv) > 5 else 0.3
every, nsample = 20, 24
m, lo, up = kin.from_json(os.path.join(_ROOT, "tests", "golden", "coman_tree.json"))
n = m.n
plan = bench.coman_stack(which, n)
dev = torch.device("cuda", 0); f64 = dict(dtype=torch.float64, device=dev)
rng = np.random.default_rng(seed)
q0 = np.zeros((B, n))
for s_ in "RL":
    q0[:, m.names.index(s_ + "HipSag")] = -0.3; q0[:, m.names.index(s_ + "KneeSag")] = 0.6
    q0[:, m.names.index(s_ + "AnkSag")] = -0.3; q0[:, m.names.index(s_ + "Elbj")] = -0.8
    q0[:, m.names.index(s_ + "ShSag")] = 0.2
q0[:, m.names.index("LShLat")] = 0.3; q0[:, m.names.index("RShLat")] = -0.3
q0[:, 6:] += rng.normal(0.0, 0.03, (B, n - 6))
q0 = np.clip(q0, np.maximum(lo, -10.0) + 1e-3, np.minimum(up, 10.0) - 1e-3)
st = BatchedStack(plan, B, device=0, want_levels=False)
K = kin.Kinematics(m, device=0)
q = torch.as_tensor(q0, **f64).contiguous()
pose = [torch.zeros((B, 12), **f64) for _ in range(4)]
com = torch.zeros((B, 3), **f64)
where, off = {}, [0] * plan.L
for k, lev in enumerate(plan.levels):
    for t in lev:
        if t.name in ("l_wrist", "r_wrist", "com"):
            where[t.name] = (st.A[k], off[k])
        if not t.implicit:
            off[k] += t.rows
kw = dict(frame_pose={f: pose[f] for f in range(4)}, frame_J={0: where["l_wrist"], 1: where["r_wrist"], 2: (st.C, 0), 3: (st.C, 6)},
          com=com, com_J=where["com"])
K.forward(q, **kw); torch.cuda.synchronize()
pose_d = [p.clone() for p in pose]
for f in (0, 1):
    pose_d[f][:, 9:] += torch.as_tensor(rng.uniform(-radius, radius, (B, 3)), **f64)
big = 1.0e3
qmin = torch.as_tensor(np.tile(np.maximum(lo, -big), (B, 1)), **f64); qmax = torch.as_tensor(np.tile(np.minimum(up, big), (B, 1)), **f64)
leaf_of = {"l_wrist": (pose[0], pose_d[0], None), "r_wrist": (pose[1], pose_d[1], None), "com": (com, com.clone(), None), "postural": (q, q.clone(), None)}
leaf = {"B": B, "task": [[leaf_of[t.name] for t in lev] for lev in plan.levels],
        "bound": [(q, qmin, qmax), (torch.full((B, n), 2.0, **f64), None, None)], "rows": [(pose[2], pose_d[2], None), (pose[3], pose_d[3], None)]}
NL = plan.L


def asm_of(i):
    return {"n": n, "B": 1, "L": NL, "eps_abs": plan.eps_abs, "m": [plan.m(k) for k in range(NL)], "ma": [plan.ma(k) for k in range(NL)],
            "A": [None if st.A[k] is None else st.A[k][i:i + 1].cpu().numpy() for k in range(NL)],
            "b": [st.b[k][i:i + 1].cpu().numpy() for k in range(NL)], "w": [st.w[k][i:i + 1].cpu().numpy() for k in range(NL)],
            "c": [None] * NL, "nc": plan.nc, "C": st.C[i:i + 1].cpu().numpy(), "lo": st.lo[i:i + 1].cpu().numpy(), "up": st.up[i:i + 1].cpu().numpy(),
            "l": st.l[i:i + 1].cpu().numpy(), "u": st.u[i:i + 1].cpu().numpy()}


def witnesses(asm):
    rx = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1, termination_tolerance=10 * 2.221e-16) if oracle.ref_available() else None
    rq = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1) if oracle.ref_available() else None
    re_ = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
    return rx, rq, re_


t0 = time.time()
solves = fails = bugs = shared = judged = mismatches = 0
bug_instances, shared_instances = set(), set()
iters_sum = 0
for cycle in range(cycles):
    K.forward(q, **kw); st.update(leaf); st.solve(B); torch.cuda.synchronize()
    s = st.status[:B].cpu().numpy()
    solves += B
    for i in np.nonzero(s)[0][:8]:
        fails += 1
        rx, rq, re_ = witnesses(asm_of(int(i)))
        if (rx is not None and rx["status"][0] == 1) or re_["status"][0] == 1:
            bugs += 1; bug_instances.add(int(i))
            print("BUG cycle", cycle, "instance", int(i), "status", int(s[i]), flush=True)
        else:
            shared += 1; shared_instances.add(int(i))
    if cycle % every == every - 1:
        dq = st.dq[:B].cpu().numpy()
        for i in rng.choice(np.nonzero(s == 0)[0], size=min(nsample, int((s == 0).sum())), replace=False):
            asm = asm_of(int(i))
            rx, rq, re_ = witnesses(asm)
            wit = [("eiQuadProg", re_["dq"][0], re_["status"][0] == 1)]
            if rx is not None:
                wit += [("qpOASES exact", rx["dq"][0], rx["status"][0] == 1), ("qpOASES", rq["dq"][0], rq["status"][0] == 1)]
            ok, why = answer_is_acceptable(asm, 0, dq[i], wit)
            judged += 1
            if not ok:
                mismatches += 1
                print("MISMATCH cycle", cycle, "instance", int(i), why, flush=True)
                if mismatches <= 6:
                    os.makedirs(os.path.join(_ROOT, "gpurun_out", "coman_loop"), exist_ok=True)
                    np.savez(os.path.join(_ROOT, "gpurun_out", "coman_loop", f"mismatch_{which}_{seed}_{mismatches}.npz"),
                             **{k: v for k, v in asm.items() if isinstance(v, np.ndarray)}, **{f"A{k}": a for k, a in enumerate(asm["A"]) if a is not None},
                             **{f"b{k}": a for k, a in enumerate(asm["b"])}, **{f"w{k}": a for k, a in enumerate(asm["w"])}, dq_dev=dq[i],
                             **{"dq_" + nm.replace(" ", "_"): x for nm, x, okw in wit if okw})
    q += st.dq[:B]
print(f"COMAN35 {which} seed {seed} goals +-{radius} m: {solves} closed-loop solves in {time.time() - t0:.0f} s, {fails} not solved (checked), "
      f"{shared} of them infeasible for the witnesses too ({len(shared_instances)} distinct instances), {bugs} product-only failures; "
      f"{judged} solved instances judged against the witnesses: {mismatches} with a mismatch")
