"""shared test helpers (test infrastructure)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

from opensot_amd import abi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(cfg):
    """-> (plan, leaf, golden dict) from tests/golden/<cfg>_b32.npz (see tests/golden/make_golden.py)."""
    z = np.load(os.path.join(GOLDEN, f"{cfg}_b{16 if cfg == 'C5' else 32}.npz"), allow_pickle=False)
    B = int(z["B"])
    plan, tmpl = synth.make_id_stack(1, seed=0) if cfg == "C5" else synth.make_velocity_stack(cfg, 1, seed=0)
    plan.eps_abs = float(z["eps_abs"])

    def get(name):
        return z[name] if name in z.files else None

    leaf = {"B": B,
            "A": [get(f"leaf_A{k}") for k in range(plan.L)],
            "task": [[tuple(get(f"leaf_task{k}_{j}_p{i}") for i in range(3)) for j in range(len(plan.levels[k]))]
                     for k in range(plan.L)],
            "bound": [tuple(get(f"leaf_bound{j}_p{i}") for i in range(3)) for j in range(len(plan.bounds))],
            "rows": [tuple(get(f"leaf_rows{j}_p{i}") for i in range(3)) for j in range(len(plan.rowblocks))],
            "C": [get(f"leaf_C{j}") for j in range(len(plan.rowblocks))]}
    return plan, leaf, z


_emu = None


def emu_lib():
    """host lock-step emulation of the product kernels (tests/emu)."""
    global _emu
    if _emu is None:
        so = os.environ.get("OSOT_EMU_LIB") or os.path.join(ROOT, "tests", "emu", "libosot_emu.so")   # (developer knob: another build of the emulator)
        import glob
        srcs = (glob.glob(os.path.join(ROOT, "opensot_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
                + [f for f in glob.glob(os.path.join(ROOT, "tests", "emu", "**", "*"), recursive=True)
                   if os.path.isfile(f) and not f.endswith(".so")])
        if "OSOT_EMU_LIB" not in os.environ and (not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)):
            subprocess.check_call(["sh", os.path.join(ROOT, "tests", "emu", "build.sh")])
        L = C.CDLL(so)
        L.emu_ihqp_solve.argtypes = [C.POINTER(abi.PlanDesc), C.POINTER(abi.QpBatch), C.c_void_p, C.c_void_p]
        L.emu_stack_update.argtypes = [C.POINTER(abi.PlanDesc), C.POINTER(abi.LeafBatch), C.POINTER(abi.AssembledOut)]
        vp = C.c_void_p
        L.emu_qp_solve_batch.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp,
                                         C.c_double, C.c_int, vp, vp, vp]
        _emu = L
    return _emu


def emu_cascade(plan, asm, active=None, task_active=None, hot=None):
    """run the cascade kernel body on host pointers through the emulator.  task_active: {(level, task): bool}
    (Task::setActive); asm may carry "WA" / "Wb" (levels with a non-diagonal weight, see emu_update); hot: int32
    [B][L][32 or 64] hot-start state (read and rewritten in place; start from -1 everywhere), None = cold start"""
    B, n, L = asm["B"], asm["n"], asm["L"]
    qb = abi.QpBatch()
    qb.B = B
    keep = []
    for k in range(L):
        for name in ("A", "b", "w", "c", "WA", "Wb"):
            if name not in asm:
                continue
            a = asm[name][k]
            if a is not None:
                a = np.ascontiguousarray(a, dtype=np.float64)
                keep.append(a)
                getattr(qb, name)[k] = a.ctypes.data
    from opensot_amd.solver import stored_rows
    for name in ("C", "lo", "up", "l", "u"):
        a = asm[name]
        if a is not None:
            if name == "C":
                a = stored_rows(plan, a)
                if a.shape[1] == 0:
                    continue
            a = np.ascontiguousarray(a, dtype=np.float64)
            keep.append(a)
            setattr(qb, name, a.ctypes.data)
    if asm.get("reg") is not None:
        a = np.ascontiguousarray(asm["reg"]["b"], dtype=np.float64)
        keep.append(a)
        qb.b_reg = a.ctypes.data
        if asm["reg"].get("A") is not None:      # a regularisation task with a stored Jacobian (plan.regularisation_dense)
            a = np.ascontiguousarray(asm["reg"]["A"], dtype=np.float64)
            keep.append(a)
            qb.A_reg = a.ctypes.data
    dq = np.zeros((B, n)); xl = np.zeros((B, L, n))
    st = np.full(B, -1, dtype=np.int32); it = np.zeros(B, dtype=np.int32)
    qb.dq, qb.x_levels, qb.status, qb.iterations = dq.ctypes.data, xl.ctypes.data, st.ctypes.data, it.ctypes.data
    slack = np.zeros(B)
    qb.accepted_slack = slack.ctypes.data
    emu_cascade.last_accepted_slack = slack
    if active is not None:
        act = (C.c_ubyte * L)(*[1 if a else 0 for a in active])
        keep.append(act)
        qb.level_active = C.addressof(act)
    pd = plan.to_c()
    ta = None
    if task_active:
        ta = (C.c_ubyte * (abi.MAX_LEVELS * abi.MAX_TASKS))(*([1] * (abi.MAX_LEVELS * abi.MAX_TASKS)))
        for (k, j), on in task_active.items():
            ta[k * abi.MAX_TASKS + j] = 1 if on else 0
    if hot is not None:
        assert hot.dtype == np.int32 and hot.flags.c_contiguous and hot.shape == (B, L, 32 if n <= 32 else 64)
    rc = emu_lib().emu_ihqp_solve(C.byref(pd), C.byref(qb), C.cast(ta, C.c_void_p) if ta is not None else None,
                                  hot.ctypes.data if hot is not None else None)
    assert rc in (0, 100)        # (100: the BOX instantiation ran -- OSOT_EMU_BOX=1, see tests/emu/emu_driver.cpp)
    emu_cascade.ran_box = (rc == 100)
    return dq, xl, st, it


def emu_update(plan, leaf):
    """AutoStack::update (osot_update_kernel) on host arrays through the emulator: leaf dict (opensot_amd.synth layout) ->
    assembled dict in the oracle's layout (b, w, l, u, C [stored rows], lo, up, reg b, WA / Wb for dense-weight levels)"""
    B, n, L = leaf["B"], plan.n, plan.L
    keep = []

    def p(a):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dtype=np.float64)
        keep.append(a)
        return a.ctypes.data

    lb = abi.LeafBatch(); lb.B = B
    for k, lev in enumerate(leaf["task"]):
        for j, (p0, p1, p2) in enumerate(lev):
            lp = lb.task[k][j]
            lp.p0, lp.p1, lp.p2 = p(p0), p(p1), p(p2)
            if leaf.get("W") is not None:
                lp.W = p(leaf["W"][k][j])
    for j, (p0, p1, p2) in enumerate(leaf["bound"]):
        lb.bound[j].p0, lb.bound[j].p1, lb.bound[j].p2 = p(p0), p(p1), p(p2)
    for j, (p0, p1, p2) in enumerate(leaf["rows"]):
        lb.rows[j].p0, lb.rows[j].p1, lb.rows[j].p2 = p(p0), p(p1), p(p2)
    out = abi.AssembledOut()
    res = {"b": [np.zeros((B, plan.m(k))) for k in range(L)], "w": [np.ones((B, plan.m(k))) for k in range(L)],
           "WA": [np.zeros((B, plan.ma(k), n)) if plan.dense_level(k) and plan.ma(k) else None for k in range(L)],
           "Wb": [np.zeros((B, plan.m(k))) if plan.dense_level(k) else None for k in range(L)]}
    A = [np.ascontiguousarray(a, dtype=np.float64) if a is not None else None for a in leaf["A"]]
    for k in range(L):
        out.b[k], out.w[k] = res["b"][k].ctypes.data, res["w"][k].ctypes.data
        if res["Wb"][k] is not None:
            out.Wb[k] = res["Wb"][k].ctypes.data
            out.WA[k] = res["WA"][k].ctypes.data if res["WA"][k] is not None else None
            out.A[k] = A[k].ctypes.data if A[k] is not None else None
    nc, ncs = plan.nc, plan.nc_stored
    res["C"] = np.zeros((B, ncs, n)) if ncs else None
    for j, Cj in enumerate(leaf.get("C", [])):   # rows the producer writes in place
        if Cj is not None:
            o = plan.rows_stored_offset(j)
            res["C"][:, o:o + Cj.shape[1]] = Cj
    res["lo"] = np.zeros((B, nc)) if nc else None
    res["up"] = np.zeros((B, nc)) if nc else None
    res["l"] = np.zeros((B, n)) if plan.bounds else None
    res["u"] = np.zeros((B, n)) if plan.bounds else None
    for name in ("C", "lo", "up", "l", "u"):
        if res[name] is not None:
            setattr(out, name, res[name].ctypes.data)
    if plan.regularisation is not None:
        p0, p1, p2 = leaf["reg"]
        lb.regularisation.p0, lb.regularisation.p1, lb.regularisation.p2 = p(p0), p(p1), p(p2)
        res["b_reg"] = np.zeros((B, plan.regularisation.rows))
        out.b_reg = res["b_reg"].ctypes.data
    pd = plan.to_c()
    rc = emu_lib().emu_stack_update(C.byref(pd), C.byref(lb), C.byref(out))
    assert rc == 0
    return res


def emu_qp(H, g, A, lA, uA, l, u, eps_abs=0.0, max_iter=0):
    """B generic QPs through the emulated osot_qp_kernel; arrays are [B][...]."""
    H = np.ascontiguousarray(H, dtype=np.float64)
    B, n = H.shape[0], H.shape[1]
    nc = 0 if A is None else A.shape[1]
    arrs = [np.ascontiguousarray(a, dtype=np.float64) if a is not None else None for a in (g, A, lA, uA, l, u)]
    x = np.zeros((B, n)); st = np.full(B, -1, dtype=np.int32); it = np.zeros(B, dtype=np.int32)
    p = lambda a: None if a is None else a.ctypes.data
    rc = emu_lib().emu_qp_solve_batch(B, n, nc, p(H), *[p(a) for a in arrs], eps_abs, max_iter, p(x), p(st), p(it))
    assert rc == 0
    return x, st, it


def random_qp(rng, B, n, nc, n_eq=0, box=True, scale=1.0):
    """strictly convex random QPs with x = 0 strictly feasible for the inequalities."""
    M = rng.normal(size=(B, n + 3, n))
    H = np.einsum("bki,bkj->bij", M, M) * scale
    g = rng.normal(size=(B, n)) * 3.0 * scale
    A = rng.normal(size=(B, nc, n)) if nc else None
    lA = uA = None
    if nc:
        lA = -rng.uniform(0.05, 1.0, size=(B, nc))
        uA = rng.uniform(0.05, 1.0, size=(B, nc))
        if n_eq:
            v = rng.normal(size=(B, n_eq)) * 0.1
            lA[:, :n_eq] = v; uA[:, :n_eq] = v
        # a few one-sided rows
        lA[:, -1] = -np.inf
    l = u = None
    if box:
        l = -rng.uniform(0.05, 0.5, size=(B, n)); u = rng.uniform(0.05, 0.5, size=(B, n))
    return H, g, A, lA, uA, l, u


def kkt_check(H, g, A, lA, uA, l, u, x, eps_abs, tol=1e-7):
    """primal feasibility + existence of multipliers with the right signs (least squares on the active set)."""
    n = x.shape[0]
    Hr = H + eps_abs * np.eye(n)
    grad = Hr @ x + g
    normals, signs = [], []
    if l is not None:
        assert (x >= l - tol).all() and (x <= u + tol).all()
        for i in range(n):
            if abs(x[i] - l[i]) <= tol:
                e = np.zeros(n); e[i] = 1; normals.append(e); signs.append(+1 if l[i] < u[i] else 0)
            elif abs(x[i] - u[i]) <= tol:
                e = np.zeros(n); e[i] = -1; normals.append(e); signs.append(+1)
    if A is not None and A.shape[0]:
        ax = A @ x
        assert (ax >= np.maximum(lA, -1e20) - tol).all() and (ax <= np.minimum(uA, 1e20) + tol).all()
        for r in range(A.shape[0]):
            if lA[r] == uA[r]:
                normals.append(A[r]); signs.append(0)
            elif abs(ax[r] - lA[r]) <= tol:
                normals.append(A[r]); signs.append(+1)
            elif abs(ax[r] - uA[r]) <= tol:
                normals.append(-A[r]); signs.append(+1)
    if not normals:
        return np.abs(grad).max()
    N = np.array(normals).T
    lam, *_ = np.linalg.lstsq(N, grad, rcond=None)
    resid = np.abs(N @ lam - grad).max()
    for s, v in zip(signs, lam):
        if s > 0:
            assert v >= -1e-6 * (1 + np.abs(lam).max()), "negative multiplier on an active inequality"
    return resid


def emu_kinematics(model, q, env_pose=None):
    """the kinematics kernel body on host arrays through the emulator: q [B][n] -> poses, frame Jacobians, com, Jcom"""
    L = emu_lib()
    L.emu_kinematics.argtypes = [C.POINTER(abi.KinDesc), C.POINTER(abi.KinBatch)]
    q = np.ascontiguousarray(q, dtype=np.float64)
    B, n = q.shape
    F = len(model.frames)
    d = model.desc()
    kb = abi.KinBatch()
    kb.B = B
    kb.q = q.ctypes.data
    poses = [np.zeros((B, 12)) for _ in range(F)]
    J = np.full((B, 6 * F + 3, n), 7.0)
    com = np.zeros((B, 3))
    for f in range(F):
        kb.frame_pose[f] = poses[f].ctypes.data
        kb.frame_J[f] = J.ctypes.data + 8 * 6 * f * n
        kb.frame_J_stride[f] = (6 * F + 3) * n
    kb.com = com.ctypes.data
    kb.com_J = J.ctypes.data + 8 * 6 * F * n
    kb.com_J_stride = (6 * F + 3) * n
    P = len(getattr(model, "pairs", []))
    if P:
        pd = np.zeros((B, P)); pJ = np.full((B, P + 1, n), 7.0)
        kb.pair_dist = pd.ctypes.data; kb.pair_J = pJ.ctypes.data; kb.pair_J_stride = (P + 1) * n
        if env_pose is None and getattr(model, "env_shapes", None):
            env_pose = model.env_pose_array()
        if env_pose is not None and np.asarray(env_pose).size:
            env_pose = np.ascontiguousarray(env_pose, dtype=np.float64)
            kb.env_pose = env_pose.ctypes.data
            kb.env_pose_stride = env_pose.shape[-2] * 12 if env_pose.ndim == 3 else 0
    assert L.emu_kinematics(C.byref(d), C.byref(kb)) == 0
    if P:
        return poses, J, com, pd, pJ
    return poses, J, com


def closed_loop_plan(mode="tasks", eps_factor=1e6):
    """the stack of tests/stress_closed_loop.py on the 32-DoF humanoid: feet (first level, or TaskToConstraint rows as in
    examples/cpp/coman_ik.cpp:437-442) / wrist positions / Postural << joint limits << velocity limits << 16 capsule pairs"""
    from opensot_amd import kinematics as kin
    from opensot_amd.plan import StackPlan, Task, Bound, Rows, subtask, eps_abs_from_factor
    m = kin.humanoid32_pairs(kin.humanoid32())
    n, P = m.n, len(m.pairs)
    wrist = lambda nm: subtask(Task(abi.TASK_CARTESIAN, 6, lam=0.1, name=nm), [0, 1, 2])
    sc = Rows(abi.ROWS_COLLISION, P, d_threshold=0.02, detection_threshold=0.0, bound_scaling=0.2, name="sc")
    if mode == "ttc":
        levels = [[wrist("l_wrist"), wrist("r_wrist")], [Task(abi.TASK_POSTURAL, n, lam=0.01, name="postural")]]
        rowblocks = [Rows(abi.ROWS_TASK_CARTESIAN, 6, lam=0.1, name="l_sole"), Rows(abi.ROWS_TASK_CARTESIAN, 6, lam=0.1, name="r_sole"), sc]
    else:
        levels = [[Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="l_sole"), Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="r_sole")],
                  [wrist("l_wrist"), wrist("r_wrist")], [Task(abi.TASK_POSTURAL, n, lam=0.01, name="postural")]]
        rowblocks = [sc]
    plan = StackPlan(n=n, levels=levels, bounds=[Bound(abi.BOUND_JOINT_LIMITS, scaling=1.0, name="jl"), Bound(abi.BOUND_VELOCITY_LIMITS, dT=0.01, name="vl")],
                     rowblocks=rowblocks, eps_abs=eps_abs_from_factor(eps_factor))
    return plan, m


def default_eps_stuck_instances(mode="tasks"):
    """instances met by tests/stress_closed_loop.py at iHQP's DEFAULT eps factor 2e2 (iHQP.h:32) that the kernel reported
    INFEASIBLE while a witness went on -- kept as data (assembled arrays of the cycle).  mode "tasks": five instances of
    seeds 4 and 7 (round 1) and a sixth of seed 7 that only the hardware failed with a ratio tolerance of 1e-10; "ttc": one of seed 21 with the feet as TaskToConstraint rows (a noise-level "positive" entry
    of the dual direction gave a dual step of 6.6e8: kRatioTol in osot_qp_core.h)"""
    # "ttc_exchange" (round 5, seed 44 of the sweep, HARDWARE only): a bound 3.5e-9 outside with its normal in the span of the working set
    # was exchanged along a barely independent direction and the level ended INFEASIBLE (kSpanAccept in osot_qp_core.h)
    files = {"tasks": "default_eps_stuck_instances.npz", "ttc": "default_eps_stuck_ttc_instance.npz", "ttc_exchange": "default_eps_roundoff_exchange_instance.npz"}
    plan, _ = closed_loop_plan("tasks" if mode == "tasks" else "ttc", 2e2)
    z = np.load(os.path.join(GOLDEN, files[mode]))
    B = z["b0"].shape[0]
    L = plan.L
    asm = {"n": plan.n, "B": B, "L": L, "eps_abs": plan.eps_abs, "m": [plan.m(k) for k in range(L)], "ma": [plan.ma(k) for k in range(L)],
           "A": [z[f"A{k}"] if f"A{k}" in z.files else None for k in range(L)], "b": [z[f"b{k}"] for k in range(L)],
           "w": [z[f"w{k}"] for k in range(L)], "c": [None] * L,
           "nc": plan.nc, "C": z["C"], "lo": z["lo"], "up": z["up"], "l": z["l"], "u": z["u"]}
    return plan, asm


def accepted_slack_instance():
    """the one instance of round 4's randomised sweep that failed the literal acceptance rule on hardware (profiles/r04_stress_parity.txt:
    the kernel accepted a 4.2e-7 violation of a global inequality row as round-off of the levels above at the DEFAULT eps) with the three
    witnesses' answers -- tests/golden/default_eps_accepted_slack_instance.npz, built by tests/golden/make_accepted_slack_fixture.py.
    -> plan, asm, [(name, dq, solved)]"""
    sys.path.insert(0, GOLDEN)
    import make_accepted_slack_fixture as mk
    plan = mk.plan_of()
    z = np.load(os.path.join(GOLDEN, "default_eps_accepted_slack_instance.npz"))
    L = plan.L
    asm = {"n": plan.n, "B": 1, "L": L, "eps_abs": plan.eps_abs, "m": [plan.m(k) for k in range(L)], "ma": [plan.ma(k) for k in range(L)],
           "A": [z[f"A{k}"] if f"A{k}" in z.files else None for k in range(L)], "b": [z[f"b{k}"] for k in range(L)],
           "w": [z[f"w{k}"] for k in range(L)], "c": [None] * L,
           "nc": plan.nc, "C": z["C"], "lo": z["lo"], "up": z["up"], "l": None, "u": None}
    wit = [("qpOASES", z["dq_qpoases"][0], bool(z["ok_qpoases"][0])), ("qpOASES exact", z["dq_qpoases_exact"][0], bool(z["ok_qpoases_exact"][0])),
           ("eiQuadProg", z["dq_eiquadprog"][0], bool(z["ok_eiquadprog"][0]))]
    return plan, asm, wit


def answer_is_acceptable(asm, i, dq_dev, witnesses, tol=1e-6, feas_tol=1e-7, active=None):
    """Is the device's final point for instance i an answer to the reference's problem (iHQP.cpp:263-358)?  Yes if it is
    within `tol` of a witness -- or, where the witnesses themselves disagree (ill-conditioned levels: qpOASES at OpenSoT's
    options stops up to 2e-2 from the optimum with its constraints violated by 2e-7), if it is FEASIBLE to `feas_tol` (1e-7:
    half of qpOASES' own terminationTolerance under OpenSoT's options, 2.2e-7; the kernel reports what it accepted in
    osot_qp_batch.accepted_slack) and
    its lexicographic cost vector (oracle/lexcheck.py) is not worse than that of any witness that is as feasible as it is.
    witnesses: list of (name, dq, solved).  Returns (ok, why)."""
    from oracle import lexcheck as lc
    solved = [(nm, x) for nm, x, ok in witnesses if ok]
    if not solved:
        # nobody to compare with: the point has to carry its own certificate -- feasible, and a KKT point of every level with
        # non-negative multipliers (oracle/lexcheck.py:kkt_certificate; VERDICT r2: this branch used to return True)
        gv = lc.global_violation(asm, i, dq_dev)
        if gv > feas_tol:
            return False, f"no witness solved it and the device point is infeasible by {gv:.1e}"
        for c in lc.kkt_certificate(asm, i, dq_dev, active):
            if c["viol"] > feas_tol or c["kkt"] > c["kkt_allowed"]:
                return False, (f"no witness solved it; level {c['level']}: violation {c['viol']:.1e}, KKT residual {c['kkt']:.1e} "
                               f"(allowed {c['kkt_allowed']:.1e})")
        return True, f"no witness; feasible to {gv:.1e} and a KKT point of every level with non-negative multipliers"
    d = min(np.abs(dq_dev - x).max() for _, x in solved)
    if d <= tol:
        return True, f"within {d:.1e} of a witness"
    gv = lc.global_violation(asm, i, dq_dev)
    if gv > feas_tol:
        return False, f"{d:.1e} from the closest witness and infeasible by {gv:.1e}"
    cd = lc.lex_costs(asm, i, dq_dev, active)
    # a level's cost is resolved to about cond(H) * unit round-off = (|A'WA| / eps) * 1.1e-16 in relative terms (1e-9 at
    # the benchmark's eps factor 1e6, 5e-5 at iHQP's default 2e2): costs closer than that are a tie
    rtol = max(1e-9, 10 * 2.2e-16 / asm["eps_abs"])
    for nm, x in solved:
        gw = lc.global_violation(asm, i, x)
        if gw > max(10 * gv, 1e-12):
            continue      # that witness bought its cost with a constraint violation the device point does not have (at the
                          # default eps a violation of 3e-9 buys 10 % of the Postural level's cost on these instances)
        if lc.lex_compare(cd, lc.lex_costs(asm, i, x, active), rtol=rtol, atol=1e-13, rtol_better=1e-9) > 0:
            return False, f"{d:.1e} from the closest witness; lexicographically worse than {nm} (device {cd}, violation {gv:.1e})"
    return True, f"{d:.1e} from the closest witness, feasible to {gv:.1e} and lexicographically not worse than any as-feasible witness"


def nhqp_fill_options(opt, free_vars=None, min_sv_ratio=None, ab_regularization=True, selective_ns_regularization=True):
    """osot_nhqp_options from the reference's setters: scalars are the solver-wide form, lists (one entry per level) the per-level one
    (nHQP::setPerformAbRegularization(level, .), setPerformSelectiveNullSpaceRegularization(level, .), setMinSingularValueRatio(vector))"""
    if free_vars is not None:
        for k, v in enumerate(free_vars):
            opt.free_vars[k] = int(v)
    if isinstance(min_sv_ratio, (list, tuple)):
        for k, v in enumerate(min_sv_ratio):
            if v is not None:
                opt.level_min_sv_ratio[k] = float(v); opt.level_min_sv_ratio_is_set[k] = 1
    elif min_sv_ratio is not None:
        opt.min_sv_ratio = min_sv_ratio
        opt.min_sv_ratio_is_set = 1
    if isinstance(ab_regularization, (list, tuple)):
        for k, v in enumerate(ab_regularization):
            opt.level_no_ab_regularization[k] = 0 if v else 1
    else:
        opt.no_ab_regularization = 0 if ab_regularization else 1
    if isinstance(selective_ns_regularization, (list, tuple)):
        for k, v in enumerate(selective_ns_regularization):
            opt.level_no_selective_ns_regularization[k] = 0 if v else 1
    else:
        opt.no_selective_ns_regularization = 0 if selective_ns_regularization else 1
    return opt


def emu_nhqp(plan, asm, free_vars=None, min_sv_ratio=None, ab_regularization=True, selective_ns_regularization=True, task_active=None, level_W=None):
    """the null-space front-end (osot_nhqp_*.h: kernels AND host orchestration) on host pointers through the emulator.
    task_active: {(level, task): bool} (Task::setActive)"""
    B, n, L = asm["B"], asm["n"], asm["L"]
    qb = abi.QpBatch()
    qb.B = B
    keep = []
    for k in range(L):
        for name in ("A", "b", "w"):
            a = asm[name][k]
            if a is not None:
                a = np.ascontiguousarray(a, dtype=np.float64); keep.append(a)
                getattr(qb, name)[k] = a.ctypes.data
    for name in ("C", "lo", "up", "l", "u"):
        a = asm[name]
        if a is not None and a.size:
            a = np.ascontiguousarray(a, dtype=np.float64); keep.append(a)
            setattr(qb, name, a.ctypes.data)
    dq = np.zeros((B, n)); st = np.full(B, -1, dtype=np.int32)
    qb.dq, qb.status = dq.ctypes.data, st.ctypes.data
    opt = nhqp_fill_options(abi.NhqpOptions(), free_vars, min_sv_ratio, ab_regularization, selective_ns_regularization)
    if level_W is not None:        # osot_nhqp_options.level_W: the full weight matrix of a level with a non-diagonal weight (host pointers here)
        for k, Wk in enumerate(level_W):
            if Wk is not None:
                Wk = np.ascontiguousarray(Wk, dtype=np.float64); keep.append(Wk)
                opt.level_W[k] = Wk.ctypes.data
    ta = None
    if task_active:
        ta = np.ones(abi.MAX_LEVELS * abi.MAX_TASKS, dtype=np.uint8)
        for (k, j), on in task_active.items():
            ta[k * abi.MAX_TASKS + j] = 1 if on else 0
    L_ = emu_lib()
    L_.emu_nhqp_solve.argtypes = [C.POINTER(abi.PlanDesc), C.POINTER(abi.QpBatch), C.POINTER(abi.NhqpOptions), C.c_void_p]
    pd = plan.to_c()
    rc = L_.emu_nhqp_solve(C.byref(pd), C.byref(qb), C.byref(opt), None if ta is None else ta.ctypes.data)
    assert rc == 0
    return dq, st


def emu_ehqp(plan, asm, sigma_min=0.0, level_active=None, task_active=None):
    """the equality-only front-end (osot_ehqp.h) on host pointers through the emulator -> dq, status, x_levels"""
    B, n, L = asm["B"], asm["n"], asm["L"]
    qb = abi.QpBatch()
    qb.B = B
    keep = []
    for k in range(L):
        for name in ("A", "b", "w"):
            a = asm[name][k]
            if a is not None:
                a = np.ascontiguousarray(a, dtype=np.float64); keep.append(a)
                getattr(qb, name)[k] = a.ctypes.data
        for name, key in (("WA", "WA"), ("Wb", "Wb")):
            arr = asm.get(key)
            if arr is not None and arr[k] is not None:
                a = np.ascontiguousarray(arr[k], dtype=np.float64); keep.append(a)
                getattr(qb, name)[k] = a.ctypes.data
    dq = np.zeros((B, n)); st = np.full(B, -1, dtype=np.int32); xl = np.zeros((B, L, n))
    qb.dq, qb.status, qb.x_levels = dq.ctypes.data, st.ctypes.data, xl.ctypes.data
    if level_active is not None:
        la = np.ascontiguousarray(level_active, dtype=np.uint8); keep.append(la)
        qb.level_active = la.ctypes.data
    L_ = emu_lib()
    L_.emu_ehqp_solve.argtypes = [C.POINTER(abi.PlanDesc), C.POINTER(abi.QpBatch), C.c_double, C.c_void_p]
    ta = None
    if task_active is not None:      # {(level, task): False} -> the [MAX_LEVELS * MAX_TASKS] flag array of osot_solver_set_task_active
        ta = np.ones(abi.MAX_LEVELS * abi.MAX_TASKS, dtype=np.uint8)
        for (k, j), on in task_active.items():
            ta[k * abi.MAX_TASKS + j] = 1 if on else 0
        keep.append(ta)
    pd = plan.to_c()
    rc = L_.emu_ehqp_solve(C.byref(pd), C.byref(qb), float(sigma_min), None if ta is None else ta.ctypes.data)
    assert rc == 0
    return dq, st, xl


def emu_qp_admm(H, g, A, lA, uA, l, u, eps_reg=0.0, max_iter=0, scaling=0, warm=None):
    """B generic QPs through the emulated OSQP-convention ADMM kernel (osot_admm.h); arrays are [B][...].  scaling as
    osot_admm_options.scaling (0 = ten Ruiz passes, negative = none); warm: dict(x [B][n], y [B][nc + n or nc], rho [B]),
    read and rewritten in place (start with rho = 0: no state)"""
    H = np.ascontiguousarray(H, dtype=np.float64)
    B, n = H.shape[0], H.shape[1]
    nc = 0 if A is None else A.shape[1]
    arrs = [np.ascontiguousarray(a, dtype=np.float64) if a is not None else None for a in (g, A, lA, uA, l, u)]
    x = np.zeros((B, n)); st = np.full(B, -1, dtype=np.int32); it = np.zeros(B, dtype=np.int32)
    p = lambda a: None if a is None else a.ctypes.data
    L = emu_lib()
    vp = C.c_void_p
    L.emu_qp_solve_batch_admm.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_double, C.c_int, vp, vp, vp,
                                          C.c_int, vp, vp, vp]
    wx = wy = wr = None
    if warm is not None:
        wx, wy, wr = warm["x"], warm["y"], warm["rho"]
        assert wx.flags.c_contiguous and wy.flags.c_contiguous and wr.flags.c_contiguous and wx.dtype == wy.dtype == wr.dtype == np.float64
    assert L.emu_qp_solve_batch_admm(B, n, nc, p(H), *[p(a) for a in arrs], eps_reg, max_iter, p(x), p(st), p(it),
                                     scaling, p(wx), p(wy), p(wr)) == 0
    return x, st, it


def parity_census(asm, dq_dev, witnesses, tol=1e-6, feas_tol=1e-7, active=None, label=""):
    """every instance against the witnesses (list of (name, result dict of pyoracle.ihqp_solve_batch)), the FIRST one being the
    parity target (the reference's qpOASES at OpenSoT's options): absolute max-norm distance `tol` (north_star: 1e-6, no
    scaling by |dq|), and for an instance beyond it the literal acceptance rule (answer_is_acceptable: feasible and
    lexicographically not worse than every as-feasible witness, or -- no witness solved it -- its own KKT certificate).
    Prints the census (nothing passes silently) and returns (n_within_tol_of_first, n_accepted_by_rule, failures)."""
    B = asm["B"]
    first = witnesses[0][1]
    within = rule = 0
    fails = []
    worst = 0.0
    for i in range(B):
        w = [(nm, r["dq"][i], r["status"][i] == 1) for nm, r in witnesses]
        if w[0][2] and np.abs(dq_dev[i] - w[0][1]).max() <= tol:
            within += 1
            continue
        if w[0][2]:
            worst = max(worst, float(np.abs(dq_dev[i] - w[0][1]).max()))
        ok, why = answer_is_acceptable(asm, i, dq_dev[i], w, tol=tol, feas_tol=feas_tol, active=active)
        if ok:
            rule += 1
        else:
            fails.append((i, why))
    print(f"[parity census{' ' + label if label else ''}] {B} instances: {within} within {tol:g} (absolute) of {witnesses[0][0]}, "
          f"{int((first['status'] == 1).sum())} solved by it; {rule} accepted by the feasibility + lexicographic rule "
          f"(farthest from {witnesses[0][0]}: {worst:.2e}); {len(fails)} not acceptable" + (f": {fails[:3]}" if fails else ""))
    return within, rule, fails


def judge_remainder(asm, dq_dev, results, active=None, tol=1e-6, feas_tol=1e-7, label=""):
    """The tests compare the device with each witness WHERE THAT WITNESS SOLVED the instance; this judges the rest, so that no
    device answer goes unchecked: an instance that at least one of the witnesses in `results` (name -> result dict
    of pyoracle.ihqp_solve_batch) did not solve must pass the literal acceptance rule against all of them
    (answer_is_acceptable: within `tol` of a witness that did solve it, or feasible and lexicographically not worse, or -- when
    nobody solved it -- feasible to `feas_tol` with its own KKT certificate).  Asserts; returns the number judged."""
    B = asm["B"]
    names = list(results)
    todo = [i for i in range(B) if any(results[nm]["status"][i] != 1 for nm in names)]
    fails = []
    for i in todo:
        w = [(nm, results[nm]["dq"][i], results[nm]["status"][i] == 1) for nm in names]
        ok, why = answer_is_acceptable(asm, i, dq_dev[i], w, tol=tol, feas_tol=feas_tol, active=active)
        if not ok:
            fails.append((i, why))
    print(f"[remainder{' ' + label if label else ''}] {len(todo)} of {B} instances were not solved by every witness "
          f"({', '.join(names)}); judged by the acceptance rule: {len(todo) - len(fails)} accepted, {len(fails)} not" + (f": {fails[:3]}" if fails else ""))
    assert not fails, fails[:5]
    return len(todo)


def max_where(mask, values):
    """max of values[mask], 0 when the mask is empty (the remainder is judge_remainder's)"""
    return float(values[mask].max()) if np.any(mask) else 0.0


_big = None


def big_host_solve(H, g, A, lA, uA, l, u, eps_abs, max_iter=0):
    """one QP through the WIDE solver (opensot_amd/csrc/osot_qp_big.h: 65 .. 128 variables) compiled for the host with a team of one
    thread (tests/emu/big_host.cpp).  Returns (status, x, iterations)."""
    global _big
    if _big is None:
        so = os.path.join(ROOT, "tests", "emu", "libosot_big_host.so")
        srcs = [os.path.join(ROOT, "opensot_amd", "csrc", "osot_qp_big.h"), os.path.join(ROOT, "tests", "emu", "big_host.cpp")]
        if not os.path.exists(so) or any(os.path.getmtime(f) > os.path.getmtime(so) for f in srcs):
            subprocess.check_call(["sh", os.path.join(ROOT, "tests", "emu", "build.sh")])
        _big = C.CDLL(so)
    dp = C.POINTER(C.c_double)
    n = H.shape[0]
    nc = 0 if A is None else A.shape[0]
    keep = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (H, g, A, lA, uA, l, u)]
    ptr = [None if a is None else a.ctypes.data_as(dp) for a in keep]
    x = np.zeros(n)
    st, it = C.c_int(-1), C.c_int(0)
    rc = _big.osot_big_host_solve(n, nc, *ptr, C.c_double(eps_abs), int(max_iter), x.ctypes.data_as(dp), C.byref(st), C.byref(it))
    assert rc == 0, "osot_big_host_solve refused the sizes"
    return st.value, x, it.value


from opensot_amd.synth import wide_id_levels as id_like_levels      # (the generator lives with the other synthetic workloads)


_refqp = None


def ref_qpoases_solve(H, g, A, lA, uA, l, u, eps_factor, exact=False):
    """one QP through the REFERENCE's own qpOASES 3.1 (oracle/_ref/libqpoases_ref.so, compiled in place from /root/reference by
    oracle/Makefile; QPOasesBackEnd.cpp:51-76 option set, absolute eps = 2.221e-13 * eps_factor), cold-initialised.  exact: its
    termination tolerance tightened to 1e-12 (the reference's MPC option set stops at 1e-7 relative).  Returns (ok, x) or None
    where the reference build is not present."""
    global _refqp
    so = os.path.join(ROOT, "oracle", "_ref", "libqpoases_ref.so")
    if not os.path.exists(so):
        return None
    if _refqp is None:
        _refqp = C.CDLL(so)
        _refqp.refqp_create.restype = C.c_void_p
        _refqp.refqp_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
        vp, dp = C.c_void_p, C.POINTER(C.c_double)
        _refqp.refqp_init.argtypes = [vp, dp, dp, dp, dp, dp, dp, dp]
        _refqp.refqp_get_solution.argtypes = [vp, dp]
        _refqp.refqp_destroy.argtypes = [vp]
    dp = C.POINTER(C.c_double)
    n = H.shape[0]
    nc = 0 if A is None else A.shape[0]
    keep = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (H, g, A, lA, uA, l, u)]
    ptr = [None if a is None else a.ctypes.data_as(dp) for a in keep]
    h = _refqp.refqp_create(n, nc, abi.HST_UNKNOWN, float(eps_factor), 1.0e-12 if exact else -1.0)
    ok = _refqp.refqp_init(h, *ptr)
    x = np.zeros(n)
    _refqp.refqp_get_solution(h, x.ctypes.data_as(dp))
    _refqp.refqp_destroy(h)
    return bool(ok), x
