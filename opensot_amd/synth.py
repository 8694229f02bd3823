"""Seeded synthetic humanoid stacks for the BASELINE.json configurations.

Produces (StackPlan, leaf) where `leaf` holds the per-instance inputs that the reference obtains from
XBot::ModelInterface (Jacobians, poses, CoM, q) plus references and limits, as numpy fp64 arrays.
No solving happens here.  There is no robot model on this path (xbot2_interface is un-vendored):
Jacobians are N(0, 0.3^2) with kinematic-chain sparsity, poses are random rigid transforms.

leaf layout (all instance-major):
  leaf["A"][k]            : [B][ma_k][n]   stacked task Jacobians of level k (Postural rows implicit)
  leaf["task"][k][j]      : (p0, p1, p2)   per osot_leaf_ptrs in include/osot_mi355x.h
  leaf["bound"][j]        : (p0, p1, p2)
  leaf["rows"][j]         : (p0, p1, p2)
"""
import numpy as np

from . import abi
from .plan import Bound, Rows, StackPlan, Task, eps_abs_from_factor

# 32-DoF humanoid column map: 6 floating-base + 7 + 7 (arms) + 6 + 6 (legs)
_BASE = list(range(0, 6))
_LIMBS = {
    "l_arm": list(range(6, 13)),
    "r_arm": list(range(13, 20)),
    "l_leg": list(range(20, 26)),
    "r_leg": list(range(26, 32)),
}


def _rot_exp(w):
    """Rodrigues: exp([w]x) for w [..., 3] -> [..., 3, 3]."""
    th = np.linalg.norm(w, axis=-1, keepdims=True)
    th = np.where(th < 1e-12, 1e-12, th)
    k = w / th
    K = np.zeros(w.shape[:-1] + (3, 3))
    K[..., 0, 1], K[..., 0, 2] = -k[..., 2], k[..., 1]
    K[..., 1, 0], K[..., 1, 2] = k[..., 2], -k[..., 0]
    K[..., 2, 0], K[..., 2, 1] = -k[..., 1], k[..., 0]
    s, c = np.sin(th)[..., None], np.cos(th)[..., None]
    return np.eye(3) + s * K + (1 - c) * (K @ K)


def _pose(R, p):
    return np.concatenate([R.reshape(R.shape[0], 9), p], axis=1)


def _limb_jacobian(rng, B, rows, n, cols):
    J = np.zeros((B, rows, n))
    J[:, :, cols] = rng.normal(0.0, 0.3, size=(B, rows, len(cols)))
    return J


def _cartesian_leaf(rng, B):
    R = _rot_exp(rng.normal(0.0, 0.2, size=(B, 3)))
    p = rng.uniform(-1.0, 1.0, size=(B, 3))
    dth = rng.normal(size=(B, 3))
    dth *= (rng.uniform(0.0, 0.1, size=(B, 1)) / np.linalg.norm(dth, axis=1, keepdims=True))
    dp = rng.normal(size=(B, 3))
    dp *= (rng.uniform(0.0, 0.05, size=(B, 1)) / np.linalg.norm(dp, axis=1, keepdims=True))
    Rd = R @ _rot_exp(dth)
    pd = p + dp
    return _pose(R, p), _pose(Rd, pd), None


def _box_leaf(rng, B, n, jl=True, vl=True):
    bounds, leaf = [], []
    if jl:
        half = rng.uniform(0.5, 2.5, size=(B, n))
        qmin, qmax = -half, half
        q = rng.uniform(qmin, qmax)
        # some joints sit close to a limit so that the joint-limit side of the box binds
        near = rng.random((B, n)) < 0.1
        q = np.where(near, qmax - rng.uniform(0.0, 0.01, size=(B, n)), q)
        bounds.append(Bound(abi.BOUND_JOINT_LIMITS, scaling=1.0, name="joint_limits"))
        leaf.append((q, qmin, qmax))
    if vl:
        bounds.append(Bound(abi.BOUND_VELOCITY_LIMITS, dT=0.01, name="velocity_limits"))
        leaf.append((np.full((B, n), 2.0), None, None))
    return bounds, leaf


def make_velocity_stack(config, B, seed=None, n=32, eps_factor=1e6, P=16):
    """config in {"C2", "C3", "C4"}; returns (plan, leaf)."""
    assert n == 32, "the synthetic humanoid column map is 32-DoF"
    cfg_id = {"C2": 2, "C3": 3, "C4": 4}[config]
    rng = np.random.default_rng(1000 * cfg_id if seed is None else seed)

    def cart(name, limb, weight=1.0, lam=0.1):
        t = Task(abi.TASK_CARTESIAN, 6, weight=weight, lam=lam, name=name)
        J = _limb_jacobian(rng, B, 6, n, _BASE + _LIMBS[limb])
        return t, J, _cartesian_leaf(rng, B)

    def com(lam=0.1):
        t = Task(abi.TASK_COM, 3, lam=lam, name="com")
        J = rng.normal(0.0, 0.3, size=(B, 3, n))
        p = rng.uniform(-0.2, 0.2, size=(B, 3))
        pd = p + rng.uniform(-0.05, 0.05, size=(B, 3))
        return t, J, (p, pd, None)

    def postural(weight=1.0, lam=0.01, q=None):
        t = Task(abi.TASK_POSTURAL, n, weight=weight, lam=lam, name="postural")
        qq = rng.uniform(-1.0, 1.0, size=(B, n)) if q is None else q
        qd = qq + rng.normal(0.0, 0.1, size=(B, n))
        return t, None, (qq, qd, None)

    if config == "C2":
        # one level, soft priorities (cf. coman_ik.cpp:429): r_wrist + 1e-4*postural, joint-limit box
        bounds, bleaf = _box_leaf(rng, B, n, jl=True, vl=False)
        blocks = [[cart("r_wrist", "r_arm", lam=1.0), postural(weight=1e-4, lam=1.0, q=bleaf[0][0])]]
        rowblocks, rleaf = [], []
    else:
        bounds, bleaf = _box_leaf(rng, B, n, jl=True, vl=True)
        blocks = [
            [com()],
            [cart("l_wrist", "l_arm", weight=0.1), cart("r_wrist", "r_arm"),
             cart("l_sole", "l_leg"), cart("r_sole", "r_leg")],
            [postural(q=bleaf[0][0])],
        ]
        rowblocks, rleaf = [], []
        if config == "C4":
            # self-collision rows (CollisionAvoidance.cpp:96-152): P candidate pairs sorted by distance,
            # about 30 % beyond the detection threshold (-> unused zero rows)
            Jd = np.zeros((B, P, n))
            for r in range(P):
                cols = rng.choice(np.arange(6, n), size=14, replace=False)
                Jd[:, r, cols] = rng.normal(0.0, 0.3, size=(B, 14))
            d = np.sort(rng.uniform(0.0, 0.072, size=(B, P)), axis=1)
            rowblocks.append(Rows(abi.ROWS_COLLISION, P, d_threshold=0.0, detection_threshold=0.05,
                                  bound_scaling=1.0, name="self_collision"))
            rleaf.append((Jd, d, None))

    levels, A, tleaf = [], [], []
    for lev in blocks:
        levels.append([t for (t, _, _) in lev])
        Js = [J for (_, J, _) in lev if J is not None]
        A.append(np.ascontiguousarray(np.concatenate(Js, axis=1)) if Js else None)
        tleaf.append([lf for (_, _, lf) in lev])
    plan = StackPlan(n=n, levels=levels, bounds=bounds, rowblocks=rowblocks,
                     eps_abs=eps_abs_from_factor(eps_factor))
    leaf = {"B": B, "A": A, "task": tleaf, "bound": bleaf, "rows": rleaf}
    return plan, leaf


def perturb(leaf, rng, scale=0.01):
    """temporally coherent next cycle: every float input moves by ~1 % (MPC-rollout-like)."""
    def j(a):
        return None if a is None else a * (1.0 + scale * rng.standard_normal(a.shape))
    out = {"B": leaf["B"], "A": [j(a) for a in leaf["A"]],
           "task": [[tuple(j(x) for x in t) for t in lev] for lev in leaf["task"]],
           "bound": [tuple(j(x) for x in t) for t in leaf["bound"]],
           "rows": [tuple(j(x) for x in t) for t in leaf["rows"]]}
    if leaf.get("C") is not None:      # producer-written constraint rows (config 5: [B_u, -J_f'], [B, -Jc'])
        out["C"] = [j(a) for a in leaf["C"]]
    if leaf.get("W") is not None:      # dense task weights stay as they are (symmetric positive definite)
        out["W"] = leaf["W"]
    if "reg" in leaf:
        out["reg"] = tuple(j(x) for x in leaf["reg"])
    if "reg_A" in leaf:
        out["reg_A"] = j(leaf["reg_A"])
    return out


def add_regularisation(plan, leaf, kind=abi.TASK_GENERIC, rows=None, weight=1e-3, lam=0.1, seed=0, dense=False):
    """attach a user regularisation task (AutoStack::setRegularisationTask) to a synthetic stack: the reference's own
    use (tests/solvers/TestiHQP.cpp:112-120) is a minimum-velocity GenericTask(I, -qdot/dt); a Postural works alike.
    dense=True: a task with a STORED Jacobian (iHQP.cpp:265-278 takes any task): GENERIC (random A_r, b), CARTESIAN
    (6 rows: a link Jacobian and poses) or COM (3 rows); leaf["reg_A"] is A_r [B][rows][n]."""
    rng = np.random.default_rng(seed + 977)
    n, B = plan.n, leaf["B"]
    rows = n if rows is None else rows
    if dense:
        rows = {abi.TASK_CARTESIAN: 6, abi.TASK_COM: 3}.get(kind, rows)
        plan.regularisation = Task(kind, rows, weight=weight, lam=lam, name="regularisation")
        plan.regularisation_dense = True
        leaf["reg_A"] = rng.normal(0.0, 0.3, size=(B, rows, n))
        if kind == abi.TASK_CARTESIAN:
            leaf["reg"] = _cartesian_leaf(rng, B)
        elif kind == abi.TASK_COM:
            p = rng.uniform(-0.2, 0.2, size=(B, 3))
            leaf["reg"] = (p, p + rng.uniform(-0.05, 0.05, size=(B, 3)), None)
        else:
            leaf["reg"] = (rng.normal(0.0, 0.1, size=(B, rows)), None, None)
        return plan, leaf
    plan.regularisation = Task(kind, rows, weight=weight, lam=lam, lam2=2.0 * np.sqrt(lam), name="regularisation")
    if kind == abi.TASK_GENERIC:
        leaf["reg"] = (rng.normal(0.0, 0.1, size=(B, rows)), None, None)
    elif kind == abi.TASK_POSTURAL:
        q = rng.uniform(-1, 1, size=(B, rows))
        leaf["reg"] = (q, q + rng.normal(0, 0.1, size=(B, rows)), None)
    else:   # ACC_POSTURAL: [q_ref - q ; qdot_ref - qdot], qddot_ref
        leaf["reg"] = (rng.normal(0, 0.1, size=(B, 2 * rows)), None, rng.normal(0, 0.1, size=(B, rows)))
    return plan, leaf


def make_id_stack(B, seed=None, nv=38, n_contacts=4, eps_factor=1e6, torque_limits=True):
    """BASELINE config 5: floating-base inverse dynamics in torque mode, x = [qddot (nv); F (3 per point contact)]
    (src/utils/InverseDynamics.cpp:12-28; bindings/python/examples/LittleDog_id.py:60-106).

    levels : 0 = acceleration::CoM (3) + two acceleration::Cartesian (6 each)      [J 0] written by the producer
             1 = acceleration::Postural on qddot ([I_nv 0], implicit)
    rows   : DynamicFeasibility (6 eq) [B_u, -J_f'], FrictionCone (5 per contact), TorqueLimits (nv) [B, -Jc'],
             acceleration::JointLimits (nv unit rows), acceleration::VelocityLimits (nv unit rows)
    Model quantities (B, h, J, Jdot*qdot) are synthetic: B = L L' + I, h ~ N(0, 5^2), J ~ N(0, 0.3^2).
    The stack is feasible by construction around qddot = 0 with supporting contact forces.
    """
    rng = np.random.default_rng(5000 if seed is None else seed)
    nf = 3 * n_contacts
    n = nv + nf
    assert n <= abi.MAX_VARS
    # dynamics
    Lm = rng.normal(0.0, 0.3, size=(B, nv, nv))
    Bm = Lm @ np.transpose(Lm, (0, 2, 1)) + np.eye(nv)
    Jc = np.zeros((B, n_contacts, 3, nv))                      # point-contact linear Jacobians
    Jc[:, :, :, :6] = rng.normal(0.0, 0.5, size=(B, n_contacts, 3, 6))
    for ct in range(n_contacts):
        cols = 6 + ct * 6 + np.arange(6)
        Jc[:, ct][:, :, cols] = rng.normal(0.0, 0.3, size=(B, 3, 6))
    # contact frames and a nominal force inside every cone; h chosen so that (qddot = 0, F = F0) is dynamically
    # consistent on the floating base and well inside the torque limits
    wRl = _rot_exp(rng.normal(0.0, 0.15, size=(B, n_contacts, 3)))
    F0_local = np.concatenate([rng.uniform(-3, 3, size=(B, n_contacts, 2)), rng.uniform(40, 80, size=(B, n_contacts, 1))], axis=2)
    F0 = np.einsum("bcij,bcj->bci", wRl, F0_local)                # world frame
    JcT_F0 = np.einsum("bcij,bci->bj", Jc, F0)                    # sum_c Jc' F0
    h = JcT_F0 + np.concatenate([np.zeros((B, 6)), rng.normal(0.0, 5.0, size=(B, nv - 6))], axis=1)
    tau_max = np.full((B, nv), 30.0)                              # tight enough that some torque limits bind
    tau_max[:, :6] = 1.0e3                                        # floating-base rows: loose (equality handles them)
    # tasks
    Jcom = rng.normal(0.0, 0.3, size=(B, 3, nv))
    Jh = [_limb_jacobian(rng, B, 6, nv, list(range(0, 6)) + list(range(30 + 4 * k, 34 + 4 * k))) for k in range(2)]
    A0 = np.zeros((B, 15, n))
    A0[:, 0:3, :nv] = Jcom
    A0[:, 3:9, :nv] = Jh[0]
    A0[:, 9:15, :nv] = Jh[1]

    def acc_leaf(rows):
        pe = rng.normal(0.0, 0.02, size=(B, rows)); ve = rng.normal(0.0, 0.05, size=(B, rows))
        return np.concatenate([pe, ve], axis=1), rng.normal(0.0, 0.1, size=(B, rows)), None
    q = rng.uniform(-1.0, 1.0, size=(B, nv)); qd = rng.normal(0.0, 0.2, size=(B, nv))
    half = rng.uniform(1.5, 2.5, size=(B, nv))
    levels = [[Task(abi.TASK_ACC_COM, 3, lam=10.0, lam2=5.0, name="com"),
               Task(abi.TASK_ACC_CARTESIAN, 6, lam=10.0, lam2=5.0, name="l_hand"),
               Task(abi.TASK_ACC_CARTESIAN, 6, lam=10.0, lam2=5.0, name="r_hand")],
              [Task(abi.TASK_ACC_POSTURAL, nv, lam=10.0, lam2=5.0, name="postural")]]
    tleaf = [[acc_leaf(3), acc_leaf(6), acc_leaf(6)],
             [(np.concatenate([rng.normal(0.0, 0.1, size=(B, nv)), -qd], axis=1), None, None)]]
    # constraint rows written by the producer (zero-copy into C): dynamic feasibility and torque limits
    Cdyn = np.zeros((B, 6, n)); Ctau = np.zeros((B, nv, n))
    Cdyn[:, :, :nv] = Bm[:, :6, :]
    Ctau[:, :, :nv] = Bm
    for ct in range(n_contacts):
        # DynamicFeasibility.cpp:38-41: -(J[0:k, 0:6])' ; TorqueLimits.cpp:39-40: -(J[0:k, :])'
        Cdyn[:, :, nv + 3 * ct: nv + 3 * ct + 3] = -np.transpose(Jc[:, ct][:, :, :6], (0, 2, 1))
        Ctau[:, :, nv + 3 * ct: nv + 3 * ct + 3] = -np.transpose(Jc[:, ct], (0, 2, 1))
    rowblocks = [Rows(abi.ROWS_DYN_FEASIBILITY, 6, name="dynamic_feasibility"),
                 Rows(abi.ROWS_FRICTION_CONE, 5 * n_contacts, first_col=nv, mu=0.8, name="friction_cones")]
    rleaf = [(h[:, :6].copy(), None, None), (wRl.reshape(B, n_contacts, 9), None, None)]
    Cleaf = [Cdyn, None]
    if torque_limits:
        rowblocks.append(Rows(abi.ROWS_TORQUE_LIMITS, nv, name="torque_limits"))
        rleaf.append((h, tau_max, None)); Cleaf.append(Ctau)
    rowblocks.append(Rows(abi.ROWS_ACC_JOINT_LIMITS, nv, first_col=0, dT=0.001, p=20.0, name="joint_limits"))
    rleaf.append((np.concatenate([q, qd], axis=1), np.concatenate([-half, half], axis=1), np.full((B, nv), 500.0)))
    Cleaf.append(None)
    # four row blocks at most (OSOT_MAX_ROWBLOCKS): velocity limits only when torque limits are left out
    if not torque_limits:
        rowblocks.append(Rows(abi.ROWS_ACC_VELOCITY_LIMITS, nv, first_col=0, dT=0.001, p=20.0, name="velocity_limits"))
        rleaf.append((qd, np.full((B, nv), 20.0), None)); Cleaf.append(None)
    plan = StackPlan(n=n, levels=levels, bounds=[], rowblocks=rowblocks, eps_abs=eps_abs_from_factor(eps_factor))
    leaf = {"B": B, "A": [A0, None], "task": tleaf, "bound": [], "rows": rleaf, "C": Cleaf,
            "model": {"B": Bm, "h": h, "Jc": Jc, "nv": nv}}
    return plan, leaf


def computed_torque(leaf, x):
    """InverseDynamics::computedTorque (src/utils/InverseDynamics.cpp:57-96): tau = B qddot + h - sum_c Jc' F_c;
    the six floating-base rows must vanish."""
    md = leaf["model"]; nv = md["nv"]
    qdd = x[:, :nv]; F = x[:, nv:].reshape(x.shape[0], -1, 3)
    return np.einsum("bij,bj->bi", md["B"], qdd) + md["h"] - np.einsum("bcij,bci->bj", md["Jc"], F)


def make_generic_stack(B, n, level_rows, n_eq=0, n_ineq=0, seed=0, box=0.5, duplicate_eq_in_level=None,
                       postural_last=True, eps_factor=1e6, n_local=0, local_level=0, unit_box=None, local_equality=False):
    """small generic stacks (tasks::GenericTask blocks, GenericConstraint rows, generic box) for robot-free tests:
    e.g. a Panda-like 7-variable 2-level stack, or stacks whose optimality rows duplicate global equality rows
    (the `<< (l_sole + r_sole)` situation of examples/cpp/coman_ik.cpp:442 where 39 equality rows meet 35
    variables).  duplicate_eq_in_level = k copies the global equality rows into level k's task (consistent b)."""
    rng = np.random.default_rng(seed)
    levels, A, tleaf = [], [], []
    Ceq = rng.normal(0.0, 0.5, size=(B, n_eq, n)) if n_eq else None
    beq = rng.uniform(-0.02, 0.02, size=(B, n_eq)) if n_eq else None
    for k, m in enumerate(level_rows):
        Ak = rng.normal(0.0, 0.4, size=(B, m, n))
        bk = rng.normal(0.0, 0.05, size=(B, m))
        if duplicate_eq_in_level == k and n_eq:
            Ak = np.concatenate([Ceq, Ak], axis=1); bk = np.concatenate([beq, bk], axis=1)
        levels.append([Task(abi.TASK_GENERIC, Ak.shape[1], name=f"generic{k}")])
        A.append(np.ascontiguousarray(Ak)); tleaf.append([(bk, None, None)])
    if postural_last:
        q = rng.uniform(-1, 1, size=(B, n))
        levels.append([Task(abi.TASK_POSTURAL, n, lam=0.1, name="postural")])
        A.append(None); tleaf.append([(q, q + rng.normal(0, 0.1, size=(B, n)), None)])
    rowblocks, rleaf = [], []
    if n_eq:
        rowblocks.append(Rows(abi.ROWS_GENERIC, n_eq, name="equalities")); rleaf.append((Ceq, beq, beq.copy()))
    if n_ineq:
        Ci = rng.normal(0.0, 0.5, size=(B, n_ineq, n))
        rowblocks.append(Rows(abi.ROWS_GENERIC, n_ineq, name="inequalities"))
        rleaf.append((Ci, -rng.uniform(0.01, 0.2, size=(B, n_ineq)), rng.uniform(0.01, 0.2, size=(B, n_ineq))))
    if n_local:   # task-local rows (`task << constraint`): tight enough to bind at their level
        Cl = rng.normal(0.0, 0.5, size=(B, n_local, n))
        rowblocks.append(Rows(abi.ROWS_GENERIC, n_local, name="task_local", level=local_level))
        lol, upl = -rng.uniform(0.001, 0.02, size=(B, n_local)), rng.uniform(0.001, 0.02, size=(B, n_local))
        if local_equality:   # `task << equality`: a row the solution of the level above does NOT satisfy
            upl = lol.copy()
        rleaf.append((Cl, lol, upl))
    if unit_box is not None:   # (level or None, half width): a box as unit rows; with a level: a task-local bound
        lvl, hw = unit_box
        rowblocks.append(Rows(abi.ROWS_UNIT_GENERIC, n, name="unit_box", first_col=0, level=lvl))
        rleaf.append((np.full((B, n), -hw), np.full((B, n), hw), None))
    bounds, bleaf = [], []
    if box:
        bounds.append(Bound(abi.BOUND_GENERIC, name="box")); bleaf.append((np.full((B, n), -box), np.full((B, n), box), None))
    plan = StackPlan(n=n, levels=levels, bounds=bounds, rowblocks=rowblocks, eps_abs=eps_abs_from_factor(eps_factor))
    return plan, {"B": B, "A": A, "task": tleaf, "bound": bleaf, "rows": rleaf}


def make_lowrank_stack(B, n, m=3, seed=0, weight=0.3, postural_weight=None, dependent=False, zero_row=False,
                       second_level_rows=5, box=0.4, eps_factor=1e6):
    """a level with only a few stored rows (the closed-form low-rank path of the cascade kernel): m <= 4 generic
    rows with a non-unit weight, optionally sharing the level with a weighted Postural block (soft priority, as in
    coman_ik.cpp:429), optionally with a linearly dependent or an all-zero row; a second generic level and a box."""
    rng = np.random.default_rng(seed)
    A0 = rng.normal(0.0, 0.4, size=(B, m, n))
    b0 = rng.normal(0.0, 0.05, size=(B, m))
    if dependent and m >= 2:
        A0[:, m - 1] = 2.0 * A0[:, 0]          # dependent direction (inconsistent right-hand side: least squares)
    if zero_row:
        A0[:, min(1, m - 1)] = 0.0
    lev0 = [Task(abi.TASK_GENERIC, m, weight=weight, name="few_rows")]
    leaf0 = [(b0, None, None)]
    if postural_weight is not None:
        q = rng.uniform(-1, 1, size=(B, n))
        lev0.append(Task(abi.TASK_POSTURAL, n, weight=postural_weight, lam=0.1, name="postural_soft"))
        leaf0.append((q, q + rng.normal(0, 0.1, size=(B, n)), None))
    levels, A, tleaf = [lev0], [np.ascontiguousarray(A0)], [leaf0]
    if second_level_rows:
        A1 = rng.normal(0.0, 0.4, size=(B, second_level_rows, n))
        levels.append([Task(abi.TASK_GENERIC, second_level_rows, name="second")])
        A.append(A1); tleaf.append([(rng.normal(0.0, 0.05, size=(B, second_level_rows)), None, None)])
    bounds = [Bound(abi.BOUND_GENERIC, name="box")]
    bleaf = [(np.full((B, n), -box), np.full((B, n), box), None)]
    plan = StackPlan(n=n, levels=levels, bounds=bounds, rowblocks=[], eps_abs=eps_abs_from_factor(eps_factor))
    return plan, {"B": B, "A": A, "task": tleaf, "bound": bleaf, "rows": []}


def make_subtask_stack(B, seed=0, n=32, eps_factor=1e6):
    """config-3-like stack built from SubTasks (`task % {rows}`, src/tasks/SubTask.cpp): CoM restricted to x, y with
    its own lambda; position-only wrists (rows 0..2 of the Cartesian tasks, one with a sub-task lambda of 0.5), full
    feet; a Postural sub-task on the actuated joints 6..n-1 (its unit rows are stored, not implicit)."""
    from .plan import subtask
    rng = np.random.default_rng(seed)
    com = Task(abi.TASK_COM, 3, lam=0.1, name="com")
    carts = {nm: Task(abi.TASK_CARTESIAN, 6, weight=(0.1 if nm == "l_wrist" else 1.0), lam=0.1, name=nm)
             for nm in ("l_wrist", "r_wrist", "l_sole", "r_sole")}
    post = Task(abi.TASK_POSTURAL, n, lam=0.01, name="postural")
    lev0 = [subtask(com, [0, 1], lam=0.7)]
    lev1 = [subtask(carts["l_wrist"], [0, 1, 2], lam=0.5), subtask(carts["r_wrist"], [0, 1, 2]), carts["l_sole"], carts["r_sole"]]
    act = list(range(6, n))
    lev2 = [subtask(post, act, n=n)]
    bounds, bleaf = _box_leaf(rng, B, n, jl=True, vl=True)
    q = bleaf[0][0]
    A0 = rng.normal(0.0, 0.3, size=(B, 3, n))[:, [0, 1]]
    limb = {"l_wrist": "l_arm", "r_wrist": "r_arm", "l_sole": "l_leg", "r_sole": "r_leg"}
    J = {nm: _limb_jacobian(rng, B, 6, n, _BASE + _LIMBS[limb[nm]]) for nm in carts}
    A1 = np.concatenate([J["l_wrist"][:, :3], J["r_wrist"][:, :3], J["l_sole"], J["r_sole"]], axis=1)
    A2 = np.zeros((B, len(act), n))
    for r, jn in enumerate(act):
        A2[:, r, jn] = 1.0
    p = rng.uniform(-0.2, 0.2, size=(B, 3))
    tleaf = [[(p, p + rng.uniform(-0.05, 0.05, size=(B, 3)), None)],
             [_cartesian_leaf(rng, B) for _ in range(4)],
             [(q, q + rng.normal(0.0, 0.1, size=(B, n)), None)]]
    plan = StackPlan(n=n, levels=[lev0, lev1, lev2], bounds=bounds, rowblocks=[], eps_abs=eps_abs_from_factor(eps_factor))
    leaf = {"B": B, "A": [np.ascontiguousarray(A0), np.ascontiguousarray(A1), A2], "task": tleaf, "bound": bleaf, "rows": []}
    return plan, leaf


def make_feature_stack(B, seed=0, n=32, eps_factor=1e6, body_frame=True, dense=True, bands=True, candidates=24, many_blocks=True):
    """a velocity stack that exercises the options beyond the benchmark configurations (all of them parts of the reference's
    surface): a Cartesian task with a BODY Jacobian (Cartesian.cpp:93-100), a task with a full weight matrix
    (Task.h:273-300) next to scalar-weighted ones, TaskToConstraint rows with per-row error bands
    (TaskToConstraint.cpp:34-68), a collision block that picks its rows among more candidate pairs than rows
    (CollisionAvoidance.cpp:120-131), and more than four row blocks.  Returns (plan, leaf); leaf["W"] holds the weight
    matrices."""
    rng = np.random.default_rng(seed)
    bounds, bleaf = _box_leaf(rng, B, n, jl=True, vl=True)

    def cart(name, limb, weight=1.0, lam=0.1, **kw):
        return (Task(abi.TASK_CARTESIAN, 6, weight=weight, lam=lam, name=name, **kw),
                _limb_jacobian(rng, B, 6, n, _BASE + _LIMBS[limb]), _cartesian_leaf(rng, B))

    def spd(rows):
        M = rng.normal(0.0, 0.4, size=(B, rows, rows))
        return M @ np.transpose(M, (0, 2, 1)) + 0.5 * np.eye(rows)

    l0 = [cart("l_sole", "l_leg", body_frame=body_frame), cart("r_sole", "r_leg")]
    # (two tasks on the SAME limb: they conflict, so the weights -- and the off-diagonal entries of W -- shape the answer)
    l1 = [cart("l_wrist", "l_arm", weight=0.3, dense_weight=dense), cart("l_elbow", "l_arm", lam=0.2)]
    q = bleaf[0][0]
    l2 = [(Task(abi.TASK_POSTURAL, n, lam=0.01, name="postural"), None, (q, q + rng.normal(0.0, 0.1, size=(B, n)), None))]
    levels, A, tleaf, Wl = [], [], [], []
    for lev in (l0, l1, l2):
        levels.append([t for (t, _, _) in lev])
        Js = [J for (_, J, _) in lev if J is not None]
        A.append(np.ascontiguousarray(np.concatenate(Js, axis=1)) if Js else None)
        tleaf.append([lf for (_, _, lf) in lev])
        Wl.append([spd(t.rows) if t.dense_weight else None for (t, _, _) in lev])
    rowblocks, rleaf, Cleaf = [], [], []
    # CoM as a constraint with a per-row band (TaskToConstraint with err_lb / err_ub vectors)
    p = rng.uniform(-0.2, 0.2, size=(B, 3))
    elb = [-0.02, -0.01, -0.03] if bands else 0.0
    eub = [0.01, 0.02, 0.0] if bands else 0.0
    rowblocks.append(Rows(abi.ROWS_TASK_COM, 3, lam=0.1, err_lb=elb, err_ub=eub, name="com_band"))
    rleaf.append((p, p + rng.uniform(-0.01, 0.01, size=(B, 3)), None)); Cleaf.append(rng.normal(0.0, 0.3, size=(B, 3, n)))
    # self-collision: `candidates` pairs supplied in arbitrary order, the 8 closest become rows
    P = 8
    Jd = np.zeros((B, candidates, n))
    for r in range(candidates):
        cols = rng.choice(np.arange(6, n), size=14, replace=False)
        Jd[:, r, cols] = rng.normal(0.0, 0.3, size=(B, 14))
    d = rng.uniform(0.0, 0.09, size=(B, candidates))
    rowblocks.append(Rows(abi.ROWS_COLLISION, P, d_threshold=0.0, detection_threshold=0.05, bound_scaling=1.0,
                          n_candidates=(candidates if candidates != P else 0), name="self_collision"))
    rleaf.append((Jd, d, None)); Cleaf.append(None)
    if many_blocks:   # four more (small) generic blocks: six row blocks in all
        for i in range(4):
            Ci = rng.normal(0.0, 0.5, size=(B, 2, n))
            rowblocks.append(Rows(abi.ROWS_GENERIC, 2, name=f"generic{i}"))
            rleaf.append((Ci, -rng.uniform(0.05, 0.3, size=(B, 2)), rng.uniform(0.05, 0.3, size=(B, 2)))); Cleaf.append(None)
    plan = StackPlan(n=n, levels=levels, bounds=bounds, rowblocks=rowblocks, eps_abs=eps_abs_from_factor(eps_factor))
    leaf = {"B": B, "A": A, "task": tleaf, "bound": bleaf, "rows": rleaf, "C": Cleaf, "W": Wl}
    return plan, leaf


def wide_id_levels(rng, nv=56, ncon=5, tau_max=60.0):
    """two levels of a floating-base inverse-dynamics stack WIDER than 64 variables as explicit QPs in BackEnd convention (the shape of
    src/utils/InverseDynamics.cpp:12-28): x = [qddot (nv); F (3 per point contact)]; rows: dynamic feasibility (6 equalities), friction
    pyramids (5 rows per contact), torque limits (nv - 6 bilateral rows); box: acceleration limits and force limits.  Level 0: 15 task
    rows on qddot (a CoM and two Cartesian tasks), level 1: a Postural task on qddot under level 0's optimality rows.
    Returns (n, level) with level(k, xs) -> (H, g, A, lA, uA, l, u), xs = the solutions of the levels above."""
    nf = 3 * ncon
    n = nv + nf
    Q = rng.normal(size=(nv, nv)) * 0.3
    M = Q.T @ Q + np.eye(nv) * 2.0
    Jc = rng.normal(size=(nf, nv)) * 0.5
    h = rng.normal(size=nv) * 2.0
    h[2] += 9.81 * 5.0
    dyn = np.hstack([M, -Jc.T])
    rows, lo, up = [], [], []
    for r in range(6):
        rows.append(dyn[r]); lo.append(-h[r]); up.append(-h[r])
    mu = 0.7
    for c in range(ncon):
        fx, fy, fz = nv + 3 * c, nv + 3 * c + 1, nv + 3 * c + 2
        for (a, sgn) in ((fx, 1), (fx, -1), (fy, 1), (fy, -1)):
            row = np.zeros(n); row[a] = sgn; row[fz] = -mu
            rows.append(row); lo.append(-np.inf); up.append(0.0)
        row = np.zeros(n); row[fz] = 1.0
        rows.append(row); lo.append(0.0); up.append(1.0e3)
    for r in range(6, nv):
        rows.append(dyn[r]); lo.append(-tau_max - h[r]); up.append(tau_max - h[r])
    Cm, lo, up = np.array(rows), np.array(lo), np.array(up)
    l = np.concatenate([-np.full(nv, 80.0), np.full(nf, -1.0e3)])
    u = np.concatenate([np.full(nv, 80.0), np.full(nf, 1.0e3)])
    m0 = 15
    A0 = np.hstack([rng.normal(size=(m0, nv)) * 0.6, np.zeros((m0, nf))])
    b0 = rng.normal(size=m0) * 3.0
    qref = rng.normal(size=nv) * 0.5

    def level(k, xs):
        if k == 0:
            return A0.T @ A0, -A0.T @ b0, Cm, lo, up, l, u
        H = np.zeros((n, n)); H[:nv, :nv] = np.eye(nv)
        g = np.zeros(n); g[:nv] = -qref
        t = A0 @ xs[0]
        return H, g, np.vstack([Cm, A0]), np.concatenate([lo, t]), np.concatenate([up, t]), l, u
    return n, level
