"""oracle/pyoracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes view of oracle/liboracle.so (the plain-C restatement, oracle/osot_oracle.c).  Imported by
tests/, tests/golden/make_golden.py, __graft_entry__.smoke() and bench.py's cpu_baseline leg --
never by opensot_amd.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libqpoases_ref.so")
ORC_MAX_LEVELS = 8
BE_EIQP_REFFORM, BE_EIQP_EQ, BE_QPOASES_REF = 0, 1, 2
INFTY = 1.0e20
dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)

# kinds, duplicated from include/osot_mi355x.h on purpose (the oracle must not import the product)
TASK_GENERIC, TASK_CARTESIAN, TASK_COM, TASK_POSTURAL, TASK_ACC_CARTESIAN, TASK_ACC_COM, TASK_ACC_POSTURAL = range(7)
BOUND_GENERIC, BOUND_JOINT_LIMITS, BOUND_VELOCITY_LIMITS = range(3)
(ROWS_GENERIC, ROWS_COLLISION, ROWS_DYN_FEASIBILITY, ROWS_TORQUE_LIMITS, ROWS_FRICTION_CONE,
 ROWS_ACC_JOINT_LIMITS, ROWS_ACC_VELOCITY_LIMITS, ROWS_TASK_CARTESIAN, ROWS_TASK_COM, ROWS_UNIT_GENERIC) = range(10)
IMPLICIT_IDENTITY_TASKS = (TASK_POSTURAL, TASK_ACC_POSTURAL)


class OrcBatch(C.Structure):
    _fields_ = [("n", C.c_int), ("L", C.c_int), ("B", C.c_int),
                ("m", C.c_int * ORC_MAX_LEVELS), ("ma", C.c_int * ORC_MAX_LEVELS),
                ("A", dp * ORC_MAX_LEVELS), ("b", dp * ORC_MAX_LEVELS),
                ("w", dp * ORC_MAX_LEVELS), ("c", dp * ORC_MAX_LEVELS),
                ("nc", C.c_int), ("C", dp), ("lo", dp), ("up", dp), ("l", dp), ("u", dp),
                ("eps_abs", C.c_double), ("active", C.POINTER(C.c_ubyte)),
                ("mr", C.c_int), ("Ar", dp), ("br", dp), ("wr", C.c_double), ("row_level", ip),
                ("Wd", dp * ORC_MAX_LEVELS)]


_lib = None


def build(force=False):
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/external/qpOASES-ext/src") and (force or not os.path.exists(_REF_SO)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def ref_available():
    return os.path.exists(_REF_SO)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_eiquadprog.restype = C.c_double
        L.orc_eiquadprog.argtypes = [C.c_int, dp, dp, C.c_int, dp, dp, C.c_int, dp, dp, dp, C.c_double, ip, ip, ip]
        L.orc_backend_solve.argtypes = [C.c_int, C.c_int, dp, dp, C.c_int, dp, dp, dp, dp, dp, C.c_double, dp, ip]
        L.orc_cost_function.argtypes = [C.POINTER(OrcBatch), C.c_int, C.c_int, dp, dp]
        L.orc_add_regularisation.argtypes = [C.POINTER(OrcBatch), C.c_int, dp, dp]
        L.orc_ihqp_solve.argtypes = [C.POINTER(OrcBatch), C.c_int, C.c_int, C.c_void_p, dp, dp, ip]
        L.orc_ihqp_solve_batch.restype = C.c_double
        L.orc_ihqp_solve_batch.argtypes = [C.POINTER(OrcBatch), C.c_int, C.c_int, C.c_int, dp, dp, ip,
                                           C.POINTER(C.c_longlong)]
        L.orc_ref_load.argtypes = [C.c_char_p]
        L.orc_ref_configure.argtypes = [C.c_double, C.c_double]
        L.orc_rot_to_quat.argtypes = [dp, dp]
        L.orc_cartesian_error.argtypes = [dp, dp, dp, dp, dp, dp]
        L.orc_cartesian_b.argtypes = [dp, dp, dp, dp, dp, C.c_double, C.c_double, dp]
        L.orc_cartesian_b_body.argtypes = [dp, dp, dp, dp, dp, C.c_double, C.c_double, dp]
        L.orc_com_b.argtypes = [dp, dp, dp, C.c_double, dp]
        L.orc_postural_b.argtypes = [C.c_int, dp, dp, dp, C.c_double, dp]
        L.orc_joint_limits.argtypes = [C.c_int, dp, dp, dp, C.c_double, dp, dp]
        L.orc_velocity_limits.argtypes = [C.c_int, dp, C.c_double, dp, dp]
        L.orc_merge_box.argtypes = [C.c_int, dp, dp, dp, dp]
        L.orc_collision_rows.argtypes = [C.c_int, C.c_int, C.c_int, dp, dp, C.c_double, C.c_double,
                                         C.c_double, dp, dp, dp]
        L.orc_acc_task_b.argtypes = [C.c_int, dp, dp, dp, dp, C.c_double, C.c_double, dp]
        L.orc_acc_task_b_gains.argtypes = [C.c_int, dp, dp, dp, dp, dp, dp, C.c_double, C.c_double, dp]
        L.orc_cartesian_inertia_inverse.argtypes = [C.c_int, C.c_int, dp, dp, dp]
        L.orc_acc_postural_b.argtypes = [C.c_int, dp, dp, dp, C.c_double, C.c_double, dp]
        L.orc_torque_limit_bounds.argtypes = [C.c_int, dp, dp, dp, dp]
        L.orc_friction_cone_rows.argtypes = [dp, C.c_double, dp]
        L.orc_acc_joint_limits.argtypes = [C.c_int, dp, dp, dp, dp, dp, C.c_double, dp, dp]
        L.orc_acc_velocity_limits.argtypes = [C.c_int, dp, dp, C.c_double, C.c_double, dp, dp]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(dp)


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


# --------------------------------------------------------------------------------------------
# AutoStack::update restatement: leaf -> assembled arrays
# --------------------------------------------------------------------------------------------
def assemble(plan, leaf, task_active=None):
    """plan: any object with .n, .levels (lists of tasks with kind/rows/weight/lam/orientation_gain),
    .bounds, .rowblocks, .eps_abs; leaf: dict from opensot_amd.synth.  Returns the assembled dict.
    task_active: {(level, task): bool} -- Task::setActive(false) zeroes the task's A (Task.h:232-239, 383-387)"""
    L = lib()
    n, B = plan.n, leaf["B"]
    task_active = task_active or {}
    out = {"n": n, "B": B, "L": len(plan.levels), "eps_abs": plan.eps_abs,
           "m": [], "ma": [], "A": [], "b": [], "w": [], "c": []}
    zeros6 = np.zeros(6)
    zerosn = np.zeros(n)
    for k, lev in enumerate(plan.levels):
        m = sum(t.rows for t in lev)
        # a whole Postural block is implicit (A = [I 0]); a Postural SubTask stores its unit rows like any block
        is_impl = lambda t: (t.kind in IMPLICIT_IDENTITY_TASKS and not getattr(t, "row_mask", 0)
                             and not getattr(t, "dense_weight", False))
        ma = sum(t.rows for t in lev if not is_impl(t))
        b = np.zeros((B, m))
        w = np.ones((B, m))
        # non-diagonal weights: the level's W = blockdiag(weight_i * W_i) (Aggregated::generateWeight, Aggregated.cpp:265-279)
        dense = any(getattr(t, "dense_weight", False) for t in lev)
        Wd = np.zeros((B, m, m)) if dense else None
        # Task::setActive(false): A = 0 (Task.h:383-387).  An inactive implicit [I 0] block becomes a stored block of zero rows
        Ak = _c(leaf["A"][k]).copy() if ma else None
        for j, t in enumerate(lev):
            if task_active.get((k, j), True):
                continue
            o = sum(tt.rows for tt in lev[:j])
            if is_impl(t):
                Ak = np.concatenate([Ak, np.zeros((B, t.rows, n))], axis=1) if Ak is not None else np.zeros((B, t.rows, n))
                ma += t.rows
            else:
                Ak[:, o:o + t.rows] = 0.0
        off = 0
        for j, t in enumerate(lev):
            p0, p1, p2 = (_c(x) for x in leaf["task"][k][j])
            w[:, off:off + t.rows] = t.weight   # scalar * W (AutoStack.cpp:16-47), W = I by default
            if dense:
                Wi = _c(leaf["W"][k][j]) if getattr(t, "dense_weight", False) else np.broadcast_to(np.eye(t.rows), (B, t.rows, t.rows))
                Wd[:, off:off + t.rows, off:off + t.rows] = t.weight * Wi
            # SubTask (src/tasks/SubTask.cpp:22-112): the parent's b is formed in full (size pr), then the kept rows are
            # gathered in index order and scaled by the sub-task's own lambda (SubTask.cpp:44-58)
            mask = getattr(t, "row_mask", 0)
            pr = t.parent_size(n) if mask else t.rows
            kept = [q for q in range(64) if (mask >> q) & 1]
            for i in range(B):
                bi = np.zeros(pr) if mask else b[i, off:off + t.rows]
                rows_ = pr
                if t.kind == TASK_CARTESIAN:
                    tw = p2[i] if p2 is not None else zeros6
                    fn = L.orc_cartesian_b_body if getattr(t, "body_frame", False) else L.orc_cartesian_b
                    fn(_p(p0[i, :9]), _p(p0[i, 9:]), _p(p1[i, :9]), _p(p1[i, 9:]), _p(tw), t.lam, t.orientation_gain, _p(bi))
                elif t.kind == TASK_COM:
                    L.orc_com_b(_p(p0[i]), _p(p1[i]), _p(p2[i] if p2 is not None else zeros6[:3]),
                                t.lam, _p(bi))
                elif t.kind == TASK_POSTURAL:
                    L.orc_postural_b(rows_, _p(p0[i]), _p(p1[i]), _p(p2[i] if p2 is not None else zerosn),
                                     t.lam, _p(bi))
                elif t.kind in (TASK_ACC_CARTESIAN, TASK_ACC_COM):
                    pe = np.ascontiguousarray(p0[i, :rows_]); ve = np.ascontiguousarray(p0[i, rows_:2 * rows_])
                    if getattr(t, "acc_gain_matrices", False):     # p0 = [pose_err; vel_err; Gp; Gd] (Cartesian.cpp:152-173)
                        Gp = np.ascontiguousarray(p0[i, 2 * rows_:2 * rows_ + rows_ * rows_])
                        Gd = np.ascontiguousarray(p0[i, 2 * rows_ + rows_ * rows_:2 * rows_ + 2 * rows_ * rows_])
                        L.orc_acc_task_b_gains(rows_, _p(pe), _p(ve), _p(p1[i]), _p(p2[i]) if p2 is not None else None,
                                               _p(Gp), _p(Gd), t.lam, t.lam2, _p(bi))
                    else:
                        L.orc_acc_task_b(rows_, _p(pe), _p(ve), _p(p1[i]), _p(p2[i]) if p2 is not None else None,
                                         t.lam, t.lam2, _p(bi))
                elif t.kind == TASK_ACC_POSTURAL:
                    pe = np.ascontiguousarray(p0[i, :rows_]); ve = np.ascontiguousarray(p0[i, rows_:])
                    L.orc_acc_postural_b(rows_, _p(pe), _p(ve), _p(p2[i]) if p2 is not None else None,
                                         t.lam, t.lam2, _p(bi))
                else:  # GENERIC: b supplied (GenericTask.cpp:5-56)
                    bi[:] = p0[i]
                if mask:
                    b[i, off:off + t.rows] = bi[kept] * t.sub_lam
            off += t.rows
        out["m"].append(m); out["ma"].append(ma)
        out["A"].append(Ak if ma else None)
        out["b"].append(b); out["w"].append(w); out["c"].append(None)
        out.setdefault("Wdense", []).append(Wd)
    # user regularisation task (AutoStack::setRegularisationTask, AutoStack.h:78-92): an identity-Jacobian task
    # ([I_rows 0]: GenericTask(I, b) as in TestiHQP.cpp:118-120, Postural, MinimumVelocity) whose cost iHQP adds to
    # every level (iHQP.cpp:265-266, 274-278)
    t = getattr(plan, "regularisation", None)
    if t is not None:
        p0, p1, p2 = (_c(x) for x in leaf["reg"])
        br = np.zeros((B, t.rows))
        for i in range(B):
            if t.kind == TASK_POSTURAL:
                L.orc_postural_b(t.rows, _p(p0[i]), _p(p1[i]), _p(p2[i] if p2 is not None else zerosn), t.lam, _p(br[i]))
            elif t.kind == TASK_ACC_POSTURAL:
                pe = np.ascontiguousarray(p0[i, :t.rows]); ve = np.ascontiguousarray(p0[i, t.rows:])
                L.orc_acc_postural_b(t.rows, _p(pe), _p(ve), _p(p2[i]) if p2 is not None else None, t.lam, t.lam2, _p(br[i]))
            elif t.kind == TASK_CARTESIAN:
                L.orc_cartesian_b(_p(p0[i, :9]), _p(p0[i, 9:]), _p(p1[i, :9]), _p(p1[i, 9:]), _p(p2[i] if p2 is not None else zeros6),
                                  t.lam, t.orientation_gain, _p(br[i]))
            elif t.kind == TASK_COM:
                L.orc_com_b(_p(p0[i]), _p(p1[i]), _p(p2[i] if p2 is not None else zeros6[:3]), t.lam, _p(br[i]))
            else:
                br[i] = p0[i]
        # a regularisation task with a stored Jacobian (any task: iHQP.cpp:265-278): A_r [B][rows][n] from the producer
        Ar = _c(leaf["reg_A"]) if getattr(plan, "regularisation_dense", False) else None
        out["reg"] = {"A": Ar, "b": br, "w": t.weight}
    # box (constraints::Aggregated, Aggregated.cpp:141-148)
    if plan.bounds:
        l = np.full((B, n), -np.inf)
        u = np.full((B, n), np.inf)
        l2 = np.zeros(n); u2 = np.zeros(n)
        for j, bd in enumerate(plan.bounds):
            p0, p1, p2 = (_c(x) for x in leaf["bound"][j])
            for i in range(B):
                if bd.kind == BOUND_JOINT_LIMITS:
                    L.orc_joint_limits(n, _p(p0[i]), _p(p1[i]), _p(p2[i]), bd.scaling, _p(l2), _p(u2))
                elif bd.kind == BOUND_VELOCITY_LIMITS:
                    L.orc_velocity_limits(n, _p(p0[i]), bd.dT, _p(l2), _p(u2))
                else:
                    l2[:] = p0[i]; u2[:] = p1[i]
                if j == 0:
                    l[i] = l2; u[i] = u2
                else:
                    L.orc_merge_box(n, _p(l[i]), _p(u[i]), _p(l2), _p(u2))
        out["l"], out["u"] = l, u
    else:
        out["l"] = out["u"] = None
    # global rows
    nc = sum(r.rows for r in plan.rowblocks)
    out["nc"] = nc
    if nc:
        Cm = np.zeros((B, nc, n)); lo = np.zeros((B, nc)); up = np.zeros((B, nc))
        off = 0
        for j, rb in enumerate(plan.rowblocks):
            p0, p1, p2 = (_c(x) for x in leaf["rows"][j])
            for i in range(B):
                sl = slice(off, off + rb.rows)
                loi = np.zeros(rb.rows); upi = np.zeros(rb.rows)
                if rb.kind == ROWS_COLLISION:
                    Ci = np.zeros((rb.rows, n))
                    ncand = getattr(rb, "n_candidates", 0) or rb.rows
                    # getOrderedCollisionPairIndices (CollisionAvoidance.cpp:120): the pairs in order of distance
                    # (xbot2's collision module is not vendored: closest first, ties by index, is this build's reading)
                    order = np.argsort(p1[i][:ncand], kind="stable")
                    Jo = np.ascontiguousarray(p0[i][:ncand][order]); do = np.ascontiguousarray(p1[i][:ncand][order])
                    L.orc_collision_rows(n, ncand, rb.rows, _p(Jo), _p(do), rb.d_threshold,
                                         rb.detection_threshold, rb.bound_scaling, _p(Ci), _p(loi), _p(upi))
                    Cm[i, sl] = Ci; lo[i, sl] = loi; up[i, sl] = upi
                elif rb.kind == ROWS_DYN_FEASIBILITY:      # rows [B_u, -J_f'] come from the producer (leaf "C")
                    Cm[i, sl] = leaf["C"][j][i]; lo[i, sl] = -p0[i]; up[i, sl] = -p0[i]
                elif rb.kind == ROWS_UNIT_GENERIC:     # a box on variables first_col .. first_col+rows-1, as rows
                    Cm[i, sl, rb.first_col:rb.first_col + rb.rows] = np.eye(rb.rows); lo[i, sl] = p0[i]; up[i, sl] = p1[i]
                elif rb.kind in (ROWS_TASK_CARTESIAN, ROWS_TASK_COM):
                    # constraints::TaskToConstraint::generateAll (TaskToConstraint.cpp:59-68): Aineq = task A (from the
                    # producer), bLower/bUpper = task b + err_lb / err_ub
                    bt = np.zeros(rb.rows)
                    if rb.kind == ROWS_TASK_CARTESIAN:
                        tw = p2[i] if p2 is not None else zeros6
                        fn = L.orc_cartesian_b_body if getattr(rb, "body_frame", False) else L.orc_cartesian_b
                        fn(_p(p0[i, :9]), _p(p0[i, 9:]), _p(p1[i, :9]), _p(p1[i, 9:]), _p(tw), rb.lam, rb.orientation_gain, _p(bt))
                    else:
                        L.orc_com_b(_p(p0[i]), _p(p1[i]), _p(p2[i] if p2 is not None else zeros6[:3]), rb.lam, _p(bt))
                    # _bLowerBound = b + _err_lb, _bUpperBound = b + _err_ub, vectors (TaskToConstraint.cpp:61-68)
                    Cm[i, sl] = leaf["C"][j][i]
                    lo[i, sl] = bt + np.broadcast_to(np.asarray(rb.err_lb, dtype=float), (rb.rows,))
                    up[i, sl] = bt + np.broadcast_to(np.asarray(rb.err_ub, dtype=float), (rb.rows,))
                elif rb.kind == ROWS_TORQUE_LIMITS:
                    L.orc_torque_limit_bounds(rb.rows, _p(p0[i]), _p(p1[i]), _p(loi), _p(upi))
                    Cm[i, sl] = leaf["C"][j][i]; lo[i, sl] = loi; up[i, sl] = upi
                elif rb.kind == ROWS_FRICTION_CONE:
                    A53 = np.zeros(15)
                    for ct in range(rb.rows // 5):
                        L.orc_friction_cone_rows(_p(np.ascontiguousarray(p0[i, ct])), rb.mu, _p(A53))
                        Cm[i, off + 5 * ct: off + 5 * ct + 5, rb.first_col + 3 * ct: rb.first_col + 3 * ct + 3] = A53.reshape(5, 3)
                    lo[i, sl] = -1.0e20; up[i, sl] = 0.0
                elif rb.kind == ROWS_ACC_JOINT_LIMITS:
                    nr = rb.rows
                    L.orc_acc_joint_limits(nr, _p(np.ascontiguousarray(p0[i, :nr])), _p(np.ascontiguousarray(p0[i, nr:])),
                                           _p(np.ascontiguousarray(p1[i, :nr])), _p(np.ascontiguousarray(p1[i, nr:])),
                                           _p(p2[i]), rb.dT * rb.p, _p(loi), _p(upi))
                    Cm[i, sl, rb.first_col:rb.first_col + nr] = np.eye(nr); lo[i, sl] = loi; up[i, sl] = upi
                elif rb.kind == ROWS_ACC_VELOCITY_LIMITS:
                    nr = rb.rows
                    L.orc_acc_velocity_limits(nr, _p(p0[i]), _p(p1[i]), rb.dT, rb.p, _p(loi), _p(upi))
                    Cm[i, sl, rb.first_col:rb.first_col + nr] = np.eye(nr); lo[i, sl] = loi; up[i, sl] = upi
                else:
                    Cm[i, off:off + rb.rows] = p0[i]; lo[i, off:off + rb.rows] = p1[i]; up[i, off:off + rb.rows] = p2[i]
            off += rb.rows
        out["C"], out["lo"], out["up"] = Cm, lo, up
        # task-local row blocks (Task::getConstraints(), iHQP.cpp:190): 0 = global, k + 1 = rows of level k only
        rl = []
        for rb in plan.rowblocks:
            lv = getattr(rb, "level", None)
            rl += [0 if lv is None else lv + 1] * rb.rows
        if any(rl):
            out["row_level"] = np.asarray(rl, dtype=np.int32)
    else:
        out["C"] = out["lo"] = out["up"] = None
    return out


def _orc_batch(asm, active=None, sl=None):
    """build the C struct; returns (struct, keepalive)."""
    P = OrcBatch()
    keep = []
    B = asm["B"]
    s = slice(0, B) if sl is None else sl

    def take(a):
        if a is None:
            return None
        a = np.ascontiguousarray(a[s], dtype=np.float64)
        keep.append(a)
        return _p(a)

    P.n, P.L = asm["n"], asm["L"]
    P.B = len(range(*s.indices(B)))
    for k in range(asm["L"]):
        P.m[k], P.ma[k] = asm["m"][k], asm["ma"][k]
        P.A[k] = take(asm["A"][k]); P.b[k] = take(asm["b"][k])
        P.w[k] = take(asm["w"][k]); P.c[k] = take(asm["c"][k])
    P.nc = asm["nc"]
    P.C, P.lo, P.up = take(asm["C"]), take(asm["lo"]), take(asm["up"])
    P.l, P.u = take(asm["l"]), take(asm["u"])
    P.eps_abs = asm["eps_abs"]
    if asm.get("Wdense") is not None:
        for k in range(asm["L"]):
            if asm["Wdense"][k] is not None:
                P.Wd[k] = take(asm["Wdense"][k])
    if asm.get("row_level") is not None:
        rl = np.ascontiguousarray(asm["row_level"], dtype=np.int32)
        keep.append(rl)
        P.row_level = rl.ctypes.data_as(ip)
    reg = asm.get("reg")   # user regularisation task: dict(A [B][mr][n] or None = [I 0], b [B][mr], w scalar)
    if reg is not None:
        P.mr = reg["b"].shape[1]
        P.Ar, P.br, P.wr = take(reg.get("A")), take(reg["b"]), float(reg.get("w", 1.0))
    if active is not None:
        act = (C.c_ubyte * asm["L"])(*[1 if a else 0 for a in active])
        keep.append(act)
        P.active = act
    return P, keep


def cost_function(asm, inst, k, regularised=False):
    """iHQP::getCostFunction / getCostFunctionRegularized of level k"""
    P, keep = _orc_batch(asm)
    n = asm["n"]
    H = np.zeros((n, n)); g = np.zeros(n)
    lib().orc_cost_function(C.byref(P), inst, k, _p(H), _p(g))
    if regularised:
        lib().orc_add_regularisation(C.byref(P), inst, _p(H), _p(g))
    return H, g


def ihqp_solve_batch(asm, backend=BE_EIQP_EQ, nthreads=0, cycles=1, active=None, sl=None,
                     eps_factor=None, termination_tolerance=0.0):
    """returns dict(dq [B][n], x_levels [B][L][n], status [B], seconds, iterations)."""
    L = lib()
    if backend == BE_QPOASES_REF:
        if not L.orc_ref_load(_REF_SO.encode()):
            raise RuntimeError("oracle/_ref/libqpoases_ref.so not available")
        if eps_factor is None:
            eps_factor = asm["eps_abs"] / (1.0e3 * 2.221e-16)
        L.orc_ref_configure(eps_factor, termination_tolerance)
    P, keep = _orc_batch(asm, active, sl)
    B, n, Ln = P.B, asm["n"], asm["L"]
    dq = np.zeros((B, n)); xl = np.zeros((B, Ln, n))
    status = np.zeros(B, dtype=np.int32)
    it = C.c_longlong(0)
    if nthreads <= 0:
        nthreads = os.cpu_count() or 1
    sec = L.orc_ihqp_solve_batch(C.byref(P), backend, nthreads, cycles, _p(dq), _p(xl),
                                 status.ctypes.data_as(ip), C.byref(it))
    return {"dq": dq, "x_levels": xl, "status": status, "seconds": sec, "iterations": it.value,
            "threads": nthreads}


def backend_solve(H, g, A, lA, uA, l, u, eps_abs, form=BE_EIQP_EQ):
    """one QP in BackEnd convention through the restated Goldfarb-Idnani routine."""
    H, g, A, lA, uA, l, u = map(_c, (H, g, A, lA, uA, l, u))
    n = g.shape[0]
    nc = 0 if A is None else A.shape[0]
    x = np.zeros(n)
    it = C.c_int(0)
    ok = lib().orc_backend_solve(form, n, _p(H), _p(g), nc, _p(A), _p(lA), _p(uA), _p(l), _p(u),
                                 eps_abs, _p(x), C.byref(it))
    return bool(ok), x, it.value


def force_gains(J, Bi, Kp, Kd, f=None):
    """GainType::Force of acceleration::Cartesian (acceleration/Cartesian.cpp:161-169, 517-524), restated: per instance
    Mi = J Bi J', Gp = Mi Kp, Gd = Mi Kd and the acceleration Mi f of a virtual force.  J [B][rows][nv], Bi [B][nv][nv].
    -> (Gp [B][rows][rows], Gd, a_add [B][rows] or None)"""
    L = lib()
    B, rows, nv = J.shape
    Gp = np.zeros((B, rows, rows)); Gd = np.zeros((B, rows, rows))
    a = np.zeros((B, rows)) if f is not None else None
    Mi = np.zeros((rows, rows))
    for i in range(B):
        L.orc_cartesian_inertia_inverse(rows, nv, _p(np.ascontiguousarray(J[i])), _p(np.ascontiguousarray(Bi[i])), _p(Mi))
        Gp[i] = Mi @ Kp; Gd[i] = Mi @ Kd
        if f is not None:
            a[i] = Mi @ f[i]
    return Gp, Gd, a
