"""The product's kernel bodies (opensot_amd/csrc/osot_kernels.h, osot_qp_core.h) executed through the host
lock-step emulation in tests/emu and compared with the oracle / golden vectors.  No GPU: this is how the
device algorithm is debugged in the build container; the -m gpu tests repeat the comparisons on hardware."""
import numpy as np
import pytest

from helpers import answer_is_acceptable, default_eps_stuck_instances, emu_cascade, emu_qp, kkt_check, load_golden, random_qp
import helpers
from opensot_amd import synth
from oracle import pyoracle

EPS = 1e3 * 2.221e-16


@pytest.mark.parametrize("cfg", ["C2", "C3", "C4", "C5"])
def test_cascade_vs_golden(cfg, oracle):
    plan, leaf, z = load_golden(cfg)
    asm = oracle.assemble(plan, leaf)
    dq, xl, st, it = emu_cascade(plan, asm)
    assert (st == 0).all()
    ok = z["ok_ref"].astype(bool); okx = z["ok_exact"].astype(bool)
    assert np.abs(dq[ok] - z["x_ref"][ok][:, -1]).max() < 1e-6      # north_star tolerance vs qpOASES
    assert np.abs(dq[okx] - z["x_exact"][okx][:, -1]).max() < 1e-8  # vs qpOASES at tight termination
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
    assert np.abs(dq - ref["dq"]).max() < 1e-10 * max(1.0, np.abs(ref["dq"]).max())   # C5: |x| ~ 1e2


@pytest.mark.parametrize("B", [1, 3])
def test_cascade_ragged_batches(B, oracle):
    """odd batch sizes: the second team of the last wavefront has no instance"""
    plan, leaf = synth.make_velocity_stack("C3", B, seed=77)
    asm = oracle.assemble(plan, leaf)
    dq, xl, st, it = emu_cascade(plan, asm)
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
    assert (st == 0).all()
    assert np.abs(dq - ref["dq"]).max() < 1e-10


@pytest.mark.parametrize("active", [(1, 0, 1), (0, 1, 1), (1, 1, 0), (0, 0, 1)])
def test_inactive_levels(active, oracle):
    """iHQP::setActiveStack (iHQP.cpp:391-395): inactive levels contribute rows 0*x in [-1,1] and are skipped;
    the last ACTIVE level's x is returned (iHQP.cpp:349)"""
    plan, leaf = synth.make_velocity_stack("C3", 4, seed=5)
    asm = oracle.assemble(plan, leaf)
    dq, xl, st, it = emu_cascade(plan, asm, active=active)
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1, active=active)
    assert (st == 0).all() and (ref["status"] == 1).all()
    assert np.abs(dq - ref["dq"]).max() < 1e-10


def test_generic_qp_known_answers():
    """TestQPOases.cpp:208-254, 274-340 through the batched QP kernel body"""
    l = -10 * np.ones((1, 3)); u = 10 * np.ones((1, 3))
    Hm = np.array([[1.0, 1, 1]]); b = np.array([10.0])
    x, st, _ = emu_qp((Hm.T @ Hm)[None], (-Hm.T @ b)[None], np.array([[[1.0, 0, 1]]]), np.array([[20.0]]),
                      np.array([[20.0]]), l, u, eps_abs=EPS * 1e4)
    assert st[0] == 0
    np.testing.assert_allclose(x[0], [10, -10, 10], atol=1e-6)
    for Hm, b, want, tol in [(np.array([[1.0, 1, 1], [0, 1, 1]]), np.array([6.0, 5]), [1, 2.5, 2.5], 1e-6),
                             (np.array([[1.0, 1, 1], [0, 1, 1], [1, 1, 0]]), np.array([6.0, 5, 3]), [1, 2, 3], 1e-6)]:
        x, st, _ = emu_qp((Hm.T @ Hm)[None], (-Hm.T @ b)[None], None, None, None, l, u, eps_abs=EPS)
        assert st[0] == 0
        np.testing.assert_allclose(x[0], want, atol=tol)


@pytest.mark.parametrize("n,nc,n_eq", [(5, 3, 1), (17, 9, 4), (32, 20, 6), (33, 10, 3), (50, 40, 8), (64, 12, 5)])
def test_generic_qp_random_vs_oracle(n, nc, n_eq, oracle):
    """random strictly convex QPs incl. the 64-lane team path (n > 32), vs the oracle and a KKT check"""
    rng = np.random.default_rng(n * 100 + nc)
    B = 6
    H, g, A, lA, uA, l, u = random_qp(rng, B, n, nc, n_eq)
    x, st, it = emu_qp(H, g, A, lA, uA, l, u, eps_abs=1e-9)
    assert (st == 0).all()
    for i in range(B):
        ok, xo, _ = oracle.backend_solve(H[i], g[i], A[i], lA[i], uA[i], l[i], u[i], 1e-9)
        assert ok
        assert np.abs(x[i] - xo).max() < 1e-8
        assert kkt_check(H[i], g[i], A[i], lA[i], uA[i], l[i], u[i], x[i], 1e-9) < 1e-6


def test_dependent_and_redundant_equalities(oracle):
    """duplicate equality rows (linearly dependent but consistent) are skipped; inconsistent ones are
    reported infeasible (qpOASES handles LI via addConstraint_checkLI, QProblem.cpp:2847)"""
    rng = np.random.default_rng(9)
    n = 8
    H = np.eye(n)[None]; g = rng.normal(size=(1, n))
    a = rng.normal(size=n)
    A = np.stack([a, 2 * a, rng.normal(size=n)])[None]
    lA = np.array([[1.0, 2.0, 0.3]]); uA = lA.copy()
    x, st, _ = emu_qp(H, g, A, lA, uA, None, None, eps_abs=0.0)
    assert st[0] == 0
    np.testing.assert_allclose(A[0] @ x[0], lA[0], atol=1e-10)
    lA2 = np.array([[1.0, 2.5, 0.3]])
    x, st, _ = emu_qp(H, g, A, lA2, lA2.copy(), None, None, eps_abs=0.0)
    assert st[0] == 1   # OSOT_STATUS_INFEASIBLE


def test_infeasible_box_vs_row():
    """x0 + x1 >= 5 cannot hold inside the box [-1, 1]^2"""
    H = np.eye(2)[None]; g = np.zeros((1, 2))
    A = np.array([[[1.0, 1.0]]]); lA = np.array([[5.0]]); uA = np.array([[np.inf]])
    x, st, _ = emu_qp(H, g, A, lA, uA, -np.ones((1, 2)), np.ones((1, 2)))
    assert st[0] == 1
    assert (x[0] == 0).all()   # failed instances return 0 (callers zero dq, coman_ik.cpp:189-190)


def test_not_positive_definite():
    H = np.array([[[1.0, 2.0], [2.0, 1.0]]]); g = np.ones((1, 2))
    x, st, _ = emu_qp(H, g, None, None, None, None, None, eps_abs=0.0)
    assert st[0] == 3


def test_infinite_bounds_are_absent():
    """+-inf, +-DBL_MAX and +-1e20 all mean 'no bound' (QPOasesBackEnd::checkINFTY, :339-356)"""
    H = np.eye(2)[None]; g = np.array([[-3.0, 4.0]])
    A = np.array([[[1.0, 0.0], [0.0, 1.0]]])
    lA = np.array([[-np.finfo(float).max, -1e20]]); uA = np.array([[np.inf, 1e20]])
    l = np.array([[-np.inf, -1e30]]); u = np.array([[1e20, np.finfo(float).max]])
    x, st, it = emu_qp(H, g, A, lA, uA, l, u)
    assert st[0] == 0 and it[0] == 0
    np.testing.assert_allclose(x[0], [3, -4], atol=1e-14)


def test_drop_path_is_exercised(oracle):
    """problems whose unconstrained minimiser violates many bounds at once force partial steps / drops"""
    rng = np.random.default_rng(21)
    B, n, nc = 8, 12, 10
    H, g, A, lA, uA, l, u = random_qp(rng, B, n, nc, 2, scale=1.0)
    g *= 10.0
    x, st, it = emu_qp(H, g, A, lA, uA, l, u, eps_abs=1e-9)
    assert (st == 0).all()
    for i in range(B):
        ok, xo, _ = oracle.backend_solve(H[i], g[i], A[i], lA[i], uA[i], l[i], u[i], 1e-9)
        assert ok and np.abs(x[i] - xo).max() < 1e-8


def test_inverse_dynamics_stack_properties(oracle):
    """config 5 through the 64-lane path: InverseDynamics::computedTorque's own check (floating-base rows of
    tau vanish, InverseDynamics.cpp:83-92), torque limits and friction cones hold"""
    plan, leaf = synth.make_id_stack(6, seed=11)
    asm = oracle.assemble(plan, leaf)
    dq, xl, st, it = emu_cascade(plan, asm)
    assert (st == 0).all()
    tau = synth.computed_torque(leaf, dq)
    assert np.abs(tau[:, :6]).max() < 1e-9
    assert np.abs(tau[:, 6:]).max() <= 30.0 + 1e-9
    Cx = np.einsum("brn,bn->br", asm["C"], dq)
    assert (Cx <= np.minimum(asm["up"], 1e20) + 1e-9).all() and (Cx >= np.maximum(asm["lo"], -1e20) - 1e-9).all()
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
    assert np.abs(dq - ref["dq"]).max() < 1e-8


@pytest.mark.parametrize("n,rows,n_eq,n_ineq", [(7, [6], 0, 0), (7, [3, 3], 1, 2), (20, [5, 6], 4, 6), (31, [10, 12], 3, 0),
                                                (33, [8, 10], 2, 4), (35, [3, 12, 16], 3, 5), (38, [10, 14], 3, 6), (39, [10, 14], 3, 6),
                                                (48, [12, 16], 3, 6), (54, [12, 20], 4, 8),
                                                (55, [12, 20], 4, 8), (64, [16, 24], 5, 10)])
def test_small_generic_cascades(n, rows, n_eq, n_ineq, oracle):
    """n < 32 goes through the guarded (FULLN = false) instantiation: Panda-like 7-variable stacks
    (examples/cpp/panda_ik.cpp shape) and mid-size generic stacks with equality and inequality rows; 33 .. 54 variables
    through the 64-lane solver with the short LDS layouts (WaveCtx<40>: 33 .. 38, WaveCtx<56>: 39 .. 54), 55 .. 64 through the full one"""
    plan, leaf = synth.make_generic_stack(5, n, rows, n_eq=n_eq, n_ineq=n_ineq, seed=n)
    asm = oracle.assemble(plan, leaf)
    dq, xl, st, it = emu_cascade(plan, asm)
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
    assert (st == 0).all() and (ref["status"] == 1).all()
    assert np.abs(dq - ref["dq"]).max() < 1e-9
    if oracle.ref_available():
        rq = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1)
        assert (rq["status"] == 1).all() and np.abs(dq - rq["dq"]).max() < 1e-6


@pytest.mark.parametrize("rows,n_eq,n_ineq,dup,eps", [([3, 12], 12, 0, None, 1e6), ([3, 6, 6], 12, 0, None, 1e6), ([5, 12], 12, 0, None, 1e6),
                                                       ([3, 12], 12, 4, None, 1e6), ([3, 12], 6, 0, 0, 1e6),
                                                       # the DENSE levels' null-space method (nullspace_dense_wide) with the bounds and inequality rows at work,
                                                       # a 23-column reduced Hessian, a 20-row level, and the reference's default eps (factor 200)
                                                       ([8, 10], 2, 4, None, 1e6), ([6, 20], 8, 3, None, 1e6), ([3, 12], 12, 4, None, 2e2), ([3, 6, 6], 12, 0, None, 2e2),
                                                       # ... and at the FIRST level (no x_prev: the global rows' right-hand sides ride along in the elimination):
                                                       # S2's shape (15 rows over the feet), one dense level, dependent consistent rows among the global ones
                                                       ([15], 12, 0, None, 1e6), ([15], 12, 3, None, 2e2), ([20], 14, 0, None, 1e6)])
def test_nullspace_elimination_on_the_40_lane_layout(rows, n_eq, n_ineq, dup, eps, oracle):
    """round 6 -- nullspace_equalities_wide: the Postural last level of the reference's COMAN stacks (coman_ik.cpp:425-449; 35
    coordinates: the 40-lane layout) under its 27 equality rows -- the feet as global equality rows (12), the CoM level (3), the wrists
    (12) -- by Gauss-Jordan in 2 x 3 tiles instead of 27 reflections of the full J.  S3's and S4's shapes, 29 rows (six free columns),
    with inequality rows beside the box (the general instantiation), and -- last case -- 27 rows of rank 21: fourteen free columns, more
    than the elimination carries, so it hands the level back to the generic path with J' restored.  Against the restated eiQuadProg
    cascade (1e-9) and the reference's qpOASES (1e-6)."""
    plan, leaf = synth.make_generic_stack(6, 35, rows, n_eq=n_eq, n_ineq=n_ineq, seed=61 + len(rows) + n_ineq, box=(0.3 if n_eq == 12 else 0.05),
                                          duplicate_eq_in_level=dup, eps_factor=eps)
    asm = oracle.assemble(plan, leaf)
    dq, xl, st, it = emu_cascade(plan, asm)
    assert (st == 0).all()
    res = np.einsum("bij,bj->bi", asm["C"][:, :n_eq], dq) - asm["lo"][:, :n_eq]
    assert np.abs(res).max() < 1e-9
    if dup is None:
        ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
        assert (ref["status"] == 1).all() and np.abs(dq - ref["dq"]).max() < 1e-9
    if oracle.ref_available():
        rq = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1)
        ok = rq["status"] == 1
        assert ok.sum() >= 5 and np.abs(dq[ok] - rq["dq"][ok]).max() < 1e-6


def test_roundoff_violation_with_no_freedom_left(oracle):
    """an inequality active at an upper level re-appears at a level whose optimality rows leave no free direction:
    its slack is O(eps*cond) and must not be reported as infeasibility (instance 54 of this seeded family did)"""
    plan, leaf = synth.make_generic_stack(64, 7, [3, 3], n_eq=1, n_ineq=2, seed=7)
    asm = oracle.assemble(plan, leaf)
    dq, xl, st, it = emu_cascade(plan, asm)
    assert (st == 0).all()
    if oracle.ref_available():
        rq = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1)
        ok = rq["status"] == 1
        assert np.abs(dq[ok] - rq["dq"][ok]).max() < 1e-6


@pytest.mark.parametrize("n,rows", [(16, [4, 5]), (18, [2])])
def test_optimality_rows_duplicating_global_equalities(n, rows, oracle):
    """coman_ik.cpp:442 situation: the stack's global equality rows re-appear as optimality rows of a task level
    (more equality rows than could be independent).  Consistent duplicates are skipped; the answer equals the
    reference's (qpOASES drops them through its linear-independence test, QProblem.cpp:2847).
    (18, [2]): 14 equality rows of rank 8 at the Postural level: the null-space elimination starts (18 - 14 <= 8), finds ten
    free columns, more than it carries, and hands the level back to the generic path -- with J' restored in the LDS matrix
    it used as its column store."""
    plan, leaf = synth.make_generic_stack(6, n, rows, n_eq=6, seed=3, duplicate_eq_in_level=0)
    asm = oracle.assemble(plan, leaf)
    dq, xl, st, it = emu_cascade(plan, asm)
    assert (st == 0).all()
    # (the restated eiQuadProg routine, like the reference's own eiQuadProg back-end, gives up on linearly
    #  dependent equalities -- eiquadprog.hpp:246-251 "FIXME" -- so the comparison is with qpOASES only)
    if oracle.ref_available():
        rq = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1)
        assert (rq["status"] == 1).all() and np.abs(dq - rq["dq"]).max() < 1e-6
    Ceq = asm["C"][:, :6]; e = asm["lo"][:, :6]
    assert np.abs(np.einsum("brn,bn->br", Ceq, dq) - e).max() < 1e-10
    for k in range(1, plan.L):   # hierarchy: A_j x_k = A_j x_j
        for j in range(k):
            assert np.abs(np.einsum("brn,bn->br", asm["A"][j], xl[:, k] - xl[:, j])).max() < 1e-9


@pytest.mark.parametrize("kw", [dict(m=3), dict(m=4, weight=2.5), dict(m=3, postural_weight=1e-3),
                                dict(m=3, dependent=True), dict(m=4, zero_row=True, postural_weight=0.05),
                                dict(m=1, second_level_rows=0), dict(m=2, eps_factor=2e2, postural_weight=1e-3)])
def test_lowrank_levels(kw, oracle):
    """levels with <= 4 stored rows take the closed-form path (lowrank_prepare32: Gram-Schmidt of the rows, no H,
    no factorisation): weights, a Postural block in the same level, a dependent row, a zero row, the default eps"""
    n = 12 if kw.get("m", 3) != 4 else 32
    plan, leaf = synth.make_lowrank_stack(6, n, seed=11, **kw)
    asm = oracle.assemble(plan, leaf)
    dq, xl, st, it = emu_cascade(plan, asm)
    assert (st == 0).all()
    # dependent equality rows, or more equality rows than variables at the second level (a Postural block in the
    # first level leaves m + n optimality rows): outside what the restated eiQuadProg routine supports
    degenerate = (kw.get("dependent") or kw.get("zero_row")
                  or (kw.get("postural_weight") is not None and kw.get("second_level_rows", 5) != 0))
    if not degenerate:   # (the restated eiQuadProg routine mishandles linearly dependent equality rows, like the
        #                   reference's own: eiquadprog.hpp:246-251 "FIXME"; those cases are pinned by qpOASES only)
        ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
        assert (ref["status"] == 1).all()
        assert np.abs(dq - ref["dq"]).max() < (1e-9 if kw.get("eps_factor", 1e6) == 1e6 else 1e-7)
    if oracle.ref_available():
        rq = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1)
        ok = rq["status"] == 1
        assert ok.all() and np.abs(dq[ok] - rq["dq"][ok]).max() < 1e-6
    else:
        assert not degenerate, "degenerate cases need oracle/_ref (qpOASES)"


@pytest.mark.parametrize("n,rows", [(32, [45]), (32, [33, 7]), (20, [37])])
def test_more_rows_than_variables(n, rows, oracle):
    """over-determined levels: more stored rows than the 32 the H build requests per round of loads (and a row
    count that is not a multiple of the four rows one MFMA step takes)"""
    plan, leaf = synth.make_generic_stack(4, n, rows, seed=5, postural_last=False)
    asm = oracle.assemble(plan, leaf)
    dq, xl, st, it = emu_cascade(plan, asm)
    assert (st == 0).all()
    if rows[0] <= n or len(rows) == 1:
        ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
        assert (ref["status"] == 1).all() and np.abs(dq - ref["dq"]).max() < 1e-9
    elif oracle.ref_available():
        # more optimality (equality) rows than variables at the second level: outside what the restated eiQuadProg
        # routine supports (its working set is sized n, like the reference's); pinned by qpOASES
        rq = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1)
        assert (rq["status"] == 1).all() and np.abs(dq - rq["dq"]).max() < 1e-6


def test_last_free_direction_is_not_called_dependent(oracle):
    """found by tests/stress_parity.py: default eps (4.4e-11), 21 of 22 directions taken, a violated inequality whose
    normal has |d2|^2 = 6e-19 |d|^2 in the last free direction (|d|^2 is dominated by the 1/eps-scaled directions).
    A dependency threshold of 1e-18 |d|^2 called it dependent and the instance INFEASIBLE; the reference solves it."""
    kw = {'n_eq': 3, 'n_ineq': 5, 'seed': 201675281, 'box': 0.5, 'postural_last': False, 'eps_factor': 200.0}
    plan, leaf = synth.make_generic_stack(192, 22, [12, 3, 13], **kw)
    asm = oracle.assemble(plan, leaf)
    i = 140
    sub = {k: (v[i:i + 1] if isinstance(v, np.ndarray) and v.shape[:1] == (192,) else v) for k, v in asm.items()}
    for name in ("A", "b", "w", "c"):
        sub[name] = [None if a is None else a[i:i + 1] for a in asm[name]]
    sub["B"] = 1
    dq, xl, st, it = emu_cascade(plan, sub)
    assert (st == 0).all()
    if oracle.ref_available():
        rq = oracle.ihqp_solve_batch(sub, oracle.BE_QPOASES_REF, nthreads=1)
        assert (rq["status"] == 1).all() and np.abs(dq - rq["dq"]).max() < 2e-6


def test_subtasks(oracle):
    """SubTask (src/tasks/SubTask.cpp:22-112): CoM x,y with its own lambda, position-only wrists, a Postural sub-task on
    the actuated joints (stored unit rows): assembly by the oracle, cascade on the emulator, against both oracles"""
    plan, leaf = synth.make_subtask_stack(6, seed=2)
    assert [plan.m(k) for k in range(3)] == [2, 18, 26] and [plan.ma(k) for k in range(3)] == [2, 18, 26]
    asm = oracle.assemble(plan, leaf)
    # the CoM sub-task: rows 0, 1 of lambda (p_d - p), times the sub-task lambda
    p, pd, _ = leaf["task"][0][0]
    assert np.abs(asm["b"][0] - 0.7 * 0.1 * (pd - p)[:, :2]).max() < 1e-16
    dq, xl, st, it = emu_cascade(plan, asm)
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
    assert (st == 0).all() and (ref["status"] == 1).all() and np.abs(dq - ref["dq"]).max() < 1e-9
    if oracle.ref_available():
        rq = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1)
        ok = rq["status"] == 1
        assert ok.all() and np.abs(dq[ok] - rq["dq"][ok]).max() < 1e-6


def _reg_cases():
    from opensot_amd import abi
    return [("C3", abi.TASK_GENERIC, None, 1e-2), ("C3", abi.TASK_POSTURAL, None, 0.5), ("C2", abi.TASK_GENERIC, 20, 1e-3),
            ("generic40", abi.TASK_POSTURAL, None, 1e-2), ("lowrank", abi.TASK_GENERIC, None, 0.2), ("lowrank_plain", abi.TASK_POSTURAL, 20, 0.05), ("generic", abi.TASK_POSTURAL, 5, 1.0),
            ("id", abi.TASK_ACC_POSTURAL, 32, 1e-2)]


def _reg_stack(name, B, seed):
    if name == "generic40":     # 40 variables: the NP = 64 instantiation
        return synth.make_generic_stack(B, 40, [10, 12], n_eq=2, n_ineq=3, seed=seed)
    if name in ("C2", "C3"):
        return synth.make_velocity_stack(name, B, seed=seed)
    if name == "lowrank":
        return synth.make_lowrank_stack(B, 32, m=3, seed=seed, postural_weight=1e-3, second_level_rows=5)
    if name == "lowrank_plain":
        return synth.make_lowrank_stack(B, 32, m=4, seed=seed, second_level_rows=5)
    if name == "generic":
        return synth.make_generic_stack(B, 7, [3, 3], n_eq=1, n_ineq=2, seed=seed, postural_last=False)
    return synth.make_id_stack(B, seed=seed)


@pytest.mark.parametrize("name,kind,rows,weight", _reg_cases())
def test_user_regularisation_task(name, kind, rows, weight, oracle):
    """AutoStack::setRegularisationTask (AutoStack.h:78-92): the cost of an identity-Jacobian task is added to every
    level (iHQP.cpp:265-266, 274-278) and never becomes an optimality row; all H-build paths of the kernel (matrix
    core, closed-form low-rank, diagonal, NP = 64) on the emulator against the oracle's general H += Hr, g += gr"""
    plan, leaf = _reg_stack(name, 4, seed=5)
    # (dependent equality rows / more equality rows than variables: outside the restated eiQuadProg routine, qpOASES only)
    qpoases_only = name in ("id", "lowrank")
    base = None if qpoases_only else oracle.ihqp_solve_batch(oracle.assemble(plan, leaf), oracle.BE_EIQP_EQ, nthreads=1)["dq"]
    synth.add_regularisation(plan, leaf, kind=kind, rows=rows, weight=weight, seed=3)
    asm = oracle.assemble(plan, leaf)
    assert asm["reg"]["b"].shape == (4, plan.regularisation.rows)
    dq, xl, st, it = emu_cascade(plan, asm)
    assert (st == 0).all()
    if not qpoases_only:
        ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
        assert (ref["status"] == 1).all()
        assert np.abs(dq - ref["dq"]).max() < 1e-9 * max(1.0, np.abs(ref["dq"]).max())
        assert np.abs(ref["dq"] - base).max() > 1e-6      # the task does change the answer
    if oracle.ref_available():
        rq = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1)
        ok = rq["status"] == 1
        assert ok.all() and np.abs(dq[ok] - rq["dq"][ok]).max() < 1e-6 * max(1.0, np.abs(rq["dq"]).max())
    else:
        assert not qpoases_only, "this case needs oracle/_ref (qpOASES)"


def _dense_reg_cases():
    from opensot_amd import abi
    # (stack, kind of the regularisation task, rows, weight): the matrix-core H build with and without stored level rows
    # (a Postural-only level is no longer diagonal), the levels that would take the closed-form low-rank path, the 64-lane
    # instantiation
    return [("C3", abi.TASK_CARTESIAN, None, 1e-2), ("C3", abi.TASK_GENERIC, 9, 0.3), ("C2", abi.TASK_COM, None, 5e-2),
            ("lowrank_plain", abi.TASK_GENERIC, 5, 0.1), ("generic40", abi.TASK_CARTESIAN, None, 1e-2), ("generic", abi.TASK_GENERIC, 4, 1.0)]


@pytest.mark.parametrize("name,kind,rows,weight", _dense_reg_cases())
def test_regularisation_task_with_a_stored_jacobian(name, kind, rows, weight, oracle):
    """a regularisation task with a DENSE Jacobian (round 3; AutoStack::setRegularisationTask takes any task,
    AutoStack.h:78-92; iHQP.cpp:265-278 adds its H and g to every level): H += w A_r'A_r, g -= w A_r'b_r through the
    H build of every path, against the oracle's general form and the reference's qpOASES"""
    plan, leaf = _reg_stack(name, 4, seed=6)
    base = oracle.ihqp_solve_batch(oracle.assemble(plan, leaf), oracle.BE_EIQP_EQ, nthreads=1)["dq"]
    synth.add_regularisation(plan, leaf, kind=kind, rows=rows, weight=weight, seed=4, dense=True)
    asm = oracle.assemble(plan, leaf)
    assert asm["reg"]["A"].shape == (4, plan.regularisation.rows, plan.n)
    dq, xl, st, it = emu_cascade(plan, asm)
    assert (st == 0).all()
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
    assert (ref["status"] == 1).all()
    assert np.abs(dq - ref["dq"]).max() < 1e-9 * max(1.0, np.abs(ref["dq"]).max())
    assert np.abs(ref["dq"] - base).max() > 1e-6          # the task does change the answer
    if oracle.ref_available():
        rq = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1)
        ok = rq["status"] == 1
        assert ok.all() and np.abs(dq[ok] - rq["dq"][ok]).max() < 1e-6


@pytest.mark.parametrize("n,rows,local_level,n_local", [(7, [3, 3], 0, 2), (20, [5, 6], 1, 4), (31, [10, 12], 2, 3), (40, [10, 12], 0, 5)])
def test_task_local_constraint_rows(n, rows, local_level, n_local, oracle):
    """`task << constraint` (Task::getConstraints(), iHQP.cpp:190, 282-287): the rows constrain the QP of THEIR level
    only; the lower levels see that level's optimality rows, not its local constraints"""
    mk = lambda lvl: synth.make_generic_stack(5, n, rows, n_eq=1, n_ineq=2, seed=9, n_local=n_local, local_level=lvl)
    plan, leaf = mk(local_level)
    asm = oracle.assemble(plan, leaf)
    assert (asm["row_level"] == [0] * 3 + [local_level + 1] * n_local).all()
    dq, xl, st, it = emu_cascade(plan, asm)
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
    assert (st == 0).all() and (ref["status"] == 1).all()
    assert np.abs(dq - ref["dq"]).max() < 1e-9 and np.abs(xl - ref["x_levels"]).max() < 1e-9
    if oracle.ref_available():
        rq = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1)
        ok = rq["status"] == 1
        assert ok.all() and np.abs(dq[ok] - rq["dq"][ok]).max() < 1e-6
    # the local rows hold at their own level ...
    Cl, lo, up = leaf["rows"][-1]
    v = np.einsum("bri,bi->br", Cl, xl[:, local_level])
    assert (v >= lo - 1e-9).all() and (v <= up + 1e-9).all()
    # ... and the tag matters: the same rows as GLOBAL rows give a different cascade
    plan_g, leaf_g = mk(None)
    glob = oracle.ihqp_solve_batch(oracle.assemble(plan_g, leaf_g), oracle.BE_EIQP_EQ, nthreads=1)
    assert np.abs(glob["x_levels"] - ref["x_levels"]).max() > 1e-6


@pytest.mark.parametrize("n,rows", [(12, [9]), (16, [5, 6]), (32, [10, 17]), (32, [3, 24])])
def test_task_local_equality_on_a_postural_last_level(n, rows, oracle):
    """a task-local EQUALITY at a Postural last level (diagonal Hessian, many optimality rows): the previous level's
    solution does not satisfy that row, so the null-space shortcut -- which projects from x_prev and assumes every
    equality holds there -- must not be taken (it used to return SOLVED with the row violated by 0.1 .. 0.3)"""
    plan, leaf = synth.make_generic_stack(6, n, rows, n_eq=0, n_ineq=2, seed=21, n_local=1, local_level=len(rows), local_equality=True)
    asm = oracle.assemble(plan, leaf)
    dq, xl, st, it = emu_cascade(plan, asm)
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
    assert (st == 0).all() and (ref["status"] == 1).all()
    Cl, lo, up = leaf["rows"][-1]
    assert np.abs(np.einsum("bri,bi->br", Cl, dq) - lo).max() < 1e-9      # the local equality holds at its level
    assert np.abs(dq - ref["dq"]).max() < 1e-9
    if oracle.ref_available():
        rq = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1, termination_tolerance=10 * 2.221e-16)
        ok = rq["status"] == 1
        assert ok.any() and np.abs(dq[ok] - rq["dq"][ok]).max() < 1e-7


@pytest.mark.parametrize("mode", ["tasks", "ttc", "ttc_exchange"])
def test_default_eps_stuck_instances(mode, oracle):
    """the five closed-loop instances at iHQP's default eps (factor 2e2) that round 1 reported INFEASIBLE: every level's
    optimality rows are now posed relative to the previous level's solution, which is therefore an exactly feasible point
    of the level (osot_qp_core.h: kFeasMargin); all five solve, and each answer is within 1e-6 of a witness or feasible
    and lexicographically not worse than the witnesses (which disagree with each other by up to 2e-2 here)"""
    plan, asm = default_eps_stuck_instances(mode)
    dq, xl, st, it = emu_cascade(plan, asm)
    assert (st == 0).all()
    if oracle.ref_available():
        rx = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1, termination_tolerance=10 * 2.221e-16)
        rd = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1)
        re_ = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
        for i in range(asm["B"]):
            ok, why = answer_is_acceptable(asm, i, dq[i], [("qpOASES exact", rx["dq"][i], rx["status"][i] == 1),
                                                            ("qpOASES", rd["dq"][i], rd["status"][i] == 1),
                                                            ("eiQuadProg", re_["dq"][i], re_["status"][i] == 1)])
            assert ok, (i, why)


def test_accepted_slack_instance():
    """tests/golden/default_eps_accepted_slack_instance.npz on the emulator (the hardware's round-off is what trips it; see
    tests/test_gpu_cascade.py::test_accepted_slack_instance_gpu)"""
    from helpers import accepted_slack_instance
    plan, asm, wit = accepted_slack_instance()
    dq, xl, st, it = emu_cascade(plan, asm)
    assert st[0] == 0
    ok, why = answer_is_acceptable(asm, 0, dq[0], wit)
    assert ok, why
    assert emu_cascade.last_accepted_slack[0] <= 1.0e-7


def test_collision_instance_from_the_closed_loop(oracle):
    """the instance of tests/test_gpu_cascade.py::test_noise_is_not_a_direction_gpu on the emulator (exact IEEE division
    and square roots: the round-off pattern that tripped the hardware does not arise here, the parity check does)"""
    from test_gpu_cascade import _collision_last_direction_instance
    plan, asm = _collision_last_direction_instance()
    dq, xl, st, it = emu_cascade(plan, asm)
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
    assert st[0] == 0 and ref["status"][0] == 1 and np.abs(dq - ref["dq"]).max() < 1e-9
    if oracle.ref_available():
        # (qpOASES at OpenSoT's options stops 2e-2 from the optimum here, its box violated by 1e-7; run to the exact
        # optimum it agrees)
        rx = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1, termination_tolerance=10 * 2.221e-16)
        assert rx["status"][0] == 1 and np.abs(dq - rx["dq"]).max() < 1e-6


@pytest.mark.parametrize("n,rows,level", [(7, [3, 3], 0), (31, [10, 12], 0), (40, [10, 12], 0)])
def test_task_local_bounds_as_unit_rows(n, rows, level, oracle):
    """`task << bound` (iHQP merges a level's own bounds into that level's box only, iHQP.cpp:190, 336-340) as a
    level-tagged block of unit rows (OSOT_ROWS_UNIT_GENERIC, no storage): it holds at its level and only there; the
    same block without a tag is the global box.  (A tight local box at a LOWER level is infeasible by construction:
    that level must keep A_0 x = A_0 x_0 with an x_0 that was free to be large -- product and oracle agree on that.)"""
    hw = 0.01
    plan_i, leaf_i = synth.make_generic_stack(5, 20, [5, 6], n_eq=1, n_ineq=2, seed=4, box=0.5, unit_box=(1, hw))
    asm_i = oracle.assemble(plan_i, leaf_i)
    assert (emu_cascade(plan_i, asm_i)[2] == 1).all() and (oracle.ihqp_solve_batch(asm_i, oracle.BE_EIQP_EQ, nthreads=1)["status"] == 0).all()
    plan, leaf = synth.make_generic_stack(5, n, rows, n_eq=1, n_ineq=2, seed=4, box=0.5, unit_box=(level, hw))
    asm = oracle.assemble(plan, leaf)
    dq, xl, st, it = emu_cascade(plan, asm)
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
    assert (st == 0).all() and (ref["status"] == 1).all()
    assert np.abs(dq - ref["dq"]).max() < 1e-9 and np.abs(xl - ref["x_levels"]).max() < 1e-9
    if oracle.ref_available():
        rq = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1)
        assert (rq["status"] == 1).all() and np.abs(dq - rq["dq"]).max() < 1e-6
    assert np.abs(xl[:, level]).max() <= hw + 1e-9 and np.abs(xl[:, level]).max() > hw - 1e-9     # binding at its level
    assert np.abs(xl).max() > 2 * hw                                                              # ... and not elsewhere
    # untagged, the unit rows are a global box: same answer as the plan's own box of that width
    plan_g, leaf_g = synth.make_generic_stack(5, n, rows, n_eq=1, n_ineq=2, seed=4, box=0.5, unit_box=(None, hw))
    plan_b, leaf_b = synth.make_generic_stack(5, n, rows, n_eq=1, n_ineq=2, seed=4, box=hw)
    xg = emu_cascade(plan_g, oracle.assemble(plan_g, leaf_g))[0]
    xb = emu_cascade(plan_b, oracle.assemble(plan_b, leaf_b))[0]
    assert np.abs(xg - xb).max() < 1e-9


def test_hot_start_matches_cold_start_emulated():
    """hot start (osot_solver_set_hotstart; reference: QPOasesBackEnd.cpp:258-285, SQProblem.cpp:149-193): every level's
    working set of the previous cycle is re-added with signed steps, wrong guesses are taken out by the reverse of an
    addition.  Same answer as the cold start to round-off (unique minimiser per level); on an EXACT repeat of a cycle the
    hot list is the final active set, so no constraint is added twice or dropped: never more iterations than cold."""
    B = 48
    plan, leaf = synth.make_velocity_stack("C4", B, seed=4100)
    rng = np.random.default_rng(5)
    leaves = [leaf, synth.perturb(leaf, rng, 0.01)]
    hot = np.full((B, plan.L, 32), -1, dtype=np.int32)
    for i, lf in enumerate(leaves + [leaves[1]]):
        asm = pyoracle.assemble(plan, lf)
        dq0, _, st0, it0 = helpers.emu_cascade(plan, asm)
        dq1, _, st1, it1 = helpers.emu_cascade(plan, asm, hot=hot)
        assert (st0 == 0).all() and (st1 == 0).all()
        assert np.abs(dq0 - dq1).max() < 1e-9
        if i == 0:
            assert (it0 == it1).all()                      # an empty hot list IS a cold start
            assert (hot >= 0).any()                        # ... and the final working sets were recorded
        if i == 2:                                         # exact repeat of cycle 1: the hot list is the final active set
            assert (it1 <= it0).all() and it1.sum() < it0.sum()
    # a hot list that is plain wrong (every lower bound of level 0, garbage codes) only costs iterations
    hot[:, 0, :] = np.arange(32, dtype=np.int32)[None, :]
    hot[:, 1, :4] = np.array([10 ** 6, -7, 2 * 32 + 2 * 500, 63], dtype=np.int32)
    asm = pyoracle.assemble(plan, leaves[0])
    dq0, _, st0, _ = helpers.emu_cascade(plan, asm)
    dq1, _, st1, _ = helpers.emu_cascade(plan, asm, hot=hot)
    assert (st1 == 0).all() and np.abs(dq0 - dq1).max() < 1e-9


def test_hot_start_inverse_dynamics_emulated():
    """the same on the 64-lane instantiation (config 5: unit-row and stored-row inequalities, equalities ahead of them)"""
    B = 6
    plan, leaf = synth.make_id_stack(B, seed=5100)
    rng = np.random.default_rng(6)
    leaves = [leaf, synth.perturb(leaf, rng, 0.01)]
    hot = np.full((B, plan.L, 64), -1, dtype=np.int32)
    for lf in leaves + [leaves[1]]:
        asm = pyoracle.assemble(plan, lf)
        dq0, _, st0, it0 = helpers.emu_cascade(plan, asm)
        dq1, _, st1, it1 = helpers.emu_cascade(plan, asm, hot=hot)
        assert (st0 == st1).all()
        ok = st0 == 0
        assert ok.any() and np.abs(dq0[ok] - dq1[ok]).max() < 1e-8 * max(1.0, np.abs(dq0[ok]).max())
    assert (it1[ok] <= it0[ok]).all()


@pytest.mark.parametrize("cfg", ["C2", "C3"])
def test_box_instantiation_matches_the_general_one(cfg, monkeypatch):
    """plans without constraint rows (BASELINE configs 2 and 3: the bounds are the only inequalities) run the BOX instantiation
    of the cascade on the device (osot_solver_set_specialisation, default on; osot_qp_core.h: gi_inequalities).  On the
    emulator: the same instances through the BOX instantiation and through the full one agree bit for bit (dq, every level's
    x, status, iteration counts), and the golden answers hold for both"""
    plan, leaf = synth.make_velocity_stack(cfg, 24, seed=77)
    po = pyoracle
    asm = po.assemble(plan, leaf)
    full = emu_cascade(plan, asm)
    assert not emu_cascade.ran_box
    monkeypatch.setenv("OSOT_EMU_BOX", "1")
    box = emu_cascade(plan, asm)
    assert emu_cascade.ran_box
    for a, b in zip(full, box):
        assert np.array_equal(a, b)
    assert (box[2] == 0).all()
    ref = po.ihqp_solve_batch(asm, po.BE_EIQP_EQ, nthreads=1)
    ok = ref["status"] == 1
    assert ok.all() and np.abs(box[0] - ref["dq"]).max() < 1e-8


@pytest.mark.parametrize("n", [32, 40])
def test_box_instantiation_for_rows_that_are_all_equalities(n, monkeypatch):
    """round 4: a plan whose constraint rows are all TaskToConstraint blocks with a POINT band (`stack << l_sole` with
    err_lb = err_ub: the reference's COMAN stacks, coman_ik.cpp:425-449) has no inequality but the bounds, whatever its size:
    the rows are equalities of every level (the update writes lo = b + err_lb, up = b + err_ub: bit-equal), so it runs the
    BOX instantiation too (32-, 56- and 64-lane kernels).  On the emulator: bit-identical to the general instantiation, and
    the rows hold in the answer; a block with a real band keeps the plan on the general path."""
    from opensot_amd import abi
    from opensot_amd.plan import Rows, StackPlan
    B = 12
    if n == 32:
        plan, leaf = synth.make_velocity_stack("C3", B, seed=91)
    else:
        plan, leaf = synth.make_generic_stack(B, n, [6, 12, 8], n_eq=0, n_ineq=0, seed=92, box=0.4)
    rng = np.random.default_rng(5)
    J = rng.normal(0.0, 0.3, size=(B, 3, n))
    pa = rng.uniform(-0.2, 0.2, size=(B, 3))
    pd = pa + rng.uniform(-0.01, 0.01, size=(B, 3))

    def with_rows(band):
        rb = Rows(abi.ROWS_TASK_COM, 3, lam=0.1, err_lb=-band, err_ub=band, name="com_rows")
        pl = StackPlan(n=plan.n, levels=plan.levels, bounds=plan.bounds, rowblocks=list(plan.rowblocks) + [rb], eps_abs=plan.eps_abs)
        lf = dict(leaf)
        lf["rows"] = list(leaf.get("rows", [])) + [(pa, pd, None)]
        lf["C"] = list(leaf.get("C", [])) + [J]
        return pl, lf

    po = pyoracle
    pl, lf = with_rows(0.0)
    asm = po.assemble(pl, lf)
    assert np.array_equal(asm["lo"], asm["up"])
    full = emu_cascade(pl, asm)
    assert not emu_cascade.ran_box
    monkeypatch.setenv("OSOT_EMU_BOX", "1")
    box = emu_cascade(pl, asm)
    assert emu_cascade.ran_box
    for a, b in zip(full, box):
        assert np.array_equal(a, b)
    assert (box[2] == 0).all()
    res = np.einsum("bij,bj->bi", asm["C"], box[0]) - asm["lo"]
    assert np.abs(res).max() < 1e-9
    pl2, lf2 = with_rows(0.01)                     # a real band: inequalities -> the general instantiation, knob or not
    emu_cascade(pl2, po.assemble(pl2, lf2))
    assert not emu_cascade.ran_box



@pytest.mark.parametrize("n", [8, 32, 40, 64])
def test_indefinite_hessian_is_reported_not_solved(n):
    """OSOT_STATUS_NOT_PD: the Cholesky of H + eps I breaks down (a non-positive pivot) -- both tile factorisations (factor_tiles32 for
    n <= 32, round 5's factor_tiles_wide for 33 .. 64) flag it, return x = 0 and do not touch the instances beside it"""
    rng = np.random.default_rng(n)
    B = 4
    M = rng.normal(size=(B, n, n))
    H = M @ np.transpose(M, (0, 2, 1)) + 0.5 * np.eye(n)
    H[1] = H[1] - 3.0 * np.linalg.eigvalsh(H[1]).max() * np.eye(n) * (np.arange(n) == n // 2)     # one negative pivot in the middle
    H[3, n - 1, n - 1] = -1.0                                                                     # ... and one in the last column
    g = rng.normal(size=(B, n))
    x, st, it = emu_qp(H, g, None, None, None, None, None, eps_abs=1e-9)
    assert list(st) == [0, 3, 0, 3]
    assert np.abs(x[1]).max() == 0.0 and np.abs(x[3]).max() == 0.0
    for i in (0, 2):
        assert np.abs(x[i] + np.linalg.solve(H[i] + 1e-9 * np.eye(n), g[i])).max() < 1e-9
