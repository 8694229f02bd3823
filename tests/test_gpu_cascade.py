"""GPU parity of the iHQP cascade kernel (through the C-ABI) against the oracle on seeded stacks."""
import numpy as np
import pytest
import torch

from helpers import judge_remainder
from opensot_amd import synth
from opensot_amd.solver import BatchedStack

pytestmark = pytest.mark.gpu


def _run(plan, leaf, use_update=True, asm=None, active=None):
    B = leaf["B"]
    st = BatchedStack(plan, B, device=0)
    if use_update:
        dev = st.load_leaf(leaf)
        st.update(dev)
    else:
        st.load_assembled(asm)
    st.level_active = active
    st.solve(B)
    torch.cuda.synchronize()
    return (st.dq[:B].cpu().numpy(), st.x_levels[:B].cpu().numpy(), st.status[:B].cpu().numpy(),
            st.iterations[:B].cpu().numpy(), st)


@pytest.mark.parametrize("cfg,B", [("C2", 96), ("C3", 96), ("C4", 96), ("C3", 1), ("C3", 3)])
def test_cascade_matches_oracle(cfg, B, oracle, gpu_device):
    plan, leaf = synth.make_velocity_stack(cfg, B, seed=123 + B)
    asm = oracle.assemble(plan, leaf)
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
    dq, xl, status, it, _ = _run(plan, leaf)
    assert (status == 0).all()
    assert (ref["status"] == 1).all()
    # fp64 tolerance: 1e-9 against the oracle's exact active-set solution (north_star asks 1e-6 vs qpOASES)
    assert np.abs(dq - ref["dq"]).max() < 1e-9
    assert np.abs(xl - ref["x_levels"]).max() < 1e-7   # intermediate levels are eps-conditioned


def test_update_kernel_matches_oracle_assembly(oracle, gpu_device):
    plan, leaf = synth.make_velocity_stack("C4", 64, seed=5)
    asm = oracle.assemble(plan, leaf)
    st = BatchedStack(plan, 64, device=0)
    st.update(st.load_leaf(leaf))
    torch.cuda.synchronize()
    for k in range(plan.L):
        np.testing.assert_allclose(st.b[k].cpu().numpy(), asm["b"][k], rtol=0, atol=1e-15)
        np.testing.assert_array_equal(st.w[k].cpu().numpy(), asm["w"][k])
    np.testing.assert_array_equal(st.l.cpu().numpy(), asm["l"])
    np.testing.assert_array_equal(st.u.cpu().numpy(), asm["u"])
    np.testing.assert_array_equal(st.C.cpu().numpy(), asm["C"])
    np.testing.assert_array_equal(st.lo.cpu().numpy(), asm["lo"])
    np.testing.assert_array_equal(st.up.cpu().numpy(), asm["up"])


@pytest.mark.parametrize("n,rows,n_eq,n_ineq,dup", [(7, [6], 0, 0, None), (7, [3, 3], 1, 2, None), (20, [5, 6], 4, 6, None),
                                                     (16, [4, 5], 6, 0, 0), (18, [2], 6, 0, 0),
                                                     (33, [8, 10], 2, 4, None), (48, [12, 16], 3, 6, None), (54, [12, 20], 4, 8, None),
                                                     (55, [12, 20], 4, 8, None), (64, [16, 24], 5, 10, None),
                                                     # round 6 -- the 40-lane layout's null-space elimination and closed-form low-rank level: the reference's
                                                     # COMAN S3 / S4 shapes (12 global equality rows = the feet, 35 variables, 27 equality rows at the Postural
                                                     # level), 29 rows, inequality rows beside them, and the rank-deficient hand-back to the generic path
                                                     (35, [3, 12], 12, 0, None), (35, [3, 6, 6], 12, 0, None), (35, [5, 12], 12, 0, None),
                                                     (35, [3, 12], 12, 4, None), (35, [3, 12], 6, 0, 0), (38, [3, 14], 12, 2, None)])
def test_small_generic_cascades_gpu(n, rows, n_eq, n_ineq, dup, oracle, gpu_device):
    """n < 32 (guarded factor instantiation), Panda-like 7-variable stacks, and a stack whose optimality rows
    duplicate its global equality rows (coman_ik.cpp:442; (18, [2]): with so many dependent rows that the null-space
    elimination of the Postural level hands over to the generic path); vs the oracle, and vs qpOASES when oracle/_ref is there"""
    plan, leaf = synth.make_generic_stack(64, n, rows, n_eq=n_eq, n_ineq=n_ineq, seed=n, duplicate_eq_in_level=dup)
    asm = oracle.assemble(plan, leaf)
    dq, xl, status, it, _ = _run(plan, leaf)
    assert (status == 0).all()
    wit = {}
    if dup is None:
        ref = wit["eiQuadProg"] = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
        okr = ref["status"] == 1   # the restated eiQuadProg routine gives up on a few degenerate instances
        assert np.abs(dq[okr] - ref["dq"][okr]).max(initial=0.0) < 1e-9
    if oracle.ref_available():
        rq = wit["qpOASES"] = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1)
        ok = rq["status"] == 1
        assert np.abs(dq[ok] - rq["dq"][ok]).max(initial=0.0) < 1e-6
    judge_remainder(asm, dq, wit, label=f"generic n={n} rows={rows}")      # (an instance a witness gave up on is still judged)


@pytest.mark.parametrize("kw", [dict(m=3), dict(m=4, weight=2.5), dict(m=3, postural_weight=1e-3),
                                dict(m=3, dependent=True), dict(m=4, zero_row=True, postural_weight=0.05),
                                dict(m=1, second_level_rows=0), dict(m=2, eps_factor=2e2, postural_weight=1e-3)])
def test_lowrank_levels_gpu(kw, oracle, gpu_device):
    """levels with <= 4 stored rows (closed-form J and minimiser, lowrank_prepare32) against the oracle and qpOASES:
    weights, a Postural block in the same level, a dependent row, a zero row, the default eps factor"""
    n = 12 if kw.get("m", 3) != 4 else 32
    plan, leaf = synth.make_lowrank_stack(256, n, seed=11, **kw)
    asm = oracle.assemble(plan, leaf)
    dq, xl, status, it, _ = _run(plan, leaf)
    assert (status == 0).all()
    # dependent equality rows, or more equality rows than variables at the second level (a Postural block in the
    # first level leaves m + n optimality rows): outside what the restated eiQuadProg routine supports
    degenerate = (kw.get("dependent") or kw.get("zero_row")
                  or (kw.get("postural_weight") is not None and kw.get("second_level_rows", 5) != 0))
    wit = {}
    if not degenerate:   # (see test_emulated_kernels.test_lowrank_levels: dependent equalities are pinned by qpOASES only)
        ref = wit["eiQuadProg"] = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
        okr = ref["status"] == 1
        assert np.abs(dq[okr] - ref["dq"][okr]).max(initial=0.0) < (1e-9 if kw.get("eps_factor", 1e6) == 1e6 else 1e-7)
    if oracle.ref_available():
        rq = wit["qpOASES"] = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1)
        ok = rq["status"] == 1
        assert np.abs(dq[ok] - rq["dq"][ok]).max(initial=0.0) < 1e-6
    judge_remainder(asm, dq, wit, label=f"lowrank {kw}")


@pytest.mark.parametrize("n,rows", [(32, [45]), (32, [33, 7]), (20, [37])])
def test_more_rows_than_variables_gpu(n, rows, oracle, gpu_device):
    """over-determined levels: more stored rows than one round of H-build loads covers (32), row counts that are
    not multiples of four"""
    plan, leaf = synth.make_generic_stack(128, n, rows, seed=5, postural_last=False)
    asm = oracle.assemble(plan, leaf)
    dq, xl, status, it, _ = _run(plan, leaf)
    assert (status == 0).all()
    wit = {}
    if rows[0] <= n or len(rows) == 1:
        ref = wit["eiQuadProg"] = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
        okr = ref["status"] == 1
        assert np.abs(dq[okr] - ref["dq"][okr]).max(initial=0.0) < 1e-9
    elif oracle.ref_available():   # (see the emulator test: more equality rows than variables is qpOASES-only)
        rq = wit["qpOASES"] = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1)
        ok = rq["status"] == 1
        assert np.abs(dq[ok] - rq["dq"][ok]).max(initial=0.0) < 1e-6
    judge_remainder(asm, dq, wit, label=f"rows > n: n={n} rows={rows}")


def test_subtasks_gpu(oracle, gpu_device):
    """SubTask blocks through the update kernel (bit-exact against the oracle's assembly) and the cascade"""
    plan, leaf = synth.make_subtask_stack(192, seed=2)
    asm = oracle.assemble(plan, leaf)
    st = BatchedStack(plan, 192, device=0)
    st.update(st.load_leaf(leaf)); st.solve(192)
    torch.cuda.synchronize()
    for k in range(plan.L):
        np.testing.assert_allclose(st.b[k].cpu().numpy(), asm["b"][k], rtol=0, atol=1e-15)
        np.testing.assert_array_equal(st.w[k].cpu().numpy(), asm["w"][k])
    dq = st.dq[:192].cpu().numpy()
    assert (st.status[:192].cpu().numpy() == 0).all()
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
    okr = ref["status"] == 1
    assert np.abs(dq[okr] - ref["dq"][okr]).max(initial=0.0) < 1e-9
    wit = {"eiQuadProg": ref}
    if oracle.ref_available():
        rq = wit["qpOASES"] = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1)
        rx = wit["qpOASES exact"] = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1, termination_tolerance=10 * 2.221e-16)
        e = np.minimum(np.where(rq["status"] == 1, np.abs(dq - rq["dq"]).max(axis=1), np.inf),
                       np.where(rx["status"] == 1, np.abs(dq - rx["dq"]).max(axis=1), np.inf))
        assert e[np.isfinite(e)].max(initial=0.0) < 1e-6
    judge_remainder(asm, dq, wit)


@pytest.mark.gpu
@pytest.mark.parametrize("name,kind,rows,weight", [("C3", 0, None, 1e-2), ("C3", 3, None, 0.5), ("C2", 0, 20, 1e-3),
                                                   ("generic40", 3, None, 1e-2), ("lowrank", 0, None, 0.2),
                                                   ("lowrank_plain", 3, 20, 0.05), ("id", 6, 32, 1e-2)])
def test_user_regularisation_task_gpu(name, kind, rows, weight, oracle, gpu_device):
    """AutoStack::setRegularisationTask (iHQP.cpp:265-266, 274-278): b_r through the update kernel (bit-exact against
    the oracle's assembly), H += Hr, g += gr inside the cascade kernel, against qpOASES (and the restated eiQuadProg
    where it applies)"""
    from test_emulated_kernels import _reg_stack
    B = 256
    plan, leaf = _reg_stack(name, B, seed=5)
    synth.add_regularisation(plan, leaf, kind=kind, rows=rows, weight=weight, seed=3)
    asm = oracle.assemble(plan, leaf)
    st = BatchedStack(plan, B, device=0)
    st.update(st.load_leaf(leaf)); st.solve(B)
    torch.cuda.synchronize()
    np.testing.assert_allclose(st.b_reg.cpu().numpy(), asm["reg"]["b"], rtol=0, atol=1e-15)
    for k in range(plan.L):
        np.testing.assert_allclose(st.b[k].cpu().numpy(), asm["b"][k], rtol=0, atol=1e-14)
    dq = st.dq[:B].cpu().numpy()
    assert (st.status[:B].cpu().numpy() == 0).all()
    scale = max(1.0, np.abs(dq).max())
    wit = {}
    if name not in ("id", "lowrank"):
        ref = wit["eiQuadProg"] = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
        okr = ref["status"] == 1
        assert np.abs(dq[okr] - ref["dq"][okr]).max(initial=0.0) < 1e-9 * scale
    if oracle.ref_available():
        rq = wit["qpOASES"] = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1)
        rx = wit["qpOASES exact"] = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1, termination_tolerance=10 * 2.221e-16)
        e = np.minimum(np.where(rq["status"] == 1, np.abs(dq - rq["dq"]).max(axis=1), np.inf),
                       np.where(rx["status"] == 1, np.abs(dq - rx["dq"]).max(axis=1), np.inf))
        assert e[np.isfinite(e)].max(initial=0.0) < 1e-6 * scale
    else:
        assert name not in ("id", "lowrank"), "this case needs oracle/_ref (qpOASES)"
    judge_remainder(asm, dq, wit, tol=1e-6 * scale, label=f"regularisation {name}")


@pytest.mark.gpu
@pytest.mark.parametrize("name,kind,rows,weight", [("C3", 1, None, 1e-2), ("C3", 0, 9, 0.3), ("C2", 2, None, 5e-2),
                                                   ("lowrank_plain", 0, 5, 0.1), ("generic40", 1, None, 1e-2)])
def test_regularisation_task_with_a_stored_jacobian_gpu(name, kind, rows, weight, oracle, gpu_device):
    """a regularisation task with a DENSE Jacobian (iHQP.cpp:265-278 takes any task): b_r through the update kernel, A_r
    written in place, H += w A_r'A_r and g -= w A_r'b_r at every level of the cascade; against the eiQuadProg
    restatement (1e-9) and qpOASES at OpenSoT's options (absolute 1e-6, census printed)"""
    from helpers import parity_census
    from test_emulated_kernels import _reg_stack
    B = 192
    plan, leaf = _reg_stack(name, B, seed=6)
    synth.add_regularisation(plan, leaf, kind=kind, rows=rows, weight=weight, seed=4, dense=True)
    asm = oracle.assemble(plan, leaf)
    st = BatchedStack(plan, B, device=0)
    st.update(st.load_leaf(leaf)); st.solve(B)
    torch.cuda.synchronize()
    np.testing.assert_allclose(st.b_reg.cpu().numpy(), asm["reg"]["b"], rtol=0, atol=1e-15)
    dq = st.dq[:B].cpu().numpy()
    assert (st.status[:B].cpu().numpy() == 0).all()
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=4)
    okr = ref["status"] == 1
    assert okr.all() and np.abs(dq - ref["dq"]).max() < 1e-9 * max(1.0, np.abs(dq).max())
    # the fused cycle launch takes the same route
    st2 = BatchedStack(plan, B, device=0)
    st2.cycle(st2.load_leaf(leaf)); torch.cuda.synchronize()
    assert torch.equal(st2.dq[:B], st.dq[:B])
    if oracle.ref_available():
        rq = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=4)
        within, rule, fails = parity_census(asm, dq, [("qpOASES", rq), ("eiQuadProg", ref)], tol=1e-6, label=f"dense regularisation {name}")
        assert not fails


@pytest.mark.gpu
@pytest.mark.parametrize("n,rows,local_level,n_local", [(7, [3, 3], 0, 2), (20, [5, 6], 1, 4), (31, [10, 12], 2, 3), (40, [10, 12], 0, 5)])
def test_task_local_constraint_rows_gpu(n, rows, local_level, n_local, oracle, gpu_device):
    """`task << constraint` rows (Task::getConstraints(), iHQP.cpp:190, 282-287) through update + cascade on the GPU"""
    B = 200
    plan, leaf = synth.make_generic_stack(B, n, rows, n_eq=1, n_ineq=2, seed=9, n_local=n_local, local_level=local_level)
    asm = oracle.assemble(plan, leaf)
    st = BatchedStack(plan, B, device=0)
    st.update(st.load_leaf(leaf)); st.solve(B)
    torch.cuda.synchronize()
    dq = st.dq[:B].cpu().numpy(); xl = st.x_levels[:B].cpu().numpy()
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
    okr = ref["status"] == 1
    status = st.status[:B].cpu().numpy()
    solvable = okr.copy()
    if oracle.ref_available():
        rq = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1)
        rx = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1, termination_tolerance=10 * 2.221e-16)
        solvable |= (rq["status"] == 1) | (rx["status"] == 1)
    # tight local rows on top of the optimality equalities can be infeasible (1 instance in 200 at the third case):
    # the verdict must agree with the witnesses (an instance any of them solves is feasible; the restated eiQuadProg
    # routine alone gives up on a few with dependent equality rows), and a failed instance returns dq = 0
    # (coman_ik.cpp:189-190)
    assert ((status == 0) == solvable).all() and (status[~solvable] == 1).all() and (dq[~solvable] == 0.0).all()
    assert np.abs(dq[okr] - ref["dq"][okr]).max(initial=0.0) < 1e-9
    assert np.abs(xl[okr] - ref["x_levels"][okr]).max() < 1e-9
    if oracle.ref_available():
        e = np.minimum(np.where(rq["status"] == 1, np.abs(dq - rq["dq"]).max(axis=1), np.inf),
                       np.where(rx["status"] == 1, np.abs(dq - rx["dq"]).max(axis=1), np.inf))[solvable]
        assert e.max() < 1e-6


@pytest.mark.gpu
def test_diagonal_weight_matrices_gpu(oracle, gpu_device):
    """Task::setWeight(W) with a diagonal W (per-row weights; tasks::Aggregated::generateWeight, Aggregated.cpp:265-279):
    the caller fills w_k once and the update is told to leave it alone (out.w[k] = NULL)"""
    B = 256
    plan, leaf = synth.make_velocity_stack("C3", B, seed=21)
    asm = oracle.assemble(plan, leaf)
    rng = np.random.default_rng(4)
    for k in range(plan.L):
        asm["w"][k] = np.ascontiguousarray(np.broadcast_to(rng.uniform(0.2, 3.0, size=(1, plan.m(k))), (B, plan.m(k))))
    st = BatchedStack(plan, B, device=0)
    for k in range(plan.L):
        st.w[k][:B].copy_(torch.as_tensor(asm["w"][k]))
    st.update(st.load_leaf(leaf), write_weights=False); st.solve(B)
    torch.cuda.synchronize()
    for k in range(plan.L):
        np.testing.assert_array_equal(st.w[k].cpu().numpy(), asm["w"][k])
        np.testing.assert_allclose(st.b[k].cpu().numpy(), asm["b"][k], rtol=0, atol=1e-15)
    dq = st.dq[:B].cpu().numpy()
    assert (st.status[:B].cpu().numpy() == 0).all()
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
    okr = ref["status"] == 1
    assert np.abs(dq[okr] - ref["dq"][okr]).max(initial=0.0) < 1e-9
    wit = {"eiQuadProg": ref}
    if oracle.ref_available():
        rq = wit["qpOASES"] = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1)
        rx = wit["qpOASES exact"] = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1, termination_tolerance=10 * 2.221e-16)
        e = np.minimum(np.where(rq["status"] == 1, np.abs(dq - rq["dq"]).max(axis=1), np.inf),
                       np.where(rx["status"] == 1, np.abs(dq - rx["dq"]).max(axis=1), np.inf))
        assert e[np.isfinite(e)].max(initial=0.0) < 1e-6
    judge_remainder(asm, dq, wit)


def _collision_last_direction_instance():
    """an instance met at cycle 28 of the closed-loop self-collision test (tests/test_kinematics.py), kept as data:
    feet / wrist-position / Postural levels, 16 capsule-pair rows, velocity box"""
    import os
    from opensot_amd import abi
    from opensot_amd.plan import StackPlan, Task, Bound, Rows, subtask, eps_abs_from_factor
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "collision_last_direction_instance.npz"))
    n, P = 32, 16
    wrist = lambda nm: subtask(Task(abi.TASK_CARTESIAN, 6, lam=0.1, name=nm), [0, 1, 2])
    levels = [[Task(abi.TASK_CARTESIAN, 6, lam=0.1), Task(abi.TASK_CARTESIAN, 6, lam=0.1)], [wrist("l"), wrist("r")],
              [Task(abi.TASK_POSTURAL, n, lam=0.01)]]
    plan = StackPlan(n=n, levels=levels, bounds=[Bound(abi.BOUND_VELOCITY_LIMITS, dT=0.01)],
                     rowblocks=[Rows(abi.ROWS_COLLISION, P, d_threshold=0.02, bound_scaling=0.2)], eps_abs=eps_abs_from_factor(1e6))
    asm = {"n": n, "B": 1, "L": 3, "eps_abs": plan.eps_abs, "m": [12, 6, 32], "ma": [12, 6, 0],
           "A": [z["A0"][None], z["A1"][None], None], "b": [z["b0"][None], z["b1"][None], z["b2"][None]],
           "w": [z["w0"][None], z["w1"][None], z["w2"][None]], "c": [None] * 3, "nc": P, "C": z["C"][None],
           "lo": z["lo"][None], "up": z["up"][None], "l": z["l"][None], "u": z["u"][None]}
    return plan, asm


@pytest.mark.gpu
def test_noise_is_not_a_direction_gpu(oracle, gpu_device):
    """found by the closed-loop self-collision test on hardware: at the Postural level, 29 of 32 directions taken, a
    bound violated by 4e-11 whose normal had |d2|^2 = 9e-23 |d|^2 left outside the working set -- round-off of a J that
    had been through 29 updates, not a direction.  Taken as one, x jumped by 26 and the level ended INFEASIBLE while
    qpOASES and the eiQuadProg restatement solve it (kDepFloor2 in osot_qp_core.h; tests/stress_closed_loop.py is the sweep that chose its value)."""
    plan, asm = _collision_last_direction_instance()
    st = BatchedStack(plan, 1, device=0)
    st.load_assembled(asm); st.solve(1)
    torch.cuda.synchronize()
    assert int(st.status[0]) == 0
    dq = st.dq[:1].cpu().numpy()
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
    assert ref["status"][0] == 1 and np.abs(dq - ref["dq"]).max() < 1e-9
    if oracle.ref_available():
        # (qpOASES at OpenSoT's options stops 2e-2 from the optimum here, its box violated by 1e-7; run to the exact
        # optimum it agrees)
        rx = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1, termination_tolerance=10 * 2.221e-16)
        assert rx["status"][0] == 1 and np.abs(dq - rx["dq"]).max() < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("n,rows", [(7, [3, 3]), (31, [10, 12]), (40, [10, 12])])
def test_task_local_bounds_as_unit_rows_gpu(n, rows, oracle, gpu_device):
    """`task << bound` as a level-tagged block of unit rows (OSOT_ROWS_UNIT_GENERIC): update + cascade on the GPU"""
    B, hw = 160, 0.01
    plan, leaf = synth.make_generic_stack(B, n, rows, n_eq=1, n_ineq=2, seed=4, box=0.5, unit_box=(0, hw))
    asm = oracle.assemble(plan, leaf)
    st = BatchedStack(plan, B, device=0)
    st.update(st.load_leaf(leaf)); st.solve(B)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(st.lo.cpu().numpy(), asm["lo"])
    np.testing.assert_array_equal(st.up.cpu().numpy(), asm["up"])
    dq = st.dq[:B].cpu().numpy(); xl = st.x_levels[:B].cpu().numpy()
    status = st.status[:B].cpu().numpy()
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
    okr = ref["status"] == 1
    solvable = okr.copy()
    if oracle.ref_available():
        rq = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1)
        rx = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1, termination_tolerance=10 * 2.221e-16)
        solvable |= (rq["status"] == 1) | (rx["status"] == 1)
    assert ((status == 0) == solvable).all()          # (EVERY instance: solved exactly where a witness solves it)
    assert solvable.sum() >= 0.95 * B                  # (sanity of the test data, not an acceptance mask)
    assert np.abs(dq[okr] - ref["dq"][okr]).max() < 1e-9 and np.abs(xl[okr] - ref["x_levels"][okr]).max() < 1e-9
    if oracle.ref_available():
        e = np.minimum(np.where(rq["status"] == 1, np.abs(dq - rq["dq"]).max(axis=1), np.inf),
                       np.where(rx["status"] == 1, np.abs(dq - rx["dq"]).max(axis=1), np.inf))[solvable]
        assert e.max() < 1e-6
    assert np.abs(xl[solvable][:, 0]).max() <= hw + 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("n,rows", [(12, [9]), (16, [5, 6]), (32, [10, 17]), (32, [3, 24])])
def test_task_local_equality_on_a_postural_last_level_gpu(n, rows, oracle, gpu_device):
    """a task-local EQUALITY at a Postural last level: the null-space shortcut (which assumes x_prev satisfies every
    equality) must not be taken -- see the emulator test of the same name"""
    B = 96
    plan, leaf = synth.make_generic_stack(B, n, rows, n_eq=0, n_ineq=2, seed=21, n_local=1, local_level=len(rows), local_equality=True)
    asm = oracle.assemble(plan, leaf)
    dq, xl, status, it, _ = _run(plan, leaf)
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
    okr = ref["status"] == 1   # (a random equality at the last level is infeasible for a few instances: witness and product agree)
    assert (status[okr] == 0).all() and okr.sum() >= B // 2       # (sanity of the test data: most instances are feasible)
    from helpers import answer_is_acceptable
    for i in np.nonzero((status == 0) & ~okr)[0]:      # the witness gave up, the product answered: the answer carries its own certificate
        ok_i, why = answer_is_acceptable(asm, int(i), dq[i], [("eiQuadProg", ref["dq"][i], False)])
        assert ok_i, (int(i), why)
    Cl, lo, up = leaf["rows"][-1]
    assert np.abs(np.einsum("bri,bi->br", Cl, dq) - lo)[status == 0].max() < 1e-9
    assert np.abs(dq[okr] - ref["dq"][okr]).max() < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["tasks", "ttc", "ttc_exchange"])
def test_default_eps_stuck_instances_gpu(mode, oracle, gpu_device):
    """tests/golden/default_eps_stuck_instances.npz on hardware (see the emulator test of the same name)"""
    from helpers import answer_is_acceptable, default_eps_stuck_instances
    plan, asm = default_eps_stuck_instances(mode)
    B = asm["B"]
    st = BatchedStack(plan, B, device=0)
    st.load_assembled(asm); st.solve(B)
    torch.cuda.synchronize()
    assert (st.status[:B].cpu().numpy() == 0).all()
    dq = st.dq[:B].cpu().numpy()
    re_ = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
    wit = [("eiQuadProg", re_)]
    if oracle.ref_available():
        wit += [("qpOASES exact", oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1, termination_tolerance=10 * 2.221e-16)),
                ("qpOASES", oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1))]
    for i in range(B):
        ok, why = answer_is_acceptable(asm, i, dq[i], [(nm, r["dq"][i], r["status"][i] == 1) for nm, r in wit])
        assert ok, (i, why)


@pytest.mark.gpu
def test_accepted_slack_instance_gpu(gpu_device):
    """tests/golden/default_eps_accepted_slack_instance.npz: the instance of round 4's randomised sweep on which the
    kernel accepted a 4.2e-7 violation of a global inequality row as round-off of the levels above (default eps) and ended 1.6e-6
    from qpOASES: the literal rule -- within 1e-6 of a witness, or feasible to 1e-7 and lexicographically not worse -- must hold,
    and whatever the kernel accepts it reports (accepted_slack) and keeps below the rule's 1e-7"""
    from helpers import accepted_slack_instance, answer_is_acceptable
    plan, asm, wit = accepted_slack_instance()
    st = BatchedStack(plan, 1, device=0)
    st.load_assembled(asm); st.solve(1)
    torch.cuda.synchronize()
    assert int(st.status[0].item()) == 0
    dq = st.dq[:1].cpu().numpy()
    ok, why = answer_is_acceptable(asm, 0, dq[0], wit)
    assert ok, why
    assert float(st.accepted_slack[0].item()) <= 1.0e-7


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["C3", "C4", "C5"])
def test_hot_start_gpu(cfg, oracle, gpu_device):
    """osot_solver_set_hotstart (reference: QPOasesBackEnd.cpp:258-285 hotstart -> SQProblem.cpp:149-193): every level's
    working set of the instance's previous solve is re-added before the first scan.  Same dq as the cold solve to 1e-9
    (both against the oracle too) over drifting cycles; on an exact repeat of a cycle the hot list is the final active
    set: no more iterations than cold for any instance, fewer in total."""
    B = 256 if cfg != "C5" else 64
    plan, leaf = synth.make_id_stack(B, seed=5200) if cfg == "C5" else synth.make_velocity_stack(cfg, B, seed=4200)
    rng = np.random.default_rng(8)
    leaves = [leaf, synth.perturb(leaf, rng, 0.01), synth.perturb(leaf, rng, 0.002)]
    cold, hot = BatchedStack(plan, B, device=0), BatchedStack(plan, B, device=0)
    hot.set_hotstart(True)
    for i, lf in enumerate(leaves + [leaves[-1]]):
        out = []
        for st in (cold, hot):
            st.update(st.load_leaf(lf)); st.solve(B)
            torch.cuda.synchronize()
            out.append((st.dq[:B].cpu().numpy(), st.status[:B].cpu().numpy(), st.iterations[:B].cpu().numpy()))
        (dq0, st0, it0), (dq1, st1, it1) = out
        assert (st0 == st1).all()
        ok = st0 == 0
        assert ok.sum() >= 0.9 * B                     # (sanity of the test data; cold and hot agree on EVERY status above)
        scale = max(1.0, np.abs(dq0[ok]).max()) if cfg == "C5" else 1.0     # (torque-mode variables are accelerations / forces)
        assert np.abs(dq0[ok] - dq1[ok]).max() < 1e-9 * scale
        if i == 0:
            assert (it0 == it1).all()                    # nothing recorded yet: a cold start
            ref = oracle.ihqp_solve_batch(oracle.assemble(plan, lf), oracle.BE_EIQP_EQ, nthreads=4)
            both = ok & (ref["status"] == 1)
            assert np.abs(dq1[both] - ref["dq"][both]).max() < 1e-8 * scale
        if i == len(leaves):                             # exact repeat: the hot list is the final active set
            assert (it1[ok] <= it0[ok]).all() and it1[ok].sum() < it0[ok].sum()
    hot.set_hotstart(False)                              # switching it off is a cold start again, bit for bit
    hot.solve(B); cold.solve(B)
    torch.cuda.synchronize()
    assert torch.equal(hot.dq[:B], cold.dq[:B])


@pytest.mark.gpu
def test_pipelined_lanes_match_single_launch_gpu(gpu_device):
    """opensot_amd.parallel.PipelinedCycle (what bench.py times): the batch as S sub-batches on S streams, no join between
    steps -- same dq, bit for bit, as one launch over the whole batch, for every cycle of the rotation"""
    from opensot_amd.parallel import PipelinedCycle, ShardedCycle, lane_ranges
    B, S, K = 600, 3, 3
    plan, leaf = synth.make_velocity_stack("C3", B, seed=4300)
    rng = np.random.default_rng(9)
    leaves = [leaf]
    for _ in range(K - 1):
        leaves.append(synth.perturb(leaves[-1], rng, 0.01))

    def cut(lf, a, b):
        c = lambda x: None if x is None else x[a:b]
        return {"B": b - a, "A": [c(x) for x in lf["A"]], "task": [[tuple(c(x) for x in t) for t in lev] for lev in lf["task"]],
                "bound": [tuple(c(x) for x in t) for t in lf["bound"]], "rows": [tuple(c(x) for x in t) for t in lf["rows"]]}

    def build(a, b):
        st = BatchedStack(plan, b - a, device=0, want_levels=False)
        devs, As = [], []
        for lf in leaves:
            st.A = [None if t is None else torch.empty_like(t) for t in st.A]
            devs.append(st.load_leaf(cut(lf, a, b))); As.append(st.A)
        return ShardedCycle(st, devs, As, b - a, None)

    one = build(0, B)
    spans = lane_ranges(B, S)
    lanes = [build(a, b) for a, b in spans]
    pipe = PipelinedCycle(lanes, [torch.cuda.Stream() for _ in range(S)])
    torch.cuda.synchronize()
    for step in range(2 * K + 1):
        one.step(); pipe.step()
    torch.cuda.synchronize()
    got = torch.cat([ln.stack.dq[:b - a] for ln, (a, b) in zip(lanes, spans)])
    assert (one.stack.status[:B] == 0).all()
    assert torch.equal(got, one.stack.dq[:B])
    # the same steps as HIP graphs (PipelinedCycle.capture / replay): 2 K steps of every lane per graph; a replay ends on the
    # cycle the capture ended on, and gives that cycle's dq bit for bit -- replayed twice, and against the single launch
    pipe.capture(2 * K)
    k_last = (lanes[0].i - 1) % K
    pipe.replay(); pipe.replay()
    torch.cuda.synchronize()
    got_g = torch.cat([ln.stack.dq[:b - a] for ln, (a, b) in zip(lanes, spans)])
    while (one.i - 1) % K != k_last:
        one.step()
    torch.cuda.synchronize()
    assert (torch.cat([ln.stack.status[:b - a] for ln, (a, b) in zip(lanes, spans)]) == 0).all()
    assert torch.equal(got_g, one.stack.dq[:B])
    with pytest.raises(ValueError):
        pipe.capture(3)            # odd: the solver's alternating order buffers would end on the wrong side


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,B", [("C2", 300), ("C3", 700)])
def test_box_instantiation_is_bit_identical_gpu(gpu_device, cfg, B):
    """osot_solver_set_specialisation: BASELINE configs 2 and 3 have no constraint rows (their only inequalities are the joint /
    velocity limit box), so their launches run the BOX instantiation of the 32-column kernels (no row classification, no row
    scans, no bound-or-row branches: osot_qp_core.h, gi_inequalities).  It is the same arithmetic in the same order as the
    general instantiation: dq, status and iteration counts agree bit for bit, through osot_ihqp_solve and through osot_cycle,
    over a rotation of drifting cycles"""
    plan, leaf = synth.make_velocity_stack(cfg, B, seed=5150)
    rng = np.random.default_rng(3)
    leaves = [leaf, synth.perturb(leaf, rng, 0.01), synth.perturb(leaf, rng, 0.05)]
    spec = BatchedStack(plan, B, device=0, want_levels=True)
    gen = BatchedStack(plan, B, device=0, want_levels=True)
    gen.set_specialisation(False)
    for lf in leaves:
        for st in (spec, gen):
            st.update(st.load_leaf(lf)); st.solve(B)
        torch.cuda.synchronize()
        assert (spec.status[:B] == 0).all()
        assert torch.equal(spec.dq[:B], gen.dq[:B])
        assert torch.equal(spec.x_levels[:B], gen.x_levels[:B])
        assert torch.equal(spec.status[:B], gen.status[:B]) and torch.equal(spec.iterations[:B], gen.iterations[:B])
        for st in (spec, gen):
            st.cycle(st.load_leaf(lf))
        torch.cuda.synchronize()
        assert torch.equal(spec.dq[:B], gen.dq[:B]) and torch.equal(spec.iterations[:B], gen.iterations[:B])


@pytest.mark.gpu
@pytest.mark.parametrize("n,B", [(32, 300), (40, 260)])
def test_box_instantiation_for_equality_rows_is_bit_identical_gpu(gpu_device, n, B):
    """round 4: plans whose constraint rows are all TaskToConstraint blocks with a point band (the reference's COMAN stacks:
    the feet as `stack << l_sole`, coman_ik.cpp:425-449) run the BOX instantiation at every size -- the rows are equalities of
    every level, the bounds the only inequalities.  Bit-identical to the general instantiation (dq, every level's x, status,
    iteration counts) through osot_ihqp_solve and osot_cycle, over drifting cycles, 32-lane and 56-lane kernels; the rows hold
    in the answer."""
    from opensot_amd import abi
    from opensot_amd.plan import Rows, StackPlan
    if n == 32:
        plan, leaf = synth.make_velocity_stack("C3", B, seed=6001)
    else:
        plan, leaf = synth.make_generic_stack(B, n, [6, 12, 8], n_eq=0, n_ineq=0, seed=6002, box=0.4)
    rng = np.random.default_rng(8)
    J = rng.normal(0.0, 0.3, size=(B, 3, n))
    pa = rng.uniform(-0.2, 0.2, size=(B, 3))
    rb = Rows(abi.ROWS_TASK_COM, 3, lam=0.1, err_lb=0.0, err_ub=0.0, name="com_rows")
    plan = StackPlan(n=plan.n, levels=plan.levels, bounds=plan.bounds, rowblocks=list(plan.rowblocks) + [rb], eps_abs=plan.eps_abs)
    leaf = dict(leaf)
    leaf["rows"] = list(leaf.get("rows", [])) + [(pa, pa + rng.uniform(-0.01, 0.01, size=(B, 3)), None)]
    leaf["C"] = list(leaf.get("C", [])) + [J]
    leaves = [leaf, synth.perturb(leaf, rng, 0.01), synth.perturb(leaf, rng, 0.05)]
    spec = BatchedStack(plan, B, device=0, want_levels=True)
    gen = BatchedStack(plan, B, device=0, want_levels=True)
    gen.set_specialisation(False)
    for lf in leaves:
        for st in (spec, gen):
            st.update(st.load_leaf(lf)); st.solve(B)
        torch.cuda.synchronize()
        assert (spec.status[:B] == 0).all()
        assert torch.equal(spec.dq[:B], gen.dq[:B])
        assert torch.equal(spec.x_levels[:B], gen.x_levels[:B])
        assert torch.equal(spec.status[:B], gen.status[:B]) and torch.equal(spec.iterations[:B], gen.iterations[:B])
        res = torch.einsum("bij,bj->bi", spec.C[:B], spec.dq[:B]) - spec.lo[:B]
        assert torch.equal(spec.lo[:B], spec.up[:B]) and float(res.abs().max()) < 1e-9
        for st in (spec, gen):
            st.cycle(st.load_leaf(lf))
        torch.cuda.synchronize()
        assert torch.equal(spec.dq[:B], gen.dq[:B]) and torch.equal(spec.iterations[:B], gen.iterations[:B])

