"""Round-2 surface on the GPU through the C-ABI (the emulator twins are in tests/test_emulated_features_r2.py)."""
import ctypes as C

import numpy as np
import pytest
import torch

from opensot_amd import abi, synth
from opensot_amd.solver import BatchedStack, stored_rows

pytestmark = pytest.mark.gpu


def test_feature_stack_gpu(oracle, gpu_device):
    """body-frame Cartesian b, per-row TaskToConstraint bands, collision rows chosen among 24 candidates, six row blocks
    and a full weight matrix: update bit-equal to the oracle's assembly, W A / W b equal to numpy, cascade vs witnesses"""
    B = 192
    plan, leaf = synth.make_feature_stack(B, seed=5)
    asm = oracle.assemble(plan, leaf)
    st = BatchedStack(plan, B, device=0)
    st.update(st.load_leaf(leaf)); st.solve(B)
    torch.cuda.synchronize()
    for k in range(plan.L):
        np.testing.assert_allclose(st.b[k].cpu().numpy(), asm["b"][k], rtol=0, atol=1e-15)
        np.testing.assert_array_equal(st.w[k].cpu().numpy(), asm["w"][k])
    np.testing.assert_array_equal(st.l.cpu().numpy(), asm["l"]); np.testing.assert_array_equal(st.u.cpu().numpy(), asm["u"])
    np.testing.assert_array_equal(st.C.cpu().numpy(), stored_rows(plan, asm["C"]))
    np.testing.assert_allclose(st.lo.cpu().numpy(), asm["lo"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(st.up.cpu().numpy(), asm["up"], rtol=0, atol=1e-15)
    W = asm["Wdense"][1]
    np.testing.assert_allclose(st.WA[1].cpu().numpy(), W @ asm["A"][1], rtol=0, atol=1e-14)
    np.testing.assert_allclose(st.Wb[1].cpu().numpy(), np.einsum("brq,bq->br", W, asm["b"][1]), rtol=0, atol=1e-15)
    dq = st.dq[:B].cpu().numpy(); status = st.status[:B].cpu().numpy()
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=0)
    okr = ref["status"] == 1
    assert okr.mean() > 0.95 and (status[okr] == 0).all() and np.abs(dq[okr] - ref["dq"][okr]).max() < 1e-9
    if oracle.ref_available():
        rq = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=0)
        rx = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=0, termination_tolerance=10 * 2.221e-16)
        e = np.minimum(np.where(rq["status"] == 1, np.abs(dq - rq["dq"]).max(axis=1), np.inf),
                       np.where(rx["status"] == 1, np.abs(dq - rx["dq"]).max(axis=1), np.inf))
        assert np.isfinite(e).mean() > 0.95 and e[np.isfinite(e)].max() < 1e-6


@pytest.mark.parametrize("off", [[(1, 0)], [(1, 1), (1, 2)], [(2, 0)], [(0, 0)]])
def test_task_set_active_gpu(off, oracle, gpu_device):
    """Task::setActive(false) through osot_solver_set_task_active; witness: qpOASES on the problem with the task's rows
    zeroed as the reference does (Task.h:383-387); re-activating restores the all-active answer bit for bit"""
    B = 128
    plan, leaf = synth.make_velocity_stack("C3", B, seed=12)
    asm_ref = oracle.assemble(plan, leaf, task_active={kj: False for kj in off})
    st = BatchedStack(plan, B, device=0)
    dev = st.load_leaf(leaf)
    st.update(dev); st.solve(B); torch.cuda.synchronize()
    dq_all = st.dq[:B].cpu().numpy().copy()
    for k, j in off:
        st.set_task_active(k, j, False)
    st.update(dev); st.solve(B); torch.cuda.synchronize()
    dq = st.dq[:B].cpu().numpy().copy()
    assert (st.status[:B].cpu().numpy() == 0).all() and np.abs(dq - dq_all).max() > 1e-6
    if oracle.ref_available():
        rq = oracle.ihqp_solve_batch(asm_ref, oracle.BE_QPOASES_REF, nthreads=0, termination_tolerance=10 * 2.221e-16)
        rd = oracle.ihqp_solve_batch(asm_ref, oracle.BE_QPOASES_REF, nthreads=0)
        e = np.minimum(np.where(rq["status"] == 1, np.abs(dq - rq["dq"]).max(axis=1), np.inf),
                       np.where(rd["status"] == 1, np.abs(dq - rd["dq"]).max(axis=1), np.inf))
        assert np.isfinite(e).mean() > 0.95 and e[np.isfinite(e)].max() < 1e-6
    for k, j in off:
        st.set_task_active(k, j, True)
    st.update(dev); st.solve(B); torch.cuda.synchronize()
    assert np.array_equal(st.dq[:B].cpu().numpy(), dq_all)


def test_inverse_dynamics_producers_and_computed_torque_gpu(oracle, gpu_device):
    """osot_id_rows writes [B_u, -J_f'], [B, -Jc'] and the [J 0] task rows straight into C / A_0; osot_computed_torque gives
    tau = B qddot + h - sum Jc'F of the device's own solution with the floating-base rows at zero
    (InverseDynamics.cpp:57-96)"""
    from opensot_amd.dynamics import IdModel
    B = 256
    plan, leaf = synth.make_id_stack(B, seed=9)
    n, nv = plan.n, leaf["model"]["nv"]
    st = BatchedStack(plan, B, device=0)
    bare = dict(leaf); bare["A"] = [np.zeros_like(leaf["A"][0]), None]; bare["C"] = [None] * len(leaf["C"])   # nothing pre-stacked
    dev = st.load_leaf(bare)
    md = IdModel(leaf["model"]["B"], leaf["model"]["h"], leaf["model"]["Jc"], device=0)
    J = [torch.as_tensor(np.ascontiguousarray(leaf["A"][0][:, o:o + r, :nv])).to(st.device) for o, r in ((0, 3), (3, 6), (9, 6))]
    md.write_rows(st, dyn_block=0, tau_block=2, tasks=[(0, 0, J[0]), (0, 3, J[1]), (0, 9, J[2])])
    torch.cuda.synchronize()
    np.testing.assert_array_equal(st.A[0].cpu().numpy(), leaf["A"][0])
    o_dyn, o_tau = plan.rows_stored_offset(0), plan.rows_stored_offset(2)
    np.testing.assert_array_equal(st.C[:, o_dyn:o_dyn + 6].cpu().numpy(), leaf["C"][0])
    np.testing.assert_array_equal(st.C[:, o_tau:o_tau + nv].cpu().numpy(), leaf["C"][2])
    st.update(dev); st.solve(B)
    tau, ok = md.computed_torque(st.dq[:B])
    torch.cuda.synchronize()
    assert (st.status[:B].cpu().numpy() == 0).all()
    x = st.dq[:B].cpu().numpy()
    np.testing.assert_allclose(tau.cpu().numpy(), synth.computed_torque(leaf, x), rtol=0, atol=1e-10)
    assert (ok.cpu().numpy() == 1).all() and np.abs(tau[:, :6].cpu().numpy()).max() < 1e-8
    asm = oracle.assemble(plan, leaf)
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=0)
    assert (ref["status"] == 1).all() and np.abs(x - ref["dq"]).max() < 1e-8
    if oracle.ref_available():
        from helpers import parity_census      # absolute 1e-6 against qpOASES, every instance counted (see helpers)
        rq = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=0)
        within, rule, fails = parity_census(asm, x, [("qpOASES", rq), ("eiQuadProg", ref)], tol=1e-6, label="C5 device producers")
        assert not fails and within >= 0.95 * asm["B"]


@pytest.mark.parametrize("cfg,B", [("C3", 300), ("C4", 128), ("C5", 64), ("feature", 96)])
def test_cycle_is_update_then_solve_gpu(cfg, B, gpu_device):
    """osot_cycle (update + cascade of an instance by the same wavefront, one launch) gives bit-identical assembled arrays
    and solutions to osot_stack_update followed by osot_ihqp_solve"""
    if cfg == "feature":
        plan, leaf = synth.make_feature_stack(B, seed=8)
    elif cfg == "C5":
        plan, leaf = synth.make_id_stack(B, seed=8)
    else:
        plan, leaf = synth.make_velocity_stack(cfg, B, seed=8)
    a = BatchedStack(plan, B, device=0); b = BatchedStack(plan, B, device=0)
    da, db = a.load_leaf(leaf), b.load_leaf(leaf)
    a.update(da); a.solve(B)
    b.cycle(db)
    torch.cuda.synchronize()
    assert (a.status[:B] == 0).all()
    for x, y in [(a.dq, b.dq), (a.x_levels, b.x_levels), (a.status, b.status), (a.iterations, b.iterations), (a.lo, b.lo), (a.up, b.up),
                 (a.l, b.l), (a.u, b.u), (a.C, b.C)] + list(zip(a.b, b.b)) + list(zip(a.w, b.w)):
        if x is not None:
            assert torch.equal(x, y)
    # and a second cycle on the same objects (dispatch order from the first one's iteration counts)
    b.cycle(db); torch.cuda.synchronize()
    assert torch.equal(a.dq, b.dq)
