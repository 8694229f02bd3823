"""Round-2 surface on the GPU through the C-ABI (the emulator twins are in tests/test_emulated_features_r2.py)."""
import ctypes as C

import numpy as np
import pytest
import torch

from opensot_amd import abi, synth
from opensot_amd.solver import BatchedStack, stored_rows

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [32, 40, 60])
def test_feature_stack_gpu(n, oracle, gpu_device):
    """body-frame Cartesian b, per-row TaskToConstraint bands, collision rows chosen among 24 candidates, six row blocks
    and a full weight matrix: update bit-equal to the oracle's assembly, W A / W b equal to numpy, cascade vs witnesses.
    n = 40 / 60: the dense-weight instantiation of the 64-lane cascade on its short and full LDS layouts"""
    B = 192
    plan, leaf = synth.make_feature_stack(B, seed=5, n=n)
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
    assert (status == 0).all() and np.abs(dq[okr] - ref["dq"][okr]).max(initial=0.0) < 1e-9
    wit = {"eiQuadProg": ref}
    if oracle.ref_available():
        rq = wit["qpOASES"] = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=0)
        rx = wit["qpOASES exact"] = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=0, termination_tolerance=10 * 2.221e-16)
        e = np.minimum(np.where(rq["status"] == 1, np.abs(dq - rq["dq"]).max(axis=1), np.inf),
                       np.where(rx["status"] == 1, np.abs(dq - rx["dq"]).max(axis=1), np.inf))
        assert e[np.isfinite(e)].max(initial=0.0) < 1e-6
    from helpers import judge_remainder
    judge_remainder(asm, dq, wit, label="feature stack")     # (an instance a witness gave up on is still judged)


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
        assert e[np.isfinite(e)].max(initial=0.0) < 1e-6
        from helpers import judge_remainder
        judge_remainder(asm_ref, dq, {"qpOASES exact": rq, "qpOASES": rd}, label=f"tasks {off} inactive")
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


def _id_stack_with_gain_matrices(B, seed, force_type):
    """config 5 with GAIN MATRICES on its two hand tasks (acceleration::Cartesian::setGains, Cartesian.cpp:152-173): the
    left hand with GainType::Acceleration (Kp, Kd as they are), the right hand with GainType::Force when asked (Mi Kp,
    Mi Kd with Mi = J B^-1 J', plus a virtual force)"""
    plan, leaf = synth.make_id_stack(B, seed=seed)
    rng = np.random.default_rng(seed + 1)
    nv = leaf["model"]["nv"]

    def spd(scale):
        M = rng.normal(size=(6, 6))
        return scale * (M @ M.T / 6.0 + np.eye(6))
    Kp = [spd(1.0), spd(0.5)]; Kd = [spd(0.3), spd(0.2)]
    Bi = np.linalg.inv(leaf["model"]["B"])
    f_virtual = rng.normal(0.0, 2.0, size=(B, 6))
    extra = {"Kp": Kp, "Kd": Kd, "Bi": Bi, "f": f_virtual, "J": [np.ascontiguousarray(leaf["A"][0][:, 3 + 6 * k:9 + 6 * k, :nv]) for k in range(2)]}
    for k in range(2):
        t = plan.levels[0][1 + k]
        t.acc_gain_matrices = True
        p0, p1, p2 = leaf["task"][0][1 + k]
        if k == 1 and force_type:
            Gp, Gd, a_add = oracle_mod().force_gains(extra["J"][1], Bi, Kp[1], Kd[1], f_virtual)
            p2 = a_add                                            # a_ref was zero: a_ref + Mi f
        else:
            Gp = np.broadcast_to(Kp[k], (B, 6, 6)); Gd = np.broadcast_to(Kd[k], (B, 6, 6)); a_add = None
        leaf["task"][0][1 + k] = (np.concatenate([p0, Gp.reshape(B, 36), Gd.reshape(B, 36)], axis=1), p1, p2)
    return plan, leaf, extra


def oracle_mod():
    from oracle import pyoracle
    return pyoracle


@pytest.mark.parametrize("force_type", [False, True])
def test_acceleration_cartesian_gain_matrices_gpu(force_type, oracle, gpu_device):
    """Kp / Kd matrices and the Force gain type of acceleration::Cartesian (src/tasks/acceleration/Cartesian.cpp:152-173; round 2
    fixed Kp = Kd = I): b through the update kernel against the restated formula AND a by-hand numpy evaluation of the
    reference's expression; for the Force type the gains come from the device producer osot_id_force_gains (Mi = J B^-1 J',
    :517-524) and are compared with the restatement; the solved x against the eiQuadProg restatement and qpOASES"""
    from helpers import parity_census
    from opensot_amd import dynamics
    B = 96
    plan, leaf, ex = _id_stack_with_gain_matrices(B, 61, force_type)
    asm = oracle.assemble(plan, leaf)
    # by hand, the reference's expression: b = a_ref + lambda2 [Mi] Kd vel_err + lambda [Mi] Kp pose_err [+ Mi f] - Jdot qdot
    for k in range(2):
        t = plan.levels[0][1 + k]
        p0, jdq, a_ref = leaf["task"][0][1 + k]
        pe, ve = p0[:, :6], p0[:, 6:12]
        Mi = np.einsum("brk,bkl,bsl->brs", ex["J"][k], ex["Bi"], ex["J"][k]) if (k == 1 and force_type) else np.broadcast_to(np.eye(6), (B, 6, 6))
        want = t.lam2 * np.einsum("brs,st,bt->br", Mi, ex["Kd"][k], ve) + t.lam * np.einsum("brs,st,bt->br", Mi, ex["Kp"][k], pe) - jdq
        if k == 1 and force_type:
            want = want + np.einsum("brs,bs->br", Mi, ex["f"])
        np.testing.assert_allclose(asm["b"][0][:, 3 + 6 * k:9 + 6 * k], want, rtol=0, atol=1e-11)
    st = BatchedStack(plan, B, device=0)
    dev = st.load_leaf(leaf)
    if force_type:       # the right hand's gains and Mi f from the device producer, into the leaf array / a_ref in place
        p0d, _, _ = dev["task"][0][2]
        Gref = p0d[:, 12:].clone()
        p0d[:, 12:] = 0.0
        a_ref = torch.zeros((B, 6), dtype=torch.float64, device=st.device)
        f64 = dict(dtype=torch.float64, device=st.device)
        dynamics.force_gains(torch.as_tensor(ex["J"][1], **f64).contiguous(), torch.as_tensor(ex["Bi"], **f64).contiguous(),
                             ex["Kp"][1], ex["Kd"][1], p0d, 6, f_virtual=torch.as_tensor(ex["f"], **f64).contiguous(), a_ref=a_ref)
        torch.cuda.synchronize()
        assert (p0d[:, 12:] - Gref).abs().max().item() < 1e-12 * max(1.0, Gref.abs().max().item())
        assert (a_ref - dev["task"][0][2][2]).abs().max().item() < 1e-12 * max(1.0, a_ref.abs().max().item())
    st.update(dev); st.solve(B)
    torch.cuda.synchronize()
    np.testing.assert_allclose(st.b[0][:B].cpu().numpy(), asm["b"][0], rtol=0, atol=1e-12)
    x = st.dq[:B].cpu().numpy()
    assert (st.status[:B].cpu().numpy() == 0).all()
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=0)
    assert (ref["status"] == 1).all() and np.abs(x - ref["dq"]).max() < 1e-8
    if oracle.ref_available():
        rq = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=0)
        within, rule, fails = parity_census(asm, x, [("qpOASES", rq), ("eiQuadProg", ref)], tol=1e-6, label=f"acc gains force={force_type}")
        assert not fails
