"""Round-2 surface on the emulator (no GPU): the update kernel against the oracle's assembly (it used to be checked on
hardware only), body-frame Cartesian tasks, per-row TaskToConstraint bands, collision rows chosen among more candidates
than rows, non-diagonal weight matrices, Task::setActive, more than four row blocks, the inverse-dynamics producers and
computedTorque.  tests/test_gpu_features_r2.py repeats them through the C-ABI on hardware."""
import ctypes as C

import numpy as np
import pytest

from helpers import emu_cascade, emu_lib, emu_update
from opensot_amd import abi, synth


def _check_update(plan, leaf, oracle, task_active=None):
    asm = oracle.assemble(plan, leaf)
    res = emu_update(plan, leaf)
    for k in range(plan.L):
        np.testing.assert_allclose(res["b"][k], asm["b"][k], rtol=0, atol=1e-15)
        np.testing.assert_array_equal(res["w"][k], asm["w"][k])
    if plan.bounds:
        np.testing.assert_array_equal(res["l"], asm["l"]); np.testing.assert_array_equal(res["u"], asm["u"])
    if plan.nc:
        from opensot_amd.solver import stored_rows
        if plan.nc_stored:
            np.testing.assert_array_equal(res["C"], stored_rows(plan, asm["C"]))
        np.testing.assert_allclose(res["lo"], asm["lo"], rtol=0, atol=1e-15)
        np.testing.assert_allclose(res["up"], asm["up"], rtol=0, atol=1e-15)
    return asm, res


@pytest.mark.parametrize("cfg", ["C2", "C3", "C4", "C5"])
def test_update_kernel_vs_oracle_assembly(cfg, oracle):
    plan, leaf = synth.make_id_stack(6, seed=3) if cfg == "C5" else synth.make_velocity_stack(cfg, 6, seed=3)
    _check_update(plan, leaf, oracle)


@pytest.mark.parametrize("n", [32, 40, 60])
def test_feature_stack_update_and_cascade(n, oracle):
    """body-frame b, per-row bands, candidate ordering, six row blocks and a full weight matrix in one stack: assembly
    equal to the oracle's, W A and W b equal to numpy, cascade against the eiQuadProg restatement and qpOASES.
    n = 40 / 60: the same through the 64-lane cascade on its short (n <= 54) and full LDS layouts"""
    B = 6
    plan, leaf = synth.make_feature_stack(B, seed=5, n=n)
    assert len(plan.rowblocks) == 6 and plan.dense_level(1) and not plan.dense_level(0)
    asm, res = _check_update(plan, leaf, oracle)
    # the collision block really had to choose: some candidates are closer than the first `rows` ones
    d = leaf["rows"][1][1]
    assert (np.sort(d, axis=1)[:, :8] != d[:, :8]).any()
    W = asm["Wdense"][1]
    np.testing.assert_allclose(res["WA"][1], W @ asm["A"][1], rtol=0, atol=1e-14)
    np.testing.assert_allclose(res["Wb"][1], np.einsum("brq,bq->br", W, asm["b"][1]), rtol=0, atol=1e-15)
    asm["WA"] = res["WA"]; asm["Wb"] = res["Wb"]
    dq, xl, st, it = emu_cascade(plan, asm)
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
    assert (st == 0).all() and (ref["status"] == 1).all()
    assert np.abs(dq - ref["dq"]).max() < 1e-9
    if oracle.ref_available():
        rq = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1, termination_tolerance=10 * 2.221e-16)
        ok = rq["status"] == 1      # (run to the exact optimum qpOASES itself gives up on some instances)
        assert ok.mean() > 0.5 and np.abs(dq[ok] - rq["dq"][ok]).max() < 1e-7
    # the full W matters: with its diagonal only the answer is a different one
    asm_d = dict(asm); asm_d["Wdense"] = [None] * 3; asm_d.pop("WA"); asm_d.pop("Wb")
    asm_d["w"] = [w.copy() for w in asm["w"]]
    asm_d["w"][1][:, :6] = np.einsum("bii->bi", W)[:, :6]
    assert np.abs(emu_cascade(plan, asm_d)[0] - dq).max() > 1e-6


@pytest.mark.parametrize("off", [[(1, 0)], [(1, 1), (1, 2)], [(2, 0)], [(0, 0)]])
def test_task_set_active(off, oracle):
    """Task::setActive(false) (Task.h:232-239, 383-387): the task's A counts as zero -- nothing in H / g, void optimality
    rows -- for stored blocks and for the implicit Postural block; the oracle zeroes the rows as the reference does"""
    B = 5
    plan, leaf = synth.make_velocity_stack("C3", B, seed=12)
    ta = {kj: False for kj in off}
    asm_ref = oracle.assemble(plan, leaf, task_active=ta)       # the reference's way: zero rows in A
    asm = oracle.assemble(plan, leaf)
    dq, xl, st, it = emu_cascade(plan, asm, task_active=ta)
    assert (st == 0).all()
    # the flag is the zeroed Jacobian (stored blocks; the implicit Postural block has no rows to zero).  To round-off only:
    # void rows are absent from the lower levels' equality lists, zero rows are present and skipped as dependent, which
    # changes how the rows group into the Gauss-Jordan's panels of four
    if all(not plan.levels[k][j].implicit for k, j in off):
        dz, xz, sz, _ = emu_cascade(plan, asm_ref)
        assert (sz == 0).all() and np.abs(dq - dz).max() < 1e-12
    assert np.abs(dq - emu_cascade(plan, asm)[0]).max() > 1e-6      # and it is not the all-active answer
    # witness: the real qpOASES (the zero rows are linearly dependent EQUALITY rows at the levels below, which the
    # eiQuadProg restatement -- like the routine it restates, eiquadprog.hpp:246-251 -- does not handle)
    if oracle.ref_available():
        rq = oracle.ihqp_solve_batch(asm_ref, oracle.BE_QPOASES_REF, nthreads=1, termination_tolerance=10 * 2.221e-16)
        rd = oracle.ihqp_solve_batch(asm_ref, oracle.BE_QPOASES_REF, nthreads=1)
        e = np.minimum(np.where(rq["status"] == 1, np.abs(dq - rq["dq"]).max(axis=1), np.inf),
                       np.where(rd["status"] == 1, np.abs(dq - rd["dq"]).max(axis=1), np.inf))
        assert np.isfinite(e).all() and e.max() < 1e-6


def _id_model(leaf, keep):
    md = leaf["model"]
    B, nv = md["B"].shape[0], md["nv"]
    Jc = np.ascontiguousarray(md["Jc"]); Bm = np.ascontiguousarray(md["B"]); h = np.ascontiguousarray(md["h"])
    keep += [Jc, Bm, h]
    m = abi.IdModel()
    m.B, m.nv, m.n_contacts, m.contact_dim, m.floating_base = B, nv, Jc.shape[1], Jc.shape[2], 1
    m.Bm, m.h, m.Jc = Bm.ctypes.data, h.ctypes.data, Jc.ctypes.data
    return m


def test_inverse_dynamics_producers_and_computed_torque(oracle):
    """[B_u, -J_f'], [B, -Jc'] and [J 0] written by the producer kernel equal what the synthetic generator stacks on the
    host (DynamicFeasibility.cpp:22-46, TorqueLimits.cpp:25-46); tau = B qddot + h - sum Jc'F equals numpy, and the
    floating-base acceptance flag follows InverseDynamics.cpp:83-92"""
    B = 5
    plan, leaf = synth.make_id_stack(B, seed=9)
    n, nv = plan.n, leaf["model"]["nv"]
    keep = []
    m = _id_model(leaf, keep)
    L = emu_lib()
    vp = C.c_void_p
    L.emu_id_rows.argtypes = [C.POINTER(abi.IdModel), vp, C.c_longlong, vp, C.c_longlong, C.c_int, vp, vp, vp, vp]
    L.emu_computed_torque.argtypes = [C.POINTER(abi.IdModel), vp, vp, vp, C.c_double]
    Cst = np.full((B, plan.nc_stored, n), 7.0)
    o_dyn, o_tau = plan.rows_stored_offset(0), plan.rows_stored_offset(2)
    # task Jacobians -> [J 0] rows of A_0
    A0 = np.full((B, plan.ma(0), n), 7.0)
    Js = [np.ascontiguousarray(leaf["A"][0][:, o:o + r, :nv]) for o, r in ((0, 3), (3, 6), (9, 6))]
    Jp = (vp * 3)(*[j.ctypes.data for j in Js]); Jr = (C.c_int * 3)(3, 6, 6)
    Ad = (vp * 3)(*[A0.ctypes.data + 8 * o * n for o in (0, 3, 9)]); As = (C.c_longlong * 3)(*[plan.ma(0) * n] * 3)
    assert L.emu_id_rows(C.byref(m), Cst.ctypes.data + 8 * o_dyn * n, plan.nc_stored * n, Cst.ctypes.data + 8 * o_tau * n,
                         plan.nc_stored * n, 3, Jp, Jr, Ad, As) == 0
    np.testing.assert_array_equal(Cst[:, o_dyn:o_dyn + 6], leaf["C"][0])
    np.testing.assert_array_equal(Cst[:, o_tau:o_tau + nv], leaf["C"][2])
    np.testing.assert_array_equal(A0, leaf["A"][0])
    # computed torque on the oracle's solution
    asm = oracle.assemble(plan, leaf)
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
    assert (ref["status"] == 1).all()
    x = np.ascontiguousarray(ref["dq"])
    tau = np.zeros((B, nv)); ok = np.full(B, -1, dtype=np.int32)
    assert L.emu_computed_torque(C.byref(m), x.ctypes.data, tau.ctypes.data, ok.ctypes.data, 1e-2) == 0
    np.testing.assert_allclose(tau, synth.computed_torque(leaf, x), rtol=0, atol=1e-11)
    assert (ok == 1).all() and np.abs(tau[:, :6]).max() < 1e-8
    x[2, 0] += 1.0     # break the floating-base balance of one instance: "Floating Base Wrench is not 0!"
    assert L.emu_computed_torque(C.byref(m), x.ctypes.data, tau.ctypes.data, ok.ctypes.data, 1e-2) == 0
    assert ok[2] == 0 and (np.delete(ok, 2) == 1).all()


def test_acceleration_task_gain_matrices_emulated(oracle):
    """Kp / Kd matrices of acceleration::Cartesian (src/tasks/acceleration/Cartesian.cpp:152-160): the update kernel's b
    (emulator) against the oracle's restatement and a by-hand numpy evaluation of the reference's expression"""
    from helpers import emu_update
    B = 5
    plan, leaf = synth.make_id_stack(B, seed=62)
    rng = np.random.default_rng(63)
    Kp = rng.normal(size=(6, 6)); Kd = rng.normal(size=(6, 6))      # (any matrices: the expression is linear in them)
    t = plan.levels[0][1]
    t.acc_gain_matrices = True
    p0, jdq, a_ref = leaf["task"][0][1]
    leaf["task"][0][1] = (np.concatenate([p0, np.tile(Kp.reshape(1, 36), (B, 1)), np.tile(Kd.reshape(1, 36), (B, 1))], axis=1), jdq, a_ref)
    asm = oracle.assemble(plan, leaf)
    want = t.lam2 * p0[:, 6:12] @ Kd.T + t.lam * p0[:, :6] @ Kp.T - jdq
    np.testing.assert_allclose(asm["b"][0][:, 3:9], want, rtol=0, atol=1e-12)
    got = emu_update(plan, leaf)
    np.testing.assert_allclose(got["b"][0], asm["b"][0], rtol=0, atol=1e-13)
    # the other tasks of the level keep their scalar gains
    plan0, leaf0 = synth.make_id_stack(B, seed=62)
    asm0 = oracle.assemble(plan0, leaf0)
    np.testing.assert_array_equal(asm["b"][0][:, :3], asm0["b"][0][:, :3])
    np.testing.assert_array_equal(asm["b"][0][:, 9:], asm0["b"][0][:, 9:])
