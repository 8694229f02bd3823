"""The equality-only front-end (OpenSoT::solvers::eHQP, src/solvers/eHQP.cpp): oracle/pyehqp.py (numpy SVD) against the iHQP
path where the two front-ends must agree (parity of the restatement is otherwise UNPINNED: the reference has no robot-free
eHQP vector), the emulated kernel (opensot_amd/csrc/osot_ehqp.h) against the restatement, and the HIP kernel through the
C-ABI on the GPU."""
import numpy as np
import pytest

from helpers import emu_ehqp
from opensot_amd import synth
from opensot_amd.plan import StackPlan
from oracle import pyehqp


def _unconstrained(plan):
    """the same stack without its constraints and bounds (what eHQP looks at)"""
    return StackPlan(n=plan.n, levels=plan.levels, bounds=[], rowblocks=[], eps_abs=plan.eps_abs)


def _generic(B, n, rows, seed, postural=True):
    plan, leaf = synth.make_generic_stack(B, n, rows, n_eq=0, n_ineq=0, seed=seed, box=0.0, postural_last=postural, eps_factor=2e2)
    return plan, leaf


def test_damped_pinv_weights_follow_the_reference():
    """getDampedPinv (eHQP.cpp:124-146): plain inverse above sigma_min, Tikhonov with lambda = min(sigma) below, rank by
    Eigen's relative threshold"""
    s = np.array([2.0, 1.0, 0.5])
    np.testing.assert_allclose(pyehqp.damped_pinv_weights(s, 1e-12), 1.0 / s)
    s = np.array([2.0, 1.0, 1e-14])                      # min below sigma_min -> damped, and the last one is out of the rank
    w = pyehqp.damped_pinv_weights(s, 1e-12)
    lam = 1e-14
    np.testing.assert_allclose(w[:2], s[:2] / (s[:2] ** 2 + lam ** 2))
    assert w[2] == 0.0
    s = np.array([1.0, 0.3])                             # a large sigma_min switches the damping on for a healthy matrix
    w = pyehqp.damped_pinv_weights(s, 0.5)
    np.testing.assert_allclose(w[0], 1.0 / (1.0 + 0.09))
    assert w[1] == 0.0                                   # 0.3 < 0.5 * 1.0: below the rank threshold


@pytest.mark.parametrize("n,rows,seed", [(7, [3, 2], 1), (12, [4, 5], 2), (32, [3, 24], 3), (20, [6], 4)])
def test_restatement_agrees_with_ihqp_where_both_pose_the_same_problem(n, rows, seed, oracle):
    """no constraints, full-row-rank task levels and a Postural last level: the lexicographic least-squares solution is
    unique, the damped pseudo-inverse chain (eHQP) and the eps-regularised QP cascade (iHQP) must both return it"""
    plan, leaf = _generic(6, n, rows, seed)
    asm = oracle.assemble(plan, leaf)
    e = pyehqp.ehqp_solve(asm)
    r = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
    assert (r["status"] == 1).all()
    assert np.abs(e["dq"] - r["dq"]).max() < 1e-6
    # (the intermediate x_k differ in the directions the level leaves free: minimum norm here, eps-damped there; what both
    #  define is the level's task value)
    for k in range(plan.L - 1):
        Ak = asm["A"][k]
        assert np.abs(np.einsum("brn,bn->br", Ak, e["x_levels"][:, k] - r["x_levels"][:, k])).max() < 1e-6


@pytest.mark.parametrize("n,rows,seed", [(7, [3, 2], 1), (12, [4, 5], 2), (32, [3, 24], 3), (20, [6], 4), (9, [2, 2, 2], 5),
                                         (35, [12, 6], 6), (40, [10, 12], 7), (50, [15], 8), (64, [20, 30], 9)])
def test_emulated_kernel_vs_restatement(n, rows, seed, oracle):
    """the QR kernel (round 3: no eigen-decomposition, n <= 64) against the numpy-SVD restatement: 1e-12 (the Gram / eigen
    kernel of round 2 was held to 1e-9: it squared the condition number)"""
    plan, leaf = _generic(5, n, rows, seed)
    asm = oracle.assemble(plan, leaf)
    e = pyehqp.ehqp_solve(asm)
    dq, st, xl = emu_ehqp(plan, asm)
    assert (st == 0).all()
    assert np.abs(dq - e["dq"]).max() < 1e-12
    assert np.abs(xl - e["x_levels"]).max() < 1e-12


def test_stack_without_a_full_rank_last_level(oracle):
    """task levels only (4 + 5 rows in 16 variables): the minimum-norm choice among the solutions of the last level --
    x stays in the row spaces -- and the hierarchy A_j x_k = A_j x_j"""
    plan, leaf = _generic(5, 16, [4, 5], 7, postural=False)
    asm = oracle.assemble(plan, leaf)
    e = pyehqp.ehqp_solve(asm)
    dq, st, xl = emu_ehqp(plan, asm)
    assert np.abs(dq - e["dq"]).max() < 1e-9
    A0, A1 = asm["A"][0], asm["A"][1]
    assert np.abs(np.einsum("brn,bn->br", A0, dq) - asm["b"][0]).max() < 1e-9          # level 0 met exactly
    assert np.abs(np.einsum("brn,bn->br", A0, xl[:, 1] - xl[:, 0])).max() < 1e-9       # and kept by level 1
    assert np.abs(np.einsum("brn,bn->br", A1, dq) - asm["b"][1]).max() < 1e-9          # 9 rows in 16 variables: level 1 too


def test_constraints_and_bounds_are_ignored_like_in_the_reference(oracle):
    """eHQP.cpp:45-47: '# OF CONSTRAINTS: 0', '# OF BOUNDS: 0' -- a box that binds for iHQP changes nothing here"""
    plan, leaf = synth.make_generic_stack(4, 10, [3, 3], n_eq=0, n_ineq=2, seed=11, box=0.01, eps_factor=2e2)
    asm = oracle.assemble(plan, leaf)
    dq, st, xl = emu_ehqp(plan, asm)
    free = pyehqp.ehqp_solve(oracle.assemble(*_strip(plan, leaf)))
    assert np.abs(dq - free["dq"]).max() < 1e-9
    assert np.abs(dq).max() > 0.01                        # (outside the box the stack carries)


def _strip(plan, leaf):
    p2 = _unconstrained(plan)
    l2 = dict(leaf); l2["bound"] = []; l2["rows"] = []
    l2.pop("C", None)
    return p2, l2


def test_inactive_level_is_skipped(oracle):
    plan, leaf = _generic(4, 12, [4, 5], 2)
    asm = oracle.assemble(plan, leaf)
    act = [1, 0, 1]
    e = pyehqp.ehqp_solve(asm, level_active=act)
    dq, st, xl = emu_ehqp(plan, asm, level_active=act)
    assert np.abs(dq - e["dq"]).max() < 1e-9


@pytest.mark.parametrize("n,split,second", [(12, (2, 3), 4), (40, (4, 5), 6)])
def test_inactive_task_counts_as_zero_rows(n, split, second, oracle):
    """round 4: Task::setActive(false) on a task inside a level (Task.h:383-387: its A and b are zero while it is inactive).
    The front-end takes the flags of osot_solver_set_task_active and gives the task's rows weight zero; against the
    restatement run on the same stack with that task's A and b zeroed (QR kernel, 32- and 64-lane instantiation)."""
    from opensot_amd import abi
    from opensot_amd.plan import Task
    B, (ra, rb) = 5, split
    base, bleaf = synth.make_generic_stack(B, n, [ra + rb, second], n_eq=0, n_ineq=0, seed=n, box=0.0, postural_last=True, eps_factor=2e2)
    lv0 = [Task(abi.TASK_GENERIC, ra, name="a"), Task(abi.TASK_GENERIC, rb, name="b")]      # the first level as two tasks
    plan = StackPlan(n=n, levels=[lv0] + list(base.levels[1:]), bounds=[], rowblocks=[], eps_abs=base.eps_abs)
    leaf = dict(bleaf)
    b0 = bleaf["task"][0][0][0]
    leaf["task"] = [[(b0[:, :ra].copy(), None, None), (b0[:, ra:].copy(), None, None)]] + list(bleaf["task"][1:])
    asm = oracle.assemble(plan, leaf)
    full = pyehqp.ehqp_solve(asm)
    dq, st, xl = emu_ehqp(plan, asm)
    assert (st == 0).all() and np.abs(dq - full["dq"]).max() < 1e-9
    # the restatement of the same stack WITHOUT the task (rows removed).  With the rows zeroed instead -- what the reference's
    # Task::update leaves -- numpy's / Eigen's thin V still holds one (implementation-defined) completion vector per zero row
    # and P_i = P_(i-1) - V V' spends a null-space direction on each: see oracle/pyehqp.py on rank-deficient levels; the
    # rank-revealing QR of the kernel removes the row space only, i.e. it treats the task as absent.
    ref_asm = dict(asm)
    ref_asm["A"] = [a.copy() if a is not None else None for a in asm["A"]]
    ref_asm["b"] = [b.copy() for b in asm["b"]]
    ref_asm["w"] = [None if w is None else w.copy() for w in asm["w"]]
    ref_asm["A"][0] = np.ascontiguousarray(asm["A"][0][:, :ra, :])
    ref_asm["b"][0] = np.ascontiguousarray(asm["b"][0][:, :ra])
    if ref_asm["w"][0] is not None:
        ref_asm["w"][0] = np.ascontiguousarray(asm["w"][0][:, :ra])
    ref_asm["m"] = [ra] + list(asm["m"][1:]) if "m" in asm else None
    e = pyehqp.ehqp_solve(ref_asm)
    dq, st, xl = emu_ehqp(plan, asm, task_active={(0, 1): False})          # the second task of the first level
    assert (st == 0).all() and np.abs(dq - e["dq"]).max() < 1e-9
    assert np.abs(full["dq"] - e["dq"]).max() > 1e-4                       # (the task mattered)
    # ... and it is bit-identical to handing the kernel zeroed rows (the reference's arrays)
    z = dict(asm)
    z["A"] = [a.copy() if a is not None else None for a in asm["A"]]
    z["b"] = [b.copy() for b in asm["b"]]
    z["A"][0][:, ra:ra + rb, :] = 0.0
    z["b"][0][:, ra:ra + rb] = 0.0
    dqz, _, _ = emu_ehqp(plan, z)
    assert np.array_equal(dq, dqz)


def test_weights_and_dense_weights(oracle):
    """W = L L' enters as L'A, L'b in the reference; the kernel takes W A and W b (the update kernel's outputs)"""
    plan, leaf = synth.make_feature_stack(4, seed=3, body_frame=False, dense=True, bands=False, candidates=0, many_blocks=False)
    asm = oracle.assemble(plan, leaf)
    assert any(w is not None for w in asm["Wdense"])
    asm["WA"] = [None if W is None else W @ asm["A"][k] for k, W in enumerate(asm["Wdense"])]
    asm["Wb"] = [None if W is None else np.einsum("brq,bq->br", W, asm["b"][k]) for k, W in enumerate(asm["Wdense"])]
    e = pyehqp.ehqp_solve(asm)
    dq, st, xl = emu_ehqp(plan, asm)
    assert np.abs(dq - e["dq"]).max() < 1e-8


def test_overdetermined_level_is_weighted_least_squares(oracle):
    """8 rows in 6 variables: the level cannot be met, x = argmin |A x - b|_W (where the weights matter)"""
    plan, leaf = _generic(5, 6, [8], 9, postural=False)
    rng = np.random.default_rng(0)
    asm = oracle.assemble(plan, leaf)
    asm["w"][0] = rng.uniform(0.2, 3.0, size=asm["w"][0].shape)
    e = pyehqp.ehqp_solve(asm)
    dq, st, xl = emu_ehqp(plan, asm)
    assert np.abs(dq - e["dq"]).max() < 1e-9
    for i in range(5):
        sw = np.sqrt(asm["w"][0][i])
        want = np.linalg.lstsq(sw[:, None] * asm["A"][0][i], sw * asm["b"][0][i], rcond=None)[0]
        assert np.abs(dq[i] - want).max() < 1e-9


def _nearly_singular(B, seed):
    """two task levels in 12 variables whose first level has a row that is ALMOST a combination of two others (smallest singular
    value ~ 1e-4 of the largest): the plain pseudo-inverse amplifies b by 1e4 there, the damped one (sigma_min = 1e-2) does not"""
    plan, leaf = _generic(B, 12, [4, 3], seed)
    rng = np.random.default_rng(seed)
    A0 = leaf["A"][0].copy()
    A0[:, 3] = 0.5 * A0[:, 0] - 0.25 * A0[:, 1] + 1.0e-4 * rng.normal(size=A0[:, 3].shape)
    leaf["A"][0] = A0
    return plan, leaf


def _well_defined(asm):
    """instances whose first level is NEARLY singular (sigma_min ~ 1e-4 sigma_max), not exactly: on an exactly rank-deficient level
    Eigen's thin V holds an implementation-defined completion vector that enters the projector"""
    return np.array([np.linalg.svd(a, compute_uv=False)[-1] > 1e-8 for a in asm["A"][0]])


def test_sigma_min_switches_the_damped_inverse_on_emulated(oracle):
    """eHQP::setSigmaMin (eHQP.cpp:124-146, 156-166): a sigma_min above the smallest singular value must reach the
    kernel -- the QR kernel never damps, so such a call runs the Gram / eigen kernel, which does"""
    plan, leaf = _nearly_singular(6, 21)
    asm = oracle.assemble(plan, leaf)
    plain = pyehqp.ehqp_solve(asm)
    damped = pyehqp.ehqp_solve(asm, sigma_min=1.0e-2)
    assert np.abs(plain["dq"] - damped["dq"]).max() > 1.0e-2          # (the option matters on this stack)
    dq, st, xl = emu_ehqp(plan, asm, sigma_min=1.0e-2)
    sel = _well_defined(asm)
    assert (st == 0).all() and sel.sum() >= 4
    assert np.abs(dq - damped["dq"])[sel].max() < 1e-7
    dq0, _, _ = emu_ehqp(plan, asm)                                   # the default still takes the QR kernel and its accuracy
    assert np.abs(dq0 - plain["dq"])[sel].max() < 1e-9


@pytest.mark.gpu
def test_sigma_min_switches_the_damped_inverse_on_gpu(oracle, gpu_device):
    import torch
    from opensot_amd.solver import BatchedStack
    B = 32
    plan, leaf = _nearly_singular(B, 22)
    asm = oracle.assemble(plan, leaf)
    damped = pyehqp.ehqp_solve(asm, sigma_min=1.0e-2)
    st = BatchedStack(plan, B, device=0)
    st.load_assembled(asm)
    st.solve_ehqp(B, sigma_min=1.0e-2)
    torch.cuda.synchronize()
    sel = _well_defined(asm)
    assert (st.status[:B].cpu().numpy() == 0).all() and sel.sum() >= B // 2
    assert np.abs(st.dq[:B].cpu().numpy() - damped["dq"])[sel].max() < 1e-7
    assert np.abs(pyehqp.ehqp_solve(asm)["dq"] - damped["dq"])[sel].max() > 1.0e-2


def test_sigma_min_beyond_32_variables_is_refused(oracle):
    """the damped inverse lives in the n <= 32 kernel: a non-default sigma_min on a larger stack is an error, not a silent no-op"""
    import ctypes as C
    from opensot_amd import abi
    from helpers import emu_lib
    plan, leaf = _generic(2, 40, [10, 12], 7)
    asm = oracle.assemble(plan, leaf)
    with pytest.raises(AssertionError):
        emu_ehqp(plan, asm, sigma_min=1.0e-2)
    emu_ehqp(plan, asm)                                               # (the default is served by the QR kernel)


@pytest.mark.gpu
@pytest.mark.parametrize("n,rows,seed", [(7, [3, 2], 1), (32, [3, 24], 3), (35, [12, 6], 6), (50, [15], 8), (64, [20, 30], 9)])
def test_ehqp_gpu_vs_restatement(n, rows, seed, oracle, gpu_device):
    import torch
    from opensot_amd.solver import BatchedStack
    B = 64
    plan, leaf = _generic(B, n, rows, seed)
    asm = oracle.assemble(plan, leaf)
    st = BatchedStack(plan, B, device=0)
    st.load_assembled(asm)
    st.solve_ehqp(B)
    torch.cuda.synchronize()
    e = pyehqp.ehqp_solve(asm)
    assert (st.status[:B].cpu().numpy() == 0).all()
    assert np.abs(st.dq[:B].cpu().numpy() - e["dq"]).max() < 1e-12
    assert np.abs(st.x_levels[:B].cpu().numpy() - e["x_levels"]).max() < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("n,split,second", [(12, (2, 3), 4), (40, (4, 5), 6)])
def test_inactive_task_gpu(n, split, second, oracle, gpu_device):
    """osot_solver_set_task_active + osot_ehqp_solve (round 4; refused before): the inactive task's rows count with weight zero
    -- the restatement of the stack without the task, and the answer of the full stack again after re-activating it"""
    import torch
    from opensot_amd import abi
    from opensot_amd.plan import Task
    from opensot_amd.solver import BatchedStack
    B, (ra, rb) = 48, split
    base, bleaf = synth.make_generic_stack(B, n, [ra + rb, second], n_eq=0, n_ineq=0, seed=n, box=0.0, postural_last=True, eps_factor=2e2)
    lv0 = [Task(abi.TASK_GENERIC, ra, name="a"), Task(abi.TASK_GENERIC, rb, name="b")]
    plan = StackPlan(n=n, levels=[lv0] + list(base.levels[1:]), bounds=[], rowblocks=[], eps_abs=base.eps_abs)
    leaf = dict(bleaf)
    b0 = bleaf["task"][0][0][0]
    leaf["task"] = [[(b0[:, :ra].copy(), None, None), (b0[:, ra:].copy(), None, None)]] + list(bleaf["task"][1:])
    asm = oracle.assemble(plan, leaf)
    full = pyehqp.ehqp_solve(asm)
    ref = dict(asm)
    ref["A"] = [np.ascontiguousarray(asm["A"][0][:, :ra, :])] + list(asm["A"][1:])
    ref["b"] = [np.ascontiguousarray(asm["b"][0][:, :ra])] + list(asm["b"][1:])
    ref["w"] = [None if asm["w"][0] is None else np.ascontiguousarray(asm["w"][0][:, :ra])] + list(asm["w"][1:])
    ref["m"] = [ra] + list(asm["m"][1:]); ref["ma"] = [ra] + list(asm["ma"][1:])
    e = pyehqp.ehqp_solve(ref)
    st = BatchedStack(plan, B, device=0)
    st.load_assembled(asm)
    st.set_task_active(0, 1, False)
    st.solve_ehqp(B)
    torch.cuda.synchronize()
    assert (st.status[:B].cpu().numpy() == 0).all()
    assert np.abs(st.dq[:B].cpu().numpy() - e["dq"]).max() < 1e-11
    st.solve_nhqp(B)                                            # round 5: the null-space front-end takes the flag too (zero rows of A, Task.h:383-387)
    torch.cuda.synchronize()
    assert (st.status[:B].cpu().numpy() == 0).all()
    assert np.isfinite(st.dq[:B].cpu().numpy()).all()
    st.set_task_active(0, 1, True)
    st.level_active = [1, 0, 1]                                 # ... and iHQP's setActiveStack, which the reference's nHQP does not have
    with pytest.raises(RuntimeError, match="setActiveStack"):
        st.solve_nhqp(B)
    st.level_active = None
    st.solve_ehqp(B)
    torch.cuda.synchronize()
    assert np.abs(st.dq[:B].cpu().numpy() - full["dq"]).max() < 1e-11


@pytest.mark.gpu
def test_ehqp_gpu_benchmark_stack_and_torch_api(oracle, gpu_device):
    """BASELINE config 3 through update + eHQP on the device (torch_api.eHQP mirrors pyopensot.eHQP); against the
    restatement on the assembled arrays, and against iHQP without the box"""
    import torch
    from opensot_amd import torch_api
    B = 128
    plan, leaf = synth.make_velocity_stack("C3", B, seed=5)
    sol = torch_api.eHQP(plan, B)
    dev = sol.stack.load_leaf(leaf)
    dq = sol.solve(dev)
    torch.cuda.synchronize()
    asm = oracle.assemble(plan, leaf)
    e = pyehqp.ehqp_solve(asm)
    assert np.abs(dq.cpu().numpy() - e["dq"]).max() < 1e-8
    assert sol.getSigmaMin() == 1e-12
    sol.setSigmaMin(-1.0)
    assert sol.getSigmaMin() == 1e-12


@pytest.mark.gpu
def test_frontends_random_sweep_gpu(gpu_device):
    """tests/stress_frontends.py: eHQP and nHQP kernels against their restatements over random stacks (n = 2..32, one to three
    task levels, with and without a Postural level, constraints for nHQP): the eigen-solver on many sizes and spectra"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "stress_frontends.py"), "5", "60"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "60 stacks x 24 instances: 0 with a mismatch" in out.stdout, out.stdout[-2000:]

