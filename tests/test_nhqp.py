"""The null-space front-end: OpenSoT::solvers::nHQP (src/solvers/nHQP.cpp).
  * oracle/pynhqp.py (numpy SVD + the reference's qpOASES per level) restates it; PARITY UNPINNED against the reference (its
    own nHQP test needs a robot model) -- the restatement is pinned to the iHQP path instead: with both regularisations
    off the two front-ends pose the same lexicographic problem and must agree on full-rank stacks;
  * the device kernels (Jacobi SVD in LDS, QP in null-space coordinates through the batched back-end kernel) against the
    restatement, with the reference's default options (A/b regularisation at 0.05, selective null-space regularisation)."""
import numpy as np
import pytest

from helpers import emu_nhqp
from opensot_amd import synth


def test_restatement_agrees_with_ihqp_without_regularisations(oracle):
    if not oracle.ref_available():
        pytest.skip("needs oracle/_ref (qpOASES)")
    from oracle import pynhqp
    plan, leaf = synth.make_velocity_stack("C3", 12, seed=3)
    asm = oracle.assemble(plan, leaf)
    assert pynhqp.free_variables(asm) == [32, 29, 5]          # nHQP.cpp:88-103 on full-rank levels: 32 -> 29 -> 5
    ri = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1, termination_tolerance=10 * 2.221e-16)
    rn = pynhqp.nhqp_solve(asm, ab_regularization=False, selective_ns_regularization=False, termination_tolerance=10 * 2.221e-16)
    ok = (ri["status"] == 1) & (rn["status"] == 1)
    assert ok.sum() >= 0.9 * ok.size and np.abs(ri["dq"][ok] - rn["dq"][ok]).max() < 1e-6       # (two CPU restatements against each other)
    # with the reference's default regularisations nHQP is a DIFFERENT answer wherever they bite (a lifted singular value,
    # a bound active above the last level): that is the reference's own behaviour, not a defect of either path
    rd = pynhqp.nhqp_solve(asm)
    assert np.abs(rd["dq"] - ri["dq"]).max() > 1e-4


@pytest.mark.parametrize("cfg,opts", [("C3", {}), ("C3", dict(ab_regularization=False, selective_ns_regularization=False)),
                                      ("C4", {}), ("C2", {}), ("C3", dict(min_sv_ratio=0.2)), ("C3", dict(min_sv_ratio=0.0))])
def test_emulated_kernels_match_restatement(cfg, opts, oracle):
    from oracle import pynhqp
    B = 6
    plan, leaf = synth.make_velocity_stack(cfg, B, seed=17)
    asm = oracle.assemble(plan, leaf)
    backend = "qpoases" if oracle.ref_available() else "eiqp"
    kw = dict(opts)
    if "min_sv_ratio" not in kw:
        kw["min_sv_ratio"] = pynhqp.DEFAULT_MIN_SV_RATIO
    ref = pynhqp.nhqp_solve(asm, backend=backend, termination_tolerance=10 * 2.221e-16, **kw)
    dq, st = emu_nhqp(plan, asm, **opts)
    ok = ref["status"] == 1
    print(f"[nHQP emulated {cfg} {opts}] restatement solved {int(ok.sum())}/{B}; device solved {int((st == 0).sum())}/{B}; "
          f"max |dq - restatement| over the compared ones {np.abs(dq[ok] - ref['dq'][ok]).max():.2e}")
    assert (st[ok] == 0).all()
    assert np.abs(dq[ok] - ref["dq"][ok]).max(initial=0.0) < 1e-7
    # an instance the restatement's qpOASES gave up on is not compared; the device's answer there still has to be a point
    # inside the box (or a reported failure with dq = 0)
    assert np.isfinite(dq).all() and (dq >= asm["l"] - 1e-7).all() and (dq <= asm["u"] + 1e-7).all()


@pytest.mark.parametrize("n,rows,n_ineq,seed", [(35, [3, 12], 3, 2), (40, [10, 12], 4, 3), (50, [6, 20, 10], 0, 4), (64, [20, 30], 5, 5)])
def test_more_than_32_variables_emulated(n, rows, n_ineq, seed, oracle):
    """33 .. 64 variables (round 4: osot_nhqp_prepare64_kernel; the reference's COMAN has 35 coordinates, config 5 has 50): the
    64-column level preparation + the 64-lane QP kernel + the accumulation against the numpy-SVD restatement"""
    from oracle import pynhqp
    plan, leaf = synth.make_generic_stack(4, n, rows, n_eq=0, n_ineq=n_ineq, seed=seed, box=0.4)
    asm = oracle.assemble(plan, leaf)
    ref = pynhqp.nhqp_solve(asm, backend="qpoases" if oracle.ref_available() else "eiqp", termination_tolerance=10 * 2.221e-16)
    dq, st = emu_nhqp(plan, asm)
    ok = ref["status"] == 1
    assert ok.all() and (st == 0).all()
    assert np.abs(dq - ref["dq"]).max() < 1e-9


WIDE_STACKS = [(48, [40], 0, 1, {}),                      # 40 rows in 48 variables: row side, k = 40
               (35, [50], 0, 2, {}),                      # the shape of the reference's S1: 50 rows in 35 variables, one level
               (35, [50], 3, 3, dict(min_sv_ratio=0.3)),  # ... with global rows, lifting a third of the singular values
               (40, [36, 3], 4, 4, {}),                   # a wide level followed by a narrow one (its null space goes down the cascade)
               (64, [64], 0, 5, dict(ab_regularization=False))]


@pytest.mark.parametrize("n,rows,n_ineq,seed,opts", WIDE_STACKS)
def test_level_wider_than_32_emulated(n, rows, n_ineq, seed, opts, oracle):
    """min(rows, free variables) of a level beyond 32 -- refused until round 5, the reference's own stack S1 among them
    (examples/cpp/coman_ik.cpp:425-431: one level of 50 rows in 35 variables): osot_nhqp_prepare_wide_kernel (Jacobi iteration on the
    full Gram matrix, U explicit) against the numpy-SVD restatement"""
    from oracle import pynhqp
    plan, leaf = synth.make_generic_stack(3, n, rows, n_eq=0, n_ineq=n_ineq, seed=seed, box=0.4, postural_last=False)
    asm = oracle.assemble(plan, leaf)
    kw = dict(opts); kw.setdefault("min_sv_ratio", pynhqp.DEFAULT_MIN_SV_RATIO)
    ref = pynhqp.nhqp_solve(asm, backend="qpoases" if oracle.ref_available() else "eiqp", termination_tolerance=10 * 2.221e-16, **kw)
    dq, st = emu_nhqp(plan, asm, **opts)
    ok = ref["status"] == 1
    assert ok.all() and (st == 0).all()
    assert np.abs(dq - ref["dq"]).max() < 1e-8


def test_per_level_switches_emulated(oracle):
    """nHQP::setPerformAbRegularization(level, .), setPerformSelectiveNullSpaceRegularization(level, .) and
    setMinSingularValueRatio(std::vector<double>) (nHQP.cpp:127-152, 206-221): per-level entries of osot_nhqp_options against the
    restatement with the same per-level lists; and they are not the solver-wide setting"""
    from oracle import pynhqp
    plan, leaf = synth.make_velocity_stack("C3", 5, seed=23)
    asm = oracle.assemble(plan, leaf)
    opts = dict(ab_regularization=[True, False, True], selective_ns_regularization=[False, True, True], min_sv_ratio=[None, 0.3, 0.1])
    ref = pynhqp.nhqp_solve(asm, backend="qpoases" if oracle.ref_available() else "eiqp", termination_tolerance=10 * 2.221e-16, **opts)
    dq, st = emu_nhqp(plan, asm, **opts)
    ok = ref["status"] == 1
    assert ok.any() and (st[ok] == 0).all()
    assert np.abs(dq[ok] - ref["dq"][ok]).max() < 1e-7
    dflt, _ = emu_nhqp(plan, asm)
    assert np.abs(dq - dflt).max() > 1e-6


def test_inactive_task_is_a_block_of_zero_rows_emulated(oracle):
    """Task::setActive(false) through nHQP (refused until round 5): the task's rows of A are zero rows, b stays (Task.h:383-387) --
    bit for bit what handing the kernels the zeroed rows gives, and the restatement's residuals of the OTHER levels (a level
    with zero rows is rank deficient: its own lifted triplets are implementation-defined, see test_rank_deficient_level_*)"""
    from oracle import pynhqp
    plan, leaf = synth.make_velocity_stack("C3", 4, seed=29)
    asm = oracle.assemble(plan, leaf)
    fv = pynhqp.free_variables(asm, 0)      # (fixed at construction, with every task active: nHQP.cpp:6-117; setActive does not change them)
    dq, st = emu_nhqp(plan, asm, task_active={(1, 2): False}, ab_regularization=False, free_vars=fv)
    zeroed = dict(asm); zeroed["A"] = [a.copy() if a is not None else None for a in asm["A"]]
    zeroed["A"][1][:, 12:18, :] = 0.0
    dq0, st0 = emu_nhqp(plan, zeroed, ab_regularization=False, free_vars=fv)
    assert (st == 0).all() and (st0 == 0).all()
    np.testing.assert_array_equal(dq, dq0)
    ref = pynhqp.nhqp_solve(zeroed, backend="qpoases" if oracle.ref_available() else "eiqp", termination_tolerance=10 * 2.221e-16,
                            ab_regularization=False, free_vars=fv)
    ok = ref["status"] == 1
    assert ok.any()
    # levels 0 and 1 (the active rows): the same task residuals as the restatement's
    for k, rows in ((0, slice(0, 3)), (1, slice(0, 12)), (1, slice(18, 24))):
        rd = np.einsum("bri,bi->br", asm["A"][k][:, rows], dq) - asm["b"][k][:, rows]
        rr = np.einsum("bri,bi->br", asm["A"][k][:, rows], ref["dq"]) - asm["b"][k][:, rows]
        assert np.abs(rd[ok] - rr[ok]).max() < 1e-6
    dq_all, _ = emu_nhqp(plan, asm, ab_regularization=False)
    assert np.abs(dq - dq_all).max() > 1e-6           # (the switch does something)


def test_min_sv_ratio_is_honoured_without_the_flag(oracle):
    """osot_nhqp_options.min_sv_ratio = 0.2 with min_sv_ratio_is_set left at 0 used to be silently replaced by 0.05"""
    import ctypes as C
    from helpers import emu_lib
    from opensot_amd import abi
    from oracle import pynhqp
    plan, leaf = synth.make_velocity_stack("C3", 4, seed=17)
    asm = oracle.assemble(plan, leaf)
    want, _ = emu_nhqp(plan, asm, min_sv_ratio=0.2)                 # (the helper sets the flag)
    dflt, _ = emu_nhqp(plan, asm)
    assert np.abs(want - dflt).max() > 1e-6                         # (the option matters on this stack)
    dq = np.zeros_like(want); st = np.full(4, -1, dtype=np.int32)
    qb = abi.QpBatch(); qb.B = 4
    keep = []
    for k in range(asm["L"]):
        for name in ("A", "b", "w"):
            a = asm[name][k]
            if a is not None:
                a = np.ascontiguousarray(a, dtype=np.float64); keep.append(a); getattr(qb, name)[k] = a.ctypes.data
    for name in ("C", "lo", "up", "l", "u"):
        a = asm[name]
        if a is not None and a.size:
            a = np.ascontiguousarray(a, dtype=np.float64); keep.append(a); setattr(qb, name, a.ctypes.data)
    qb.dq, qb.status = dq.ctypes.data, st.ctypes.data
    opt = abi.NhqpOptions(); opt.min_sv_ratio = 0.2                 # min_sv_ratio_is_set stays 0
    L = emu_lib()
    L.emu_nhqp_solve.argtypes = [C.POINTER(abi.PlanDesc), C.POINTER(abi.QpBatch), C.POINTER(abi.NhqpOptions), C.c_void_p]
    pd = plan.to_c()
    assert L.emu_nhqp_solve(C.byref(pd), C.byref(qb), C.byref(opt), None) == 0
    np.testing.assert_array_equal(dq, want)


def test_small_generic_stack_and_no_free_variables(oracle):
    """a 12-variable generic stack (row side and column side of the SVD both occur); a stack that runs out of free
    variables is refused like the reference's constructor does (nHQP.cpp:32-35)"""
    from oracle import pynhqp
    plan, leaf = synth.make_generic_stack(5, 12, [4, 5], n_eq=0, n_ineq=3, seed=2, box=0.4)
    asm = oracle.assemble(plan, leaf)
    ref = pynhqp.nhqp_solve(asm, backend="qpoases" if oracle.ref_available() else "eiqp", termination_tolerance=10 * 2.221e-16)
    dq, st = emu_nhqp(plan, asm)
    ok = ref["status"] == 1
    assert ok.all() and (st == 0).all() and np.abs(dq - ref["dq"]).max() < 1e-7
    import ctypes as C
    from helpers import emu_lib
    from opensot_amd import abi
    plan2, leaf2 = synth.make_generic_stack(2, 8, [5, 4], seed=1, postural_last=True)    # 8 -> 3 -> -1 free variables
    qb = abi.QpBatch(); qb.B = 2
    opt = abi.NhqpOptions()
    L = emu_lib()
    L.emu_nhqp_solve.argtypes = [C.POINTER(abi.PlanDesc), C.POINTER(abi.QpBatch), C.POINTER(abi.NhqpOptions)]
    pd = plan2.to_c()
    assert L.emu_nhqp_solve(C.byref(pd), C.byref(qb), C.byref(opt)) == abi.ERR_INVALID


def _duplicated_row_stack(oracle, wide=False):
    """wide (round 4): a 12-row first level in 24 variables -- min(rows, free variables) >= 10, where the level preparation first asks
    for the singular VALUES only (sym_eigvals32), finds one at noise level and falls back to the full decomposition"""
    from oracle import pynhqp
    plan, leaf = synth.make_generic_stack(5, 24, [12, 6], n_eq=0, n_ineq=3, seed=3, box=0.4) if wide else \
        synth.make_generic_stack(5, 12, [4, 5], n_eq=0, n_ineq=3, seed=2, box=0.4)
    asm = oracle.assemble(plan, leaf)
    fv = pynhqp.free_variables(asm)                       # fixed at construction from the full-rank stack (nHQP.cpp:6-117)
    asm["A"][0][:, 3, :] = asm["A"][0][:, 0, :]           # then row 3 of the first level duplicates row 0: rank 3 of 4
    asm["b"][0][:, 3] = asm["b"][0][:, 0]
    return plan, asm, fv


def _check_rank_deficient_level(asm, dq, st, ref):
    """what is DEFINED on a rank-deficient level: regularize_A_b (nHQP.cpp:236-279) lifts the null triplet with the v_i of
    Eigen's full V, an implementation-defined unit vector of the level's null space (numpy's LAPACK picks another one, this
    build a Householder completion), and that direction is then closed to the levels below.  So dq itself is comparable
    through the first level's task only: its residual must be the restatement's (2-6e-2 before the fix, where
    the lifted singular value sat on a noise vector INSIDE the row space), and the answer must respect the box."""
    assert (st == 0).all() and (ref["status"] == 1).all()
    r_dev = np.einsum("bij,bj->bi", asm["A"][0], dq) - asm["b"][0]
    r_ref = np.einsum("bij,bj->bi", asm["A"][0], ref["dq"]) - asm["b"][0]
    # (an instance whose first level runs into the box keeps a residual in the restatement too: the same one up to what the
    #  implementation-defined null vector of the lifted triplet moves, a few per cent of it)
    rr, rd = np.abs(r_ref).max(axis=1), np.abs(r_dev).max(axis=1)
    free = rr < 1e-5
    assert free.any() and (rd[free] < 1e-5).all()
    assert (np.abs(rd[~free] - rr[~free]) < 0.1 * rr[~free]).all()
    assert (dq >= asm["l"] - 1e-9).all() and (dq <= asm["u"] + 1e-9).all()


@pytest.mark.parametrize("wide", [False, True])
def test_rank_deficient_level_emulated(wide, oracle):
    from oracle import pynhqp
    plan, asm, fv = _duplicated_row_stack(oracle, wide)
    ref = pynhqp.nhqp_solve(asm, backend="qpoases" if oracle.ref_available() else "eiqp", termination_tolerance=10 * 2.221e-16,
                            free_vars=fv)
    dq, st = emu_nhqp(plan, asm, free_vars=fv)
    _check_rank_deficient_level(asm, dq, st, ref)
    # the lifted NULL triplet only adds a penalty on a direction nothing else moves along: with the null vector taken from the
    # completion the answer is the one without the A/b regularisation (no other singular value is below the threshold
    # here); with a noise vector inside the row space it was 0.2 away
    dq_off, st_off = emu_nhqp(plan, asm, free_vars=fv, ab_regularization=False)
    assert (st_off == 0).all()
    if not wide:          # (the wide stack has instances whose first level runs into the box: there the penalty does move the answer)
        assert np.abs(dq - dq_off).max() < 1e-4


def _duplicated_row_wide_level(oracle):
    """a level beyond 32 on both sides (40 rows in 48 variables: osot_nhqp_prepare_wide_kernel) whose row 3 duplicates row 0: rank 39 of
    40, ONE null triplet -- the Gram-Schmidt completion behind the batched genuine triplets of the wide preparation"""
    from oracle import pynhqp
    plan, leaf = synth.make_generic_stack(4, 48, [40], n_eq=0, n_ineq=0, seed=7, box=0.4, postural_last=False)
    asm = oracle.assemble(plan, leaf)
    fv = pynhqp.free_variables(asm)
    asm["A"][0][:, 3, :] = asm["A"][0][:, 0, :]
    asm["b"][0][:, 3] = asm["b"][0][:, 0]
    return plan, asm, fv


def test_rank_deficient_wide_level_emulated(oracle):
    from oracle import pynhqp
    plan, asm, fv = _duplicated_row_wide_level(oracle)
    ref = pynhqp.nhqp_solve(asm, backend="qpoases" if oracle.ref_available() else "eiqp", termination_tolerance=10 * 2.221e-16,
                            free_vars=fv)
    dq, st = emu_nhqp(plan, asm, free_vars=fv)
    _check_rank_deficient_level(asm, dq, st, ref)


@pytest.mark.gpu
def test_rank_deficient_wide_level_gpu(oracle, gpu_device):
    import torch
    from oracle import pynhqp
    from opensot_amd.solver import BatchedStack
    plan, asm, fv = _duplicated_row_wide_level(oracle)
    B = asm["B"]
    st = BatchedStack(plan, B, device=0)
    st.load_assembled(asm)
    st.solve_nhqp(B, free_vars=fv, min_sv_ratio=pynhqp.DEFAULT_MIN_SV_RATIO)
    torch.cuda.synchronize()
    ref = pynhqp.nhqp_solve(asm, backend="qpoases" if oracle.ref_available() else "eiqp", termination_tolerance=10 * 2.221e-16,
                            free_vars=fv)
    _check_rank_deficient_level(asm, st.dq[:B].cpu().numpy(), st.status[:B].cpu().numpy(), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("wide", [False, True])
def test_rank_deficient_level_gpu(wide, oracle, gpu_device):
    import torch
    from oracle import pynhqp
    from opensot_amd.solver import BatchedStack
    plan, asm, fv = _duplicated_row_stack(oracle, wide)
    B = asm["B"]
    st = BatchedStack(plan, B, device=0)
    st.load_assembled(asm)
    st.solve_nhqp(B, free_vars=fv, min_sv_ratio=pynhqp.DEFAULT_MIN_SV_RATIO)
    torch.cuda.synchronize()
    ref = pynhqp.nhqp_solve(asm, backend="qpoases" if oracle.ref_available() else "eiqp", termination_tolerance=10 * 2.221e-16,
                            free_vars=fv)
    _check_rank_deficient_level(asm, st.dq[:B].cpu().numpy(), st.status[:B].cpu().numpy(), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,opts", [("C3", {}), ("C4", {}), ("C3", dict(ab_regularization=False, selective_ns_regularization=False))])
def test_nhqp_gpu(cfg, opts, oracle, gpu_device):
    """osot_nhqp_solve through the C-ABI: against the restatement (reference qpOASES per level) on a sample of the batch;
    without the regularisations also against the device's own iHQP cascade (the two front-ends must agree there)"""
    import torch
    from oracle import pynhqp
    from opensot_amd.solver import BatchedStack
    B = 512
    plan, leaf = synth.make_velocity_stack(cfg, B, seed=23)
    st = BatchedStack(plan, B, device=0)
    st.update(st.load_leaf(leaf))
    st.solve_nhqp(B, **opts)
    torch.cuda.synchronize()
    dq = st.dq[:B].cpu().numpy().copy(); status = st.status[:B].cpu().numpy().copy()
    sub = slice(0, B, 16)
    sl = {"B": len(range(*sub.indices(B))), "A": [a[sub] if a is not None else None for a in leaf["A"]],
          "task": [[tuple(None if x is None else x[sub] for x in t) for t in lev] for lev in leaf["task"]],
          "bound": [tuple(None if x is None else x[sub] for x in t) for t in leaf["bound"]],
          "rows": [tuple(None if x is None else x[sub] for x in t) for t in leaf["rows"]]}
    asm = oracle.assemble(plan, sl)
    kw = dict(opts); kw.setdefault("min_sv_ratio", pynhqp.DEFAULT_MIN_SV_RATIO)
    ref = pynhqp.nhqp_solve(asm, backend="qpoases" if oracle.ref_available() else "eiqp", termination_tolerance=10 * 2.221e-16, **kw)
    ok = ref["status"] == 1
    print(f"[nHQP gpu {cfg} {opts}] restatement solved {int(ok.sum())}/{ok.size} of the sample; device solved "
          f"{int((status == 0).sum())}/{B}; max |dq - restatement| over the compared ones {np.abs(dq[sub][ok] - ref['dq'][ok]).max():.2e}")
    assert (status[sub][ok] == 0).all()
    assert np.abs(dq[sub][ok] - ref["dq"][ok]).max(initial=0.0) < 1e-7
    assert np.isfinite(dq).all() and (dq[sub] >= asm["l"] - 1e-7).all() and (dq[sub] <= asm["u"] + 1e-7).all()
    if opts:
        st.solve(B); torch.cuda.synchronize()
        both = (status == 0) & (st.status[:B].cpu().numpy() == 0)
        assert np.abs(dq[both] - st.dq[:B].cpu().numpy()[both]).max(initial=0.0) < 1e-6


def test_symmetric_eigen_solver_emulated():
    """sym_eig32 (Householder tridiagonalisation + implicit QL; round 3: the rotations of a sweep are recorded and applied to
    the eigenvector rows after it) on its own, through the emulator: G V = V diag(lambda), V'V = I, eigenvalues against
    numpy -- full rank, rank deficient (a duplicated row and column) and the sizes the front-ends meet (3, 5, 24, 32)"""
    import ctypes as C
    from helpers import emu_lib
    L = emu_lib()
    rng = np.random.default_rng(0)
    for k in (2, 3, 5, 10, 12, 17, 24, 32):
        for trial in range(3):
            A = rng.normal(size=(k, k + 3 * (trial % 2)))
            if trial == 2 and k > 3:
                A[1] = A[0]                                   # rank deficient: a cluster of round-off-level eigenvalues
            G = A @ A.T
            K = np.zeros((32, 33)); K[:k, :k] = G
            E = np.zeros((32, 33))
            assert L.emu_sym_eig32(K.ctypes.data_as(C.c_void_p), E.ctypes.data_as(C.c_void_p), k) == 0
            lam, V = np.diag(K)[:k], E[:k, :k]
            scale = max(1.0, np.abs(G).max())
            assert np.abs(G @ V - V * lam[None, :]).max() < 1e-13 * scale
            assert np.abs(V.T @ V - np.eye(k)).max() < 1e-13
            assert np.abs(np.sort(lam) - np.linalg.eigvalsh(G)).max() < 1e-13 * scale
            # round 4: the same through sym_eig32_fast (bisection + twisted factorisation per lane, QL only for clustered spectra);
            # orthogonality there is eps |T| / gap with gaps down to 1e-7 |T|
            K2 = np.zeros((32, 33)); K2[:k, :k] = G
            E2 = np.zeros((32, 33))
            assert L.emu_sym_eig32_fast(K2.ctypes.data_as(C.c_void_p), E2.ctypes.data_as(C.c_void_p), k) == 0
            lam2, V2 = np.diag(K2)[:k], E2[:k, :k]
            assert np.abs(G @ V2 - V2 * lam2[None, :]).max() < 1e-12 * scale
            assert np.abs(V2.T @ V2 - np.eye(k)).max() < 1e-9
            assert np.abs(np.sort(lam2) - np.linalg.eigvalsh(G)).max() < 1e-13 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("n,rows,n_ineq,seed", [(35, [3, 12], 3, 2), (50, [6, 20, 10], 0, 4), (64, [20, 30], 5, 5)])
def test_more_than_32_variables_gpu(n, rows, n_ineq, seed, oracle, gpu_device):
    """the 64-column nHQP path on the device against the restatement (sample of the batch) -- see the emulated test of the same name"""
    import torch
    from opensot_amd.solver import BatchedStack
    from oracle import pynhqp
    B = 64
    plan, leaf = synth.make_generic_stack(B, n, rows, n_eq=0, n_ineq=n_ineq, seed=seed, box=0.4)
    asm = oracle.assemble(plan, leaf)
    st = BatchedStack(plan, B, device=0)
    st.load_assembled(asm)
    st.solve_nhqp(B)
    torch.cuda.synchronize()
    dq = st.dq[:B].cpu().numpy(); status = st.status[:B].cpu().numpy()
    sub = slice(0, B, 8)
    sl = dict(asm); sl["B"] = 8
    for key in ("A", "b", "w", "c"):
        sl[key] = [None if a is None else a[sub] for a in asm[key]]
    for key in ("C", "lo", "up", "l", "u"):
        sl[key] = None if asm[key] is None else asm[key][sub]
    ref = pynhqp.nhqp_solve(sl, backend="qpoases" if oracle.ref_available() else "eiqp", termination_tolerance=10 * 2.221e-16)
    ok = ref["status"] == 1
    assert ok.all() and (status == 0).all()
    assert np.abs(dq[sub] - ref["dq"]).max() < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("n,rows,n_ineq,seed,opts", WIDE_STACKS)
def test_level_wider_than_32_gpu(n, rows, n_ineq, seed, opts, oracle, gpu_device):
    """osot_nhqp_prepare_wide_kernel on the device (see the emulated test of the same name), a sample of the batch against the
    numpy-SVD restatement"""
    import torch
    from opensot_amd.solver import BatchedStack
    from oracle import pynhqp
    B = 40
    plan, leaf = synth.make_generic_stack(B, n, rows, n_eq=0, n_ineq=n_ineq, seed=seed, box=0.4, postural_last=False)
    asm = oracle.assemble(plan, leaf)
    st = BatchedStack(plan, B, device=0)
    st.load_assembled(asm)
    st.solve_nhqp(B, **opts)
    torch.cuda.synchronize()
    dq = st.dq[:B].cpu().numpy(); status = st.status[:B].cpu().numpy()
    sub = slice(0, B, 8)
    sl = dict(asm); sl["B"] = 5
    for key in ("A", "b", "w", "c"):
        sl[key] = [None if a is None else a[sub] for a in asm[key]]
    for key in ("C", "lo", "up", "l", "u"):
        sl[key] = None if asm[key] is None else asm[key][sub]
    kw = dict(opts); kw.setdefault("min_sv_ratio", pynhqp.DEFAULT_MIN_SV_RATIO)
    ref = pynhqp.nhqp_solve(sl, backend="qpoases" if oracle.ref_available() else "eiqp", termination_tolerance=10 * 2.221e-16, **kw)
    assert (ref["status"] == 1).all() and (status == 0).all()
    assert np.abs(dq[sub] - ref["dq"]).max() < 1e-8


@pytest.mark.gpu
def test_per_level_switches_and_inactive_task_gpu(oracle, gpu_device):
    """the per-level entries of osot_nhqp_options and Task::setActive(false) through osot_nhqp_solve on the device: the emulator's
    answers (tests above: pinned to the restatement there) to round-off"""
    import torch
    from opensot_amd.solver import BatchedStack
    B = 5
    plan, leaf = synth.make_velocity_stack("C3", B, seed=23)
    asm = oracle.assemble(plan, leaf)
    opts = dict(ab_regularization=[True, False, True], selective_ns_regularization=[False, True, True], min_sv_ratio=[None, 0.3, 0.1])
    want, wst = emu_nhqp(plan, asm, **opts)
    st = BatchedStack(plan, B, device=0)
    st.load_assembled(asm)
    st.solve_nhqp(B, **opts)
    torch.cuda.synchronize()
    ok = wst == 0
    assert (st.status[:B].cpu().numpy()[ok] == 0).all()
    assert np.abs(st.dq[:B].cpu().numpy()[ok] - want[ok]).max() < 1e-8
    want, wst = emu_nhqp(plan, asm, task_active={(1, 2): False}, ab_regularization=False)
    st.set_task_active(1, 2, False)
    st.solve_nhqp(B, ab_regularization=False)
    torch.cuda.synchronize()
    assert (wst == 0).all() and (st.status[:B].cpu().numpy() == 0).all()
    assert np.abs(st.dq[:B].cpu().numpy() - want).max() < 1e-8


def _dense_weight_stack(n, rows0, rows1, B, seed, box):
    """a two-level stack whose first level holds a task with a NON-DIAGONAL weight (Task::setWeight(W)) next to a plain one"""
    from opensot_amd import abi
    from opensot_amd.plan import StackPlan, Task
    rng = np.random.default_rng(seed)
    ra = rows0 // 2
    base, bleaf = synth.make_generic_stack(B, n, [rows0, rows1] if rows1 else [rows0], n_eq=0, n_ineq=0, seed=seed, box=box, postural_last=False, eps_factor=1e6)
    lv0 = [Task(abi.TASK_GENERIC, ra, name="a", dense_weight=True), Task(abi.TASK_GENERIC, rows0 - ra, name="b")]
    plan = StackPlan(n=n, levels=[lv0] + list(base.levels[1:]), bounds=list(base.bounds), rowblocks=[], eps_abs=base.eps_abs)
    leaf = dict(bleaf)
    b0 = bleaf["task"][0][0][0]
    if rows0 > n:          # an over-determined level with an INCONSISTENT right-hand side: the weight decides the answer
        b0 = b0 + rng.normal(0.0, 0.05, b0.shape)
    leaf["task"] = [[(b0[:, :ra].copy(), None, None), (b0[:, ra:].copy(), None, None)]] + list(bleaf["task"][1:])
    Mw = rng.normal(size=(B, ra, ra))
    Wa = Mw @ np.transpose(Mw, (0, 2, 1)) / ra + 0.5 * np.eye(ra)
    leaf["W"] = [[Wa, None]] + [[None] * len(lv) for lv in base.levels[1:]]
    return plan, leaf


@pytest.mark.parametrize("n,rows0,rows1,box", [(12, 6, 4, 0.0), (12, 16, 0, 0.0), (20, 10, 6, 0.3), (20, 26, 0, 0.2), (35, 12, 8, 0.0), (35, 40, 0, 0.0)])
def test_nhqp_dense_weight_emulated(oracle, n, rows0, rows1, box):
    """round 5 -- nHQP with a non-diagonal task weight (nHQP.cpp:381-382: H = AN'W AN, g = -AN'W b0 on the REGULARISED A N and b0):
    osot_nhqp_options.level_W carries the level's full weight matrix; the kernels run their H / g stage on diag(W) and add the
    off-diagonal part column by column.  n <= 32, 33..64 and a level wider than 32 on both sides (all three preparation kernels),
    with and without a box; against the restatement (oracle/pynhqp.py, parity unpinned)"""
    from oracle import pynhqp
    B = 6
    plan, leaf = _dense_weight_stack(n, rows0, rows1, B, seed=100 + n + rows0, box=box)
    asm = oracle.assemble(plan, leaf)
    assert asm["Wdense"][0] is not None
    ref = pynhqp.nhqp_solve(asm)
    LW = [asm["Wdense"][0]] + [None] * (asm["L"] - 1)
    dq, st = emu_nhqp(plan, asm, level_W=LW)
    assert (st == 0).all()
    assert np.abs(dq - ref["dq"]).max() < 1e-8 * max(1.0, np.abs(ref["dq"]).max())
    if rows0 > n:
        # an over-determined level with an inconsistent right-hand side, A/b regularisation OFF (with it on regularize_A_b zeroes the
        # part of b0 outside range(U_k), nHQP.cpp:254-257, and the level is solved exactly whatever W): the weight decides the answer,
        # and its diagonal alone gives another one
        ref2 = pynhqp.nhqp_solve(asm, ab_regularization=False)
        dq2, st2 = emu_nhqp(plan, asm, level_W=LW, ab_regularization=False)
        assert (st2 == 0).all() and np.abs(dq2 - ref2["dq"]).max() < 1e-8 * max(1.0, np.abs(ref2["dq"]).max())
        asm_d = dict(asm); asm_d["Wdense"] = None
        asm_d["w"] = [np.ascontiguousarray(np.diagonal(asm["Wdense"][0], axis1=1, axis2=2))] + list(asm["w"][1:])
        assert np.abs(pynhqp.nhqp_solve(asm_d, ab_regularization=False)["dq"] - ref2["dq"]).max() > 1e-4


def test_nhqp_dense_weight_without_level_W_is_refused(oracle):
    """a level with a non-diagonal weight and no level_W: refused with the reason, not solved on W A / W b"""
    plan, leaf = _dense_weight_stack(12, 6, 4, 4, seed=7, box=0.0)
    asm = oracle.assemble(plan, leaf)
    with pytest.raises(AssertionError):
        emu_nhqp(plan, asm)


@pytest.mark.gpu
@pytest.mark.parametrize("n,rows0,rows1", [(20, 10, 6), (20, 26, 0), (35, 12, 8), (35, 40, 0)])
def test_nhqp_dense_weight_gpu(oracle, gpu_device, n, rows0, rows1):
    """the same through osot_nhqp_solve on the device (BatchedStack.solve_nhqp(level_W=...)): against the restatement"""
    import torch
    from oracle import pynhqp
    from opensot_amd.solver import BatchedStack
    B = 24
    plan, leaf = _dense_weight_stack(n, rows0, rows1, B, seed=300 + n + rows0, box=0.0)
    asm = oracle.assemble(plan, leaf)
    ref = pynhqp.nhqp_solve(asm)
    st = BatchedStack(plan, B, device=0)
    st.load_assembled(asm)
    W0 = torch.as_tensor(np.ascontiguousarray(asm["Wdense"][0]), dtype=torch.float64, device=torch.device("cuda", 0))
    st.solve_nhqp(B, level_W=[W0] + [None] * (plan.L - 1))
    torch.cuda.synchronize()
    assert (st.status[:B].cpu().numpy() == 0).all()
    assert np.abs(st.dq[:B].cpu().numpy() - ref["dq"]).max() < 1e-8 * max(1.0, np.abs(ref["dq"]).max())
    with pytest.raises(RuntimeError, match="level_W"):
        st.solve_nhqp(B)


@pytest.mark.gpu
def test_nhqp_option_ranges_and_list_lengths_are_refused(gpu_device):
    """nHQP::setMinSingularValueRatio throws outside [0, 1] and for a vector whose size is not the number of layers
    (nHQP.cpp:127-152); here OSOT_ERR_INVALID / ValueError instead of a silently different lifting rule"""
    import torch
    from opensot_amd import abi, synth
    from opensot_amd.solver import BatchedStack
    B = 8
    plan, leaf = synth.make_velocity_stack("C3", B, seed=3)
    st = BatchedStack(plan, B, device=0, want_levels=False)
    st.update(st.load_leaf(leaf))
    for bad in (-0.1, 1.5, float("nan")):
        with pytest.raises(RuntimeError):
            st.solve_nhqp(B, min_sv_ratio=bad)
        with pytest.raises(RuntimeError):
            st.solve_nhqp(B, min_sv_ratio=[0.05, bad, 0.05])
    with pytest.raises(ValueError):
        st.solve_nhqp(B, min_sv_ratio=[0.05, 0.05])                 # two entries, three levels
    with pytest.raises(ValueError):
        st.solve_nhqp(B, ab_regularization=[True, True, True, True])
    W = torch.eye(3, dtype=torch.float32, device="cuda").repeat(B, 1, 1)
    with pytest.raises(ValueError):
        st.solve_nhqp(B, level_W=[W, None, None])                   # float32
    st.solve_nhqp(B, min_sv_ratio=[0.05, 0.0, 1.0])                 # the ends of the range are values
    torch.cuda.synchronize()
    assert (st.status[:B] == 0).all()
    assert st.resident_waves_nhqp() >= 256 and st.resident_waves() >= 256
