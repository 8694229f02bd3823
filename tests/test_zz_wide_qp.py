"""The explicit-QP surface beyond 64 variables (round 6: osot_qp_solve_batch and the osot_backend_* plugin take 65 .. 128 variables;
opensot_amd/csrc/osot_qp_big.h: one 256-thread workgroup per QP).  include/OpenSoT/Task.h:47-565 has no limit on the variables, so a
45-DoF robot or a floating-base inverse-dynamics stack with five or more contacts is a plugin problem of 65 .. 128 columns.

CPU tests run the SAME source compiled for the host with a team of one thread (tests/emu/big_host.cpp) against the oracle; the GPU
tests run the product through the C-ABI against the oracle, qpOASES where oracle/_ref exists, and the host build."""
import ctypes as C

import numpy as np
import pytest

from helpers import big_host_solve, id_like_levels, kkt_check, random_qp, ref_qpoases_solve
from opensot_amd import abi


def _one(args, i):
    H, g, A, lA, uA, l, u = args
    pick = lambda a: None if a is None else a[i]
    return pick(H), pick(g), pick(A), pick(lA), pick(uA), pick(l), pick(u)


@pytest.mark.parametrize("n,nc,n_eq,box", [(65, 0, 0, True), (70, 30, 6, True), (96, 60, 20, False), (128, 80, 40, True), (128, 200, 10, True)])
def test_wide_solver_source_on_the_host_against_the_oracle(n, nc, n_eq, box, oracle):
    rng = np.random.default_rng(7 * n + nc)
    args = random_qp(rng, 3, n, nc, n_eq, box=box)
    for i in range(3):
        q = _one(args, i)
        st, x, iters = big_host_solve(*q, 1e-9)
        ok, xo, _ = oracle.backend_solve(*q, 1e-9)
        assert st == 0 and ok
        assert np.abs(x - xo).max() < 1e-8 * max(1.0, np.abs(xo).max())
        assert kkt_check(*q, x, 1e-9) < 1e-6
        assert iters >= n_eq


def test_wide_solver_on_rank_deficient_hessians_host(oracle):
    """H = M'M with fewer rows than variables: the eps of the back-end and the box make it well posed (QPOasesBackEnd.cpp:253-255)"""
    rng = np.random.default_rng(5)
    for n, r in ((72, 15), (100, 3), (128, 64)):
        M = rng.normal(size=(r, n)); H = M.T @ M; g = -M.T @ rng.normal(size=r)
        l = -rng.uniform(0.05, 0.5, size=n); u = rng.uniform(0.05, 0.5, size=n)
        A = rng.normal(size=(20, n)); lA = -rng.uniform(0.05, 1.0, size=20); uA = rng.uniform(0.05, 1.0, size=20)
        eps = 2.221e-7
        st, x, _ = big_host_solve(H, g, A, lA, uA, l, u, eps)
        ok, xo, _ = oracle.backend_solve(H, g, A, lA, uA, l, u, eps)
        assert st == 0 and ok
        assert np.abs(x - xo).max() < 1e-6
        rq = ref_qpoases_solve(H, g, A, lA, uA, l, u, eps / 2.221e-13)
        if rq is not None:
            assert rq[0] and np.abs(x - rq[1]).max() < 1e-6
        assert kkt_check(H, g, A, lA, uA, l, u, x, eps, tol=1e-6) < 1e-5


def test_wide_solver_statuses_host():
    n = 80
    rng = np.random.default_rng(2)
    M = rng.normal(size=(n + 3, n)); H = M.T @ M; g = rng.normal(size=n)
    # two inconsistent equality rows
    a = rng.normal(size=n)
    A = np.vstack([a, a]); lA = np.array([1.0, 2.0]); uA = lA.copy()
    st, x, _ = big_host_solve(H, g, A, lA, uA, None, None, 1e-9)
    assert st == 1 and not x.any()
    # the same row twice with the same right-hand side: redundant and consistent
    st, x, _ = big_host_solve(H, g, A, np.array([1.0, 1.0]), np.array([1.0, 1.0]), None, None, 1e-9)
    assert st == 0 and abs(a @ x - 1.0) < 1e-9
    # an inequality that contradicts the box
    l = -np.ones(n); u = np.ones(n)
    row = np.ones((1, n))
    st, _, _ = big_host_solve(H, g, row, np.array([2.0 * n]), np.array([np.inf]), l, u, 1e-9)
    assert st == 1
    # an indefinite Hessian without regularisation
    Hn = H.copy(); Hn[3, 3] = -1.0
    st, _, _ = big_host_solve(Hn, g, None, None, None, l, u, 0.0)
    assert st == 3
    # the iteration cap
    args = _one(random_qp(rng, 1, n, 60, 0), 0)
    st, _, it = big_host_solve(*args, 1e-9, max_iter=2)
    assert st == 2 and it == 3


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_inverse_dynamics_levels_wider_than_64_host(seed, oracle):
    """the reference's iHQP over its plugin, level by level, on a floating-base inverse-dynamics stack of 66 .. 88 variables"""
    rng = np.random.default_rng(100 + seed)
    nv = int(rng.integers(50, 64)); ncon = int(rng.integers(4, 9))
    n, level = id_like_levels(rng, nv, ncon, tau_max=float(rng.choice([40.0, 80.0, 200.0])))
    assert 64 < n <= 128
    eps = 2.221e-7
    xs = []
    for k in range(2):
        q = level(k, xs)
        st, x, _ = big_host_solve(*q, eps)
        ok, xo, _ = oracle.backend_solve(*q, eps)
        assert st == 0 and ok
        assert np.abs(x - xo).max() < 1e-6
        rq = ref_qpoases_solve(*q, eps / 2.221e-13)              # the reference's own qpOASES 3.1 (oracle/_ref), where it is present
        if rq is not None:
            assert rq[0] and np.abs(x - rq[1]).max() < 1e-6      # north_star's tolerance against the reference's own solver
        xs.append(x)


# ---------------------------------------------------------------------------------------------------------------------------------
def _solve_batch_gpu(args, eps, B, n, nc, max_iter=0):
    import torch
    dev = torch.device("cuda", 0)
    t = lambda a: None if a is None else torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev).contiguous()
    ts = [t(a) for a in args]
    x = torch.zeros((B, n), dtype=torch.float64, device=dev)
    st = torch.full((B,), -1, dtype=torch.int32, device=dev)
    it = torch.zeros((B,), dtype=torch.int32, device=dev)
    p = lambda a: None if a is None else C.c_void_p(a.data_ptr())
    rc = abi.lib().osot_qp_solve_batch(B, n, nc, *[p(a) for a in ts], eps, max_iter, p(x), p(st), p(it),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == abi.OK, abi.lib().osot_last_error()
    torch.cuda.synchronize()
    return x.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("n,nc,n_eq,B", [(65, 10, 2, 40), (70, 30, 6, 33), (96, 60, 20, 17), (128, 80, 40, 9), (66, 4, 1, 600)])
def test_qp_solve_batch_wider_than_64_gpu(n, nc, n_eq, B, oracle, gpu_device):
    """B generic QPs of 65 .. 128 variables through osot_qp_solve_batch (B = 600: more QPs than workgroups in the launch)"""
    rng = np.random.default_rng(n + nc)
    args = random_qp(rng, B, n, nc, n_eq)
    x, st, it = _solve_batch_gpu(args, 1e-9, B, n, nc)
    assert (st == 0).all()
    for i in range(0, B, max(1, B // 8)):
        q = _one(args, i)
        ok, xo, _ = oracle.backend_solve(*q, 1e-9)
        assert ok and np.abs(x[i] - xo).max() < 1e-8 * max(1.0, np.abs(xo).max())
        sth, xh, ith = big_host_solve(*q, 1e-9)          # the same source, one thread on the host: same path, same iteration count
        assert sth == 0 and np.abs(x[i] - xh).max() < 1e-9 * max(1.0, np.abs(xh).max()) and ith == it[i]
    for i in range(B):
        assert kkt_check(*_one(args, i), x[i], 1e-9) < 1e-6


@pytest.mark.gpu
def test_statuses_wider_than_64_gpu(gpu_device):
    n = 80
    rng = np.random.default_rng(2)
    M = rng.normal(size=(n + 3, n)); H = (M.T @ M)[None]; g = rng.normal(size=(1, n))
    a = rng.normal(size=n)
    A = np.vstack([a, a])[None]
    x, st, _ = _solve_batch_gpu((H, g, A, np.array([[1.0, 2.0]]), np.array([[1.0, 2.0]]), None, None), 1e-9, 1, n, 2)
    assert st[0] == 1 and not x.any()
    Hn = H.copy(); Hn[0, 3, 3] = -1.0
    x, st, _ = _solve_batch_gpu((Hn, g, None, None, None, None, None), 0.0, 1, n, 0)
    assert st[0] == 3
    args = random_qp(rng, 1, n, 60, 0)
    x, st, it = _solve_batch_gpu(args, 1e-9, 1, n, 60, max_iter=2)
    assert st[0] == 2 and it[0] == 3
    # beyond the surface's limit: refused with the reason, nothing launched
    rc = abi.lib().osot_qp_solve_batch(1, 129, 0, None, None, None, None, None, None, None, 1e-9, 0, None, None, None, None)
    assert rc != abi.OK


@pytest.mark.gpu
def test_plugin_surface_with_70_and_with_88_variables_gpu(oracle, gpu_device):
    """the reference's loop over levels (iHQP.cpp:263-358) through the BackEnd surface: initProblem / solve / updateTask +
    updateConstraints between the levels, on inverse-dynamics stacks wider than a wavefront"""
    from opensot_amd.solver import BackEnd
    for seed, nv, ncon in ((11, 55, 5), (12, 61, 9)):
        rng = np.random.default_rng(seed)
        n, level = id_like_levels(rng, nv, ncon)
        assert n in (70, 88)
        H, g, A, lA, uA, l, u = level(0, [])
        qp = BackEnd(n, A.shape[0], abi.HST_SEMIDEF, 1e6)       # eps = 1e3 * 2.221e-16 * 1e6 (QPOasesBackEnd.cpp:57, 67)
        eps = qp.getEpsRegularisation() if hasattr(qp, "getEpsRegularisation") else 1.0e3 * 2.221e-16 * 1e6
        assert qp.initProblem(H, g, A, lA, uA, l, u)
        x0 = qp.getSolution().copy()
        ok, xo, _ = oracle.backend_solve(H, g, A, lA, uA, l, u, eps)
        assert ok and np.abs(x0 - xo).max() < 1e-6
        H1, g1, A1, lA1, uA1, _, _ = level(1, [x0])
        assert qp.updateTask(H1, g1) and qp.updateConstraints(A1, lA1, uA1) and qp.solve()
        x1 = qp.getSolution().copy()
        ok, xo, _ = oracle.backend_solve(H1, g1, A1, lA1, uA1, l, u, eps)
        assert ok and np.abs(x1 - xo).max() < 1e-6
        rq = ref_qpoases_solve(H1, g1, A1, lA1, uA1, l, u, eps / 2.221e-13)
        if rq is not None:
            assert rq[0] and np.abs(x1 - rq[1]).max() < 1e-6
        # the level-0 optimum is kept by level 1 (the hierarchy), and a second solve() of the same problem repeats the answer
        A0 = A1[A.shape[0]:]
        assert np.abs(A0 @ x1 - A0 @ x0).max() < 1e-8
        assert qp.solve() and np.abs(qp.getSolution() - x1).max() < 1e-12


# ---------------------------------------------------------------------------------------------------------------------------------
# the committed golden vectors: answers of the reference's own qpOASES 3.1 (tests/golden/make_wide_golden.py)
def _golden():
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wide_id_levels.npz"))
    return z, len(z["cases"]), float(z["eps_factor"])


def _golden_level(z, ci, k):
    return tuple(z[f"c{ci}_k{k}_{nm}"] for nm in ("H", "g", "A", "lA", "uA", "l", "u"))


def test_golden_qpoases_wider_than_64_host():
    """strict: north_star's 1e-6 against the reference's qpOASES at OpenSoT's options, 1e-7 against its exact optimum"""
    z, ncases, factor = _golden()
    eps = 2.221e-13 * factor
    for ci in range(ncases):
        for k in range(2):
            q = _golden_level(z, ci, k)
            assert q[0].shape[0] > 64
            st, x, _ = big_host_solve(*q, eps)
            assert st == 0
            assert np.abs(x - z[f"c{ci}_k{k}_x_qpoases"]).max() < 1e-6
            assert np.abs(x - z[f"c{ci}_k{k}_x_qpoases_exact"]).max() < 1e-7


@pytest.mark.gpu
def test_golden_qpoases_wider_than_64_through_the_plugin_gpu(gpu_device):
    """the same vectors through osot_backend_* (initProblem for level 0; updateTask + updateConstraints + solve for level 1, posed on
    the reference's level-0 answer as the fixture was) and, as one batch, through osot_qp_solve_batch"""
    from opensot_amd.solver import BackEnd
    z, ncases, factor = _golden()
    for ci in range(ncases):
        H, g, A, lA, uA, l, u = _golden_level(z, ci, 0)
        qp = BackEnd(H.shape[0], A.shape[0], abi.HST_SEMIDEF, factor)
        assert qp.initProblem(H, g, A, lA, uA, l, u)
        assert np.abs(qp.getSolution() - z[f"c{ci}_k0_x_qpoases"]).max() < 1e-6
        H1, g1, A1, lA1, uA1, _, _ = _golden_level(z, ci, 1)
        assert qp.updateTask(H1, g1) and qp.updateConstraints(A1, lA1, uA1) and qp.solve()
        x1 = qp.getSolution()
        assert np.abs(x1 - z[f"c{ci}_k1_x_qpoases"]).max() < 1e-6
        assert np.abs(x1 - z[f"c{ci}_k1_x_qpoases_exact"]).max() < 1e-7
    same = [ci for ci in range(ncases) if _golden_level(z, ci, 1)[2].shape == _golden_level(z, 0, 1)[2].shape]
    args = tuple(np.stack([_golden_level(z, ci, 1)[j] for ci in same]) for j in range(7))
    n, nc = args[0].shape[1], args[2].shape[1]
    x, st, _ = _solve_batch_gpu(args, 2.221e-13 * factor, len(same), n, nc)
    assert (st == 0).all()
    for b, ci in enumerate(same):
        assert np.abs(x[b] - z[f"c{ci}_k1_x_qpoases"]).max() < 1e-6
