"""Second back-end and tensor entry points.  The OSQP-convention ADMM kernel restates the published
algorithm (osqp is not vendored in the reference: PARITY UNPINNED against osqp itself) and is checked against the
active-set kernel / the oracle at 1e-4 -- the accuracy OSQP's own tolerances (eps_abs = eps_rel = 1e-5,
OSQPBackEnd.cpp:38-39) buy."""
import numpy as np
import pytest

from helpers import emu_qp, emu_qp_admm, random_qp


@pytest.mark.parametrize("n,nc,n_eq", [(5, 3, 1), (17, 9, 0), (32, 20, 3), (50, 40, 6), (64, 30, 4)])
def test_admm_matches_active_set_on_random_qps(n, nc, n_eq, oracle):
    rng = np.random.default_rng(100 + n)
    B = 4
    H, g, A, lA, uA, l, u = random_qp(rng, B, n, nc, n_eq)
    xa, sa, _ = emu_qp(H, g, A, lA, uA, l, u, eps_abs=1e-9)
    x, st, it = emu_qp_admm(H, g, A, lA, uA, l, u, eps_reg=1e-9)
    assert (sa == 0).all() and (st == 0).all() and (it <= 4000).all()
    assert np.abs(x - xa).max() < 1e-4
    for i in range(B):
        ok, xo, _ = oracle.backend_solve(H[i], g[i], A[i], lA[i], uA[i], l[i], u[i], 1e-9)
        assert ok and np.abs(x[i] - xo).max() < 1e-4


def test_admm_known_answers_and_infeasibility():
    """TestQPOases.cpp:208-254 ((10, -10, 10)) through the ADMM kernel; no box; an infeasible problem is not reported solved"""
    l = -10 * np.ones((1, 3)); u = 10 * np.ones((1, 3))
    Hm = np.array([[1.0, 1, 1]]); b = np.array([10.0])
    x, st, _ = emu_qp_admm((Hm.T @ Hm)[None], (-Hm.T @ b)[None], np.array([[[1.0, 0, 1]]]), np.array([[20.0]]), np.array([[20.0]]),
                           l, u, eps_reg=2.22e-13 * 1e4)
    assert st[0] == 0
    np.testing.assert_allclose(x[0], [10, -10, 10], atol=1e-3)
    H = np.eye(4)[None] * 2.0; g = np.array([[1.0, -2, 3, -4]])
    x, st, _ = emu_qp_admm(H, g, None, None, None, None, None)
    assert st[0] == 0
    np.testing.assert_allclose(x[0], -g[0] / 2.0, atol=1e-5)
    H = np.eye(2)[None]; g = np.zeros((1, 2))
    A = np.array([[[1.0, 1.0]]]); lA = np.array([[5.0]]); uA = np.array([[np.inf]])
    x, st, _ = emu_qp_admm(H, g, A, lA, uA, -np.ones((1, 2)), np.ones((1, 2)))
    assert st[0] != 0 and (x[0] == 0).all()


def test_admm_ruiz_equilibration_on_badly_scaled_problems():
    """osqp's scaling = 10 (Ruiz, then the cost factor): variables in different units (columns of H and A scaled by
    1e-2 .. 1e2).  The residual test is on the unscaled residuals either way, so both runs stop at the same accuracy of the
    KKT conditions -- which on such a problem pins the objective (checked against the active-set kernel's optimum at 1e-3
    relative) far better than x itself; the equilibrated iteration gets there in a few hundred steps, the plain one needs
    thousands (this seed: 100 .. 425 against 225 .. 20000)"""
    rng = np.random.default_rng(77)
    B, n, nc = 6, 24, 12
    H, g, A, lA, uA, l, u = random_qp(rng, B, n, nc, 2)
    sc = 10.0 ** rng.uniform(-2, 2, size=(B, n))
    H = H * sc[:, :, None] * sc[:, None, :]; g = g * sc; A = A * sc[:, None, :]; l = l / sc; u = u / sc
    f = lambda x: 0.5 * np.einsum("bi,bij,bj->b", x, H, x) + (g * x).sum(1)
    xa, sa, _ = emu_qp(H, g, A, lA, uA, l, u, eps_abs=1e-12)
    x1, s1, it1 = emu_qp_admm(H, g, A, lA, uA, l, u, eps_reg=1e-12, scaling=0)
    x0, s0, it0 = emu_qp_admm(H, g, A, lA, uA, l, u, eps_reg=1e-12, scaling=-1, max_iter=20000)
    assert (sa == 0).all() and (s1 == 0).all()
    assert (np.abs(f(x1) - f(xa)) < 1e-3 * np.abs(f(xa))).all()
    assert np.abs((x1 - xa) * sc).max() < 1e-2                     # in the variables' own units
    Ax = np.einsum("brn,bn->br", A, x1)
    assert (Ax >= lA - 1e-3).all() and (Ax <= uA + 1e-3).all() and (x1 * sc >= l * sc - 1e-3).all() and (x1 * sc <= u * sc + 1e-3).all()
    print(f"iterations with / without Ruiz: {it1.tolist()} / {it0.tolist()}")
    assert (it1 <= 1000).all() and 5 * it1.sum() < it0.sum()


def test_admm_warm_start_across_control_cycles():
    """OSQPBackEnd keeps its workspace: the next cycle's solve starts from the previous x, y and rho (OSQPBackEnd.cpp:120-143,
    268-287).  The same problem again converges at the first residual test; a drifted problem takes fewer iterations warm than
    cold and lands on the same answer; an instance without state (rho = 0) runs cold"""
    rng = np.random.default_rng(78)
    B, n, nc = 5, 20, 14
    H, g, A, lA, uA, l, u = random_qp(rng, B, n, nc, 2)
    warm = {"x": np.zeros((B, n)), "y": np.zeros((B, nc + n)), "rho": np.zeros(B)}
    x0, s0, it0 = emu_qp_admm(H, g, A, lA, uA, l, u, eps_reg=1e-9, warm=warm)
    xc, sc_, itc = emu_qp_admm(H, g, A, lA, uA, l, u, eps_reg=1e-9)
    assert (s0 == 0).all() and np.array_equal(it0, itc) and np.array_equal(x0, xc)   # no state yet: the cold solve, bit for bit
    assert (warm["rho"] > 0).all() and np.abs(warm["x"] - x0).max() == 0.0
    x1, s1, it1 = emu_qp_admm(H, g, A, lA, uA, l, u, eps_reg=1e-9, warm=warm)
    assert (s1 == 0).all() and (it1 == 25).all() and np.abs(x1 - x0).max() < 1e-5
    g2 = g + 0.02 * rng.normal(size=g.shape) * np.abs(g).max()
    warm["rho"][3] = 0.0                                                                 # one instance forgets its state
    xw, sw, itw = emu_qp_admm(H, g2, A, lA, uA, l, u, eps_reg=1e-9, warm=warm)
    xk, sk, itk = emu_qp_admm(H, g2, A, lA, uA, l, u, eps_reg=1e-9)
    xa, sa, _ = emu_qp(H, g2, A, lA, uA, l, u, eps_abs=1e-9)
    assert (sw == 0).all() and (sk == 0).all() and (sa == 0).all()
    assert np.abs(xw - xa).max() < 1e-4 and np.abs(xk - xa).max() < 1e-4
    print(f"iterations warm / cold on the drifted problem: {itw.tolist()} / {itk.tolist()}")
    assert itw[3] == itk[3] and np.array_equal(xw[3], xk[3])
    others = np.arange(B) != 3
    assert itw[others].sum() < itk[others].sum()


@pytest.mark.gpu
def test_torch_qp_solve_both_back_ends_gpu(oracle, gpu_device):
    """tensors in, tensors out: the caller's device pointers go straight to the C-ABI; the two back-ends agree at 1e-4, the
    active-set one with the oracle at 1e-8"""
    import torch
    from opensot_amd import torch_api as ta
    rng = np.random.default_rng(5)
    B, n, nc = 256, 32, 24
    H, g, A, lA, uA, l, u = random_qp(rng, B, n, nc, 4)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev).contiguous()
    args = [t(a) for a in (H, g, A, lA, uA, l, u)]
    xa, sa, ia = ta.qp_solve(*args, eps_regularisation=1.0, be_solver=ta.solver_back_ends.qpOASES)
    xo, so, io = ta.qp_solve(*args, eps_regularisation=1.0, be_solver=ta.solver_back_ends.OSQP)
    torch.cuda.synchronize()
    assert xa.is_cuda and xa.shape == (B, n) and (sa == 0).all() and (so == 0).all()
    assert float((xa - xo).abs().max()) < 1e-4
    for i in range(0, B, 32):
        ok, xr, _ = oracle.backend_solve(H[i], g[i], A[i], lA[i], uA[i], l[i], u[i], 1e3 * 2.221e-16)
        assert ok and np.abs(xa[i].cpu().numpy() - xr).max() < 1e-8
    with pytest.raises(ValueError):
        ta.qp_solve(args[0], args[1][:, :5])
    with pytest.raises(TypeError):
        ta.qp_solve(args[0].cpu(), args[1])


@pytest.mark.gpu
def test_torch_ihqp_and_nhqp_classes_gpu(oracle, gpu_device):
    """pyopensot-shaped front-ends: iHQP(...).solve() -> dq tensor, setActiveStack; nHQP with its option setters"""
    import torch
    from opensot_amd import synth, torch_api as ta
    B = 128
    plan, leaf = synth.make_velocity_stack("C3", B, seed=4)
    s = ta.iHQP(plan, B, eps_regularisation=1e6)
    assert s.getNumberOfTasks() == 3
    dev_leaf = s.stack.load_leaf(leaf)
    dq = s.solve(dev_leaf)
    torch.cuda.synchronize()
    asm = oracle.assemble(plan, leaf)
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=0)
    assert isinstance(dq, torch.Tensor) and dq.is_cuda and (s.status() == 0).all()
    assert np.abs(dq.cpu().numpy() - ref["dq"]).max() < 1e-9
    s.setActiveStack(1, False)
    dq2 = s.solve(dev_leaf).clone(); torch.cuda.synchronize()
    ref2 = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=0, active=(1, 0, 1))
    assert np.abs(dq2.cpu().numpy() - ref2["dq"]).max() < 1e-9
    s.activateAllStacks()
    with pytest.raises(RuntimeError):
        ta.iHQP(plan, B, be_solver=ta.solver_back_ends.OSQP)
    nh = ta.nHQP(plan, B, eps_regularisation=1e6)
    nh.setPerformAbRegularization(False); nh.setPerformSelectiveNullSpaceRegularization(False)
    dqn = nh.solve(nh.stack.load_leaf(leaf)); torch.cuda.synchronize()
    ok = (nh.status() == 0).cpu().numpy()
    assert ok.mean() > 0.95 and np.abs(dqn.cpu().numpy()[ok] - ref["dq"][ok]).max() < 1e-6


@pytest.mark.gpu
def test_admm_cross_checks_config5_qps_gpu(gpu_device):
    """the use SURVEY 8f-4 names: an independent check of config 5's 102-row QPs -- level 0 of the inverse-dynamics stack as
    one explicit QP per instance through both back-ends"""
    import torch
    from opensot_amd import synth, torch_api as ta
    from oracle import pyoracle as po
    B = 64
    plan, leaf = synth.make_id_stack(B, seed=3)
    asm = po.assemble(plan, leaf)
    n = plan.n
    H = np.zeros((B, n, n)); g = np.zeros((B, n))
    for i in range(B):
        H[i], g[i] = po.cost_function(asm, i, 0)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev).contiguous()
    lo = np.clip(np.nan_to_num(asm["lo"], neginf=-1e20, posinf=1e20), -1e20, 1e20); up = np.clip(np.nan_to_num(asm["up"], neginf=-1e20, posinf=1e20), -1e20, 1e20)
    args = [t(H), t(g), t(asm["C"]), t(lo), t(up)]
    xa, sa, _ = ta.qp_solve(*args, eps_regularisation=1e6, be_solver=ta.solver_back_ends.qpOASES)
    xo, so, io = ta.qp_solve(*args, eps_regularisation=1e6 * 1e3 * 2.221e-16 / 2.22e-13, be_solver=ta.solver_back_ends.OSQP)
    torch.cuda.synchronize()
    assert (sa == 0).all() and (so == 0).float().mean() > 0.9
    ok = (so == 0).cpu().numpy()
    # H has rank 15 of 50 here (+ eps 2.2e-7): along its null directions x is decided by eps |x|^2 alone, far below what
    # eps_abs = eps_rel = 1e-5 resolves -- the two solvers are compared on what the QP is about: the objective value and
    # the constraints
    Xa, Xo = xa.cpu().numpy(), xo.cpu().numpy()
    f = lambda X: 0.5 * np.einsum("bi,bij,bj->b", X, H, X) + np.einsum("bi,bi->b", g, X)
    fa, fo = f(Xa), f(Xo)
    assert (np.abs(fa - fo)[ok] <= 1e-3 * (1.0 + np.abs(fa[ok]))).all()
    ax = np.einsum("brj,bj->br", asm["C"], Xo)
    viol = np.maximum(np.where(lo > -1e20, lo - ax, 0.0), np.where(up < 1e20, ax - up, 0.0)).max(axis=1)
    assert viol[ok].max() < 1e-2      # rows of norm ~30 and bounds of 30 .. 1e3: 1e-5-class relative residuals


@pytest.mark.gpu
def test_admm_warm_start_and_scaling_gpu(gpu_device):
    """osot_qp_solve_batch_admm_warm on the device: the state tensors carry x, y, rho across drifting cycles (fewer iterations
    than cold solves of the same problems, same answers); without state the call is the cold one"""
    import torch
    from opensot_amd import torch_api as ta
    rng = np.random.default_rng(91)
    B, n, nc = 256, 32, 20
    H, g, A, lA, uA, l, u = random_qp(rng, B, n, nc, 2)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev).contiguous()
    tH, tA, tlA, tuA, tl, tu = t(H), t(A), t(lA), t(uA), t(l), t(u)
    warm = ta.admm_state(B, n, nc, box=True, device=0)
    tot_w = tot_c = 0
    for cyc in range(4):
        gk = t(g + 0.01 * cyc * np.abs(g).max() * rng.normal(size=g.shape))
        xw, sw, iw = ta.qp_solve(tH, gk, tA, tlA, tuA, tl, tu, eps_regularisation=1e4, be_solver=ta.solver_back_ends.OSQP, warm=warm)
        xc, sc_, ic = ta.qp_solve(tH, gk, tA, tlA, tuA, tl, tu, eps_regularisation=1e4, be_solver=ta.solver_back_ends.OSQP)
        xa, sa, _ = ta.qp_solve(tH, gk, tA, tlA, tuA, tl, tu, eps_regularisation=1e4)
        torch.cuda.synchronize()
        assert (sw == 0).all() and (sc_ == 0).all() and (sa == 0).all()
        assert (xw - xa).abs().max().item() < 1e-4 and (xc - xa).abs().max().item() < 1e-4
        if cyc == 0:
            assert torch.equal(xw, xc) and torch.equal(iw, ic)
        else:
            tot_w += int(iw.sum()); tot_c += int(ic.sum())
    print(f"ADMM iterations over 3 drifting cycles x {B} instances: warm {tot_w}, cold {tot_c}")
    assert tot_w < 0.8 * tot_c
    assert (warm["rho"] > 0).all()
