"""The reference's robot-free known-answer tests, restated, against (a) the oracle's restated Goldfarb-Idnani
back-end in both problem forms and (b) the reference's real qpOASES (oracle/_ref) when it is built.
These pin the oracle.  No GPU."""
import numpy as np
import pytest

from oracle import pyref

EPS_BASE = 1e3 * 2.221e-16


def _backends(oracle):
    """callables solve(H, g, A, lA, uA, l, u, eps_factor) -> (ok, x)"""
    out = []
    for form in (oracle.BE_EIQP_REFFORM, oracle.BE_EIQP_EQ):
        def f(H, g, A, lA, uA, l, u, eps_factor, form=form):
            ok, x, _ = oracle.backend_solve(H, g, A, lA, uA, l, u, EPS_BASE * eps_factor, form)
            return ok, x
        out.append((f"oracle-form{form}", f))
    if oracle.ref_available():
        def q(H, g, A, lA, uA, l, u, eps_factor):
            nc = 0 if A is None else A.shape[0]
            be = pyref.RefBackEnd(len(g), nc, pyref.HST_SEMIDEF, eps_factor)
            ok = be.initProblem(H, g, A, lA, uA, l, u)
            return ok, be.getSolution()
        out.append(("qpOASES-ref", q))
    return out


def test_update_constraint(oracle):
    """tests/solvers/TestQPOases.cpp:208-254"""
    for name, solve in _backends(oracle):
        Hm = np.array([[1.0, 1, 1]]); b = np.array([10.0])
        l = -10 * np.ones(3); u = 10 * np.ones(3)
        ok, x = solve(Hm.T @ Hm, -Hm.T @ b, np.zeros((1, 3)), np.zeros(1), np.zeros(1), l, u, 1e4)
        assert ok, name
        np.testing.assert_allclose(x, [3.333, 3.333, 3.333], atol=1e-3, err_msg=name)
        ok, x = solve(Hm.T @ Hm, -Hm.T @ b, np.array([[1.0, 0, 1]]), np.array([20.0]), np.array([20.0]), l, u, 1e4)
        assert ok, name
        np.testing.assert_allclose(x, [10, -10, 10], atol=1e-6, err_msg=name)


def test_update_task(oracle):
    """tests/solvers/TestQPOases.cpp:274-340"""
    l = -10 * np.ones(3); u = 10 * np.ones(3)
    cases = [(np.array([[1.0, 1, 1], [0, 1, 1]]), np.array([6.0, 5]), [1, 2.5, 2.5], 1e-6),
             (np.array([[1.0, 1, 1], [0, 1, 1], [1, 1, 0]]), np.array([6.0, 5, 3]), [1, 2, 3], 1e-6),
             (np.array([[1.0, 1, 1], [0, 1, 1], [1, 1, 0], [1, 0, 1]]), np.array([6.0, 5, 3, 3]),
              [.5714, 2.5714, 2.5714], 1e-4)]
    for name, solve in _backends(oracle):
        for Hm, b, want, tol in cases:
            ok, x = solve(Hm.T @ Hm, -Hm.T @ b, None, None, None, l, u, 1.0)
            assert ok, name
            np.testing.assert_allclose(x, want, atol=tol, err_msg=name)


def test_simple_problem(oracle):
    """tests/solvers/TestQPOases.cpp:83-116, 346-412: H = I, g = (-5, 5), zero rows in [-10, 10] -> x = -g"""
    H = np.eye(2); A = np.zeros((2, 2))
    lA = -10 * np.ones(2); uA = 10 * np.ones(2); l = -10 * np.ones(2); u = 10 * np.ones(2)
    for name, solve in _backends(oracle):
        for g in (np.array([-5.0, 5.0]), np.array([-1.0, 1.0])):
            ok, x = solve(H, g, A, lA, uA, l, u, 1e-9)
            assert ok, name
            np.testing.assert_allclose(x, -g, atol=1e-14, err_msg=name)


def test_eps_regularisation_value(oracle):
    """tests/solvers/TestQPOases.cpp:798-836: factor 1 -> 2.221e-13"""
    if not oracle.ref_available():
        pytest.skip("oracle/_ref not built")
    be = pyref.RefBackEnd(2, 0, pyref.HST_SEMIDEF, 1.0)
    assert be.eps_abs == pytest.approx(2.221e-13, rel=1e-12)


def test_generic_task_sum_to_one(oracle):
    """tests/tasks/TestGenericTask.cpp:175-203: minimise ||x|| subject to x0 + x1 = 1 inside a box"""
    n = 2
    H = np.eye(n); g = np.zeros(n)
    A = np.ones((1, n)); lA = np.array([1.0]); uA = np.array([1.0])
    l = -np.ones(n); u = np.ones(n)
    for name, solve in _backends(oracle):
        ok, x = solve(H, g, A, lA, uA, l, u, 1.0)
        assert ok, name
        assert abs(x.sum() - 1.0) < 1e-9, name


def test_ihqp_cost_function_and_regularisation(oracle):
    """tests/solvers/TestiHQP.cpp:70-143: H == A'A exactly, g == -A'b exactly, x = -H^-1 g @1e-12,
    user regularisation task doubles H."""
    n = 7
    I = np.eye(n); b = np.ones(n)
    asm = {"n": n, "B": 1, "L": 1, "eps_abs": EPS_BASE * 2e2, "m": [n], "ma": [n], "A": [I[None]],
           "b": [b[None]], "w": [None], "c": [None], "nc": 0, "C": None, "lo": None, "up": None, "l": None, "u": None}
    H, g = oracle.cost_function(asm, 0, 0)
    assert (H == I.T @ I).all() and (g == -(I.T @ b)).all()
    r = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
    np.testing.assert_allclose(r["dq"][0], b, atol=1e-12)
    # regularisation task (TestiHQP.cpp:112-139; iHQP.cpp:265-266, 274-278): H += Hr, g += gr with Hr = I'I, gr = -I' b_r
    rng = np.random.default_rng(0)
    br = -(1.0 / 0.001) * 1e-4 * rng.uniform(-1, 1, n)
    asm["reg"] = {"A": I[None], "b": br[None], "w": 1.0}
    Hreg, greg = oracle.cost_function(asm, 0, 0, regularised=True)
    assert (Hreg == 2 * I.T @ I).all() and (greg == -(I.T @ b + I.T @ br)).all()      # TestiHQP.cpp:130-131
    r = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
    np.testing.assert_allclose(r["dq"][0], -np.linalg.solve(Hreg, greg), atol=1e-12)   # TestiHQP.cpp:134-138
    asm["reg"] = {"A": None, "b": br[None], "w": 1.0}      # the implicit [I 0] form the product supports
    r2 = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
    assert (r2["dq"] == r["dq"]).all()
    if oracle.ref_available():
        rq = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1)
        np.testing.assert_allclose(rq["dq"][0], r["dq"][0], atol=1e-9)


def test_cross_backend_parity(oracle):
    """tests/solvers/TesteiQuadProg.cpp:72-130 (same pattern in TestOSQP.cpp:79-129): identical
    H,g,A,lA,uA,l,u into two back-ends created with eps factor 0, ||x_a - x_b|| <= 1e-12 after initProblem
    and after repeated solve(); the eiQuadProg side must return -g exactly."""
    if not oracle.ref_available():
        pytest.skip("oracle/_ref not built")
    H = np.eye(2); g = np.array([-5.0, 5.0]); A = np.zeros((2, 2))
    lA = -10 * np.ones(2); uA = 10 * np.ones(2); l = -10 * np.ones(2); u = 10 * np.ones(2)
    be = pyref.RefBackEnd(2, 2, pyref.HST_IDENTITY, 0.0)
    assert be.initProblem(H, g, A, lA, uA, l, u)
    ok, x, _ = oracle.backend_solve(H, g, A, lA, uA, l, u, 0.0, oracle.BE_EIQP_REFFORM)
    assert ok and np.linalg.norm(x - be.getSolution()) <= 1e-12
    assert (x == -g).all()
    for _ in range(10):
        assert be.solve()
        assert np.linalg.norm(x - be.getSolution()) <= 1e-12
