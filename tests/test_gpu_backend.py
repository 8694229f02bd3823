"""The BackEnd plugin surface (batch-of-one, host pointers) on the GPU: the reference's robot-free known-answer
tests restated with the same method names (tests/solvers/TestQPOases.cpp), plus generic batched QPs."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import kkt_check, random_qp
from opensot_amd import abi
from opensot_amd.solver import BackEnd

pytestmark = pytest.mark.gpu


def test_update_constraint(gpu_device):
    """TestQPOases.cpp:208-254"""
    Hm = np.array([[1.0, 1, 1]]); b = np.array([10.0])
    A = np.zeros((1, 3)); lA = np.zeros(1); uA = np.zeros(1)
    l = -10 * np.ones(3); u = 10 * np.ones(3)
    qp = BackEnd(3, 1, abi.HST_SEMIDEF, 1e4)
    assert qp.initProblem(Hm.T @ Hm, -Hm.T @ b, A, lA, uA, l, u)
    assert qp.solve()
    np.testing.assert_allclose(qp.getSolution(), [3.333, 3.333, 3.333], atol=1e-3)
    assert qp.updateConstraints(np.array([[1.0, 0, 1]]), np.array([20.0]), np.array([20.0]))
    assert qp.solve()
    np.testing.assert_allclose(qp.getSolution(), [10, -10, 10], atol=1e-6)


def test_update_task(gpu_device):
    """TestQPOases.cpp:274-340"""
    qp = BackEnd(3, 0, abi.HST_UNKNOWN, 1.0)
    l = -10 * np.ones(3); u = 10 * np.ones(3)
    Hm = np.array([[1.0, 1, 1], [0, 1, 1]]); b = np.array([6.0, 5])
    assert qp.initProblem(Hm.T @ Hm, -Hm.T @ b, np.zeros((0, 0)), np.zeros(0), np.zeros(0), l, u)
    assert qp.solve()
    np.testing.assert_allclose(qp.getSolution(), [1, 2.5, 2.5], atol=1e-6)
    Hm = np.array([[1.0, 1, 1], [0, 1, 1], [1, 1, 0]]); b = np.array([6.0, 5, 3])
    assert qp.updateTask(Hm.T @ Hm, -Hm.T @ b) and qp.solve()
    np.testing.assert_allclose(qp.getSolution(), [1, 2, 3], atol=1e-6)
    Hm = np.array([[1.0, 1, 1], [0, 1, 1], [1, 1, 0], [1, 0, 1]]); b = np.array([6.0, 5, 3, 3])
    assert qp.updateTask(Hm.T @ Hm, -Hm.T @ b) and qp.solve()
    np.testing.assert_allclose(qp.getSolution(), [.5714, 2.5714, 2.5714], atol=1e-4)
    # size mismatch is refused like BackEnd::updateTask (BackEnd.cpp:23-41)
    assert not qp.updateTask(np.eye(4), np.zeros(4))


def test_simple_and_updated_problem(gpu_device):
    """TestQPOases.cpp:346-412: H = I, g = (-5, 5) -> x = -g, 1000 repeated solves, then an updated g"""
    H = np.eye(2); g = np.array([-5.0, 5.0]); A = np.zeros((2, 2))
    lA = -10 * np.ones(2); uA = 10 * np.ones(2); l = -10 * np.ones(2); u = 10 * np.ones(2)
    qp = BackEnd(2, 2, abi.HST_IDENTITY, 1e-9)
    assert qp.initProblem(H, g, A, lA, uA, l, u)
    for _ in range(50):
        assert qp.solve()
        np.testing.assert_allclose(qp.getSolution(), -g, atol=1e-14)
    g2 = np.array([-1.0, 1.0])
    assert qp.updateTask(H, g2) and qp.solve()
    np.testing.assert_allclose(qp.getSolution(), -g2, atol=1e-14)
    assert qp.getObjective() == pytest.approx(-1.0, abs=1e-12)


def test_infeasible_returns_false(gpu_device):
    qp = BackEnd(2, 1, abi.HST_IDENTITY, 1.0)
    ok = qp.initProblem(np.eye(2), np.zeros(2), np.array([[1.0, 1.0]]), np.array([5.0]), np.array([np.inf]),
                        -np.ones(2), np.ones(2))
    assert not ok


def test_generic_task_sum_to_one(gpu_device):
    """tests/tasks/TestGenericTask.cpp:175-203"""
    qp = BackEnd(2, 1, abi.HST_IDENTITY, 1.0)
    assert qp.initProblem(np.eye(2), np.zeros(2), np.ones((1, 2)), np.array([1.0]), np.array([1.0]),
                          -np.ones(2), np.ones(2))
    assert abs(qp.getSolution().sum() - 1.0) < 1e-9


@pytest.mark.parametrize("n,nc,n_eq,B", [(7, 5, 2, 33), (32, 24, 8, 257), (35, 12, 3, 48), (50, 60, 6, 64), (64, 16, 4, 31)])
def test_qp_solve_batch_random(n, nc, n_eq, B, oracle, gpu_device):
    """B generic QPs of one shape through osot_qp_solve_batch (incl. the 64-lane team path, n > 32)"""
    rng = np.random.default_rng(n + nc)
    H, g, A, lA, uA, l, u = random_qp(rng, B, n, nc, n_eq)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev).contiguous()
    tH, tg, tA, tlA, tuA, tl, tu = map(t, (H, g, A, lA, uA, l, u))
    x = torch.zeros((B, n), dtype=torch.float64, device=dev)
    st = torch.full((B,), -1, dtype=torch.int32, device=dev)
    it = torch.zeros((B,), dtype=torch.int32, device=dev)
    p = lambda a: C.c_void_p(a.data_ptr())
    rc = abi.lib().osot_qp_solve_batch(B, n, nc, p(tH), p(tg), p(tA), p(tlA), p(tuA), p(tl), p(tu), 1e-9, 0,
                                       p(x), p(st), p(it), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == abi.OK, abi.lib().osot_last_error()
    torch.cuda.synchronize()
    x = x.cpu().numpy()
    assert (st.cpu().numpy() == 0).all()
    for i in range(0, B, max(1, B // 16)):
        ok, xo, _ = oracle.backend_solve(H[i], g[i], A[i], lA[i], uA[i], l[i], u[i], 1e-9)
        assert ok and np.abs(x[i] - xo).max() < 1e-8
    for i in range(B):
        assert kkt_check(H[i], g[i], A[i], lA[i], uA[i], l[i], u[i], x[i], 1e-9) < 1e-6


def test_qp_solve_batch_argument_errors(gpu_device):
    L = abi.lib()
    z = C.c_void_p(0)
    assert L.osot_qp_solve_batch(1, 0, 0, z, z, z, z, z, z, z, 0.0, 0, z, z, z, z) == abi.ERR_INVALID
    assert L.osot_qp_solve_batch(1, 65, 0, z, z, z, z, z, z, z, 0.0, 0, z, z, z, z) == abi.ERR_INVALID
    assert L.osot_qp_solve_batch(0, 4, 0, z, z, z, z, z, z, z, 0.0, 0, z, z, z, z) == abi.OK   # empty batch
    assert L.osot_qp_solve_batch(1, 4, 0, z, z, z, z, z, z, z, 0.0, 0, z, z, z, z) == abi.ERR_INVALID


def test_allgather_world_of_one(gpu_device):
    """osot_comm_* / osot_allgather_dq with a single-rank RCCL communicator"""
    L = abi.lib()
    uid = (C.c_ubyte * 128)()
    assert L.osot_comm_unique_id(C.cast(uid, C.c_void_p)) == abi.OK
    h = C.c_void_p()
    assert L.osot_comm_create(C.cast(uid, C.c_void_p), 0, 1, 0, C.byref(h)) == abi.OK, L.osot_last_error()
    a = torch.arange(64, dtype=torch.float64, device="cuda:0")
    b = torch.zeros_like(a)
    rc = L.osot_allgather_dq(h, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), 64,
                             C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == abi.OK
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert L.osot_comm_destroy(h) == abi.OK


def test_randomised_qp_sweep(oracle, gpu_device):
    """40 random QP shapes (n = 2..64, up to 39 rows, equalities, one-sided rows, boxes, full-rank and rank-deficient
    Hessians with g in range(H), three eps values) x 64 instances through osot_qp_solve_batch: KKT on every solved
    instance, a sample against the oracle's single-QP solve, and every unsolved instance must be one the oracles
    cannot solve either (tests/stress_qp.py)"""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "stress_qp.py"), "5", "40"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "40 QP shapes x 64 instances" in out.stdout and ": 0 with a mismatch" in out.stdout, out.stdout[-2000:]


def test_backend_options(gpu_device):
    """BackEnd::getOptions / setOptions (BackEnd.h:139-145; the qpOASES back-end hands out its Options with nWSR,
    QPOasesBackEnd.cpp:30, 309-318): the iteration cap of the active-set loop is settable, the last solve's iteration count
    and status are readable; a cap below what the problem needs makes solve() return false like an exhausted nWSR does"""
    n = 6
    H = np.eye(n); g = -np.arange(1.0, n + 1.0)                 # unconstrained minimiser (1 .. 6): every upper bound 0.5 binds
    qp = BackEnd(n, 0, abi.HST_IDENTITY, 1.0)
    assert qp.initProblem(H, g, None, None, None, -np.ones(n), 0.5 * np.ones(n))
    np.testing.assert_allclose(qp.getSolution(), 0.5 * np.ones(n), atol=1e-12)
    o = qp.getOptions()
    assert o["max_iterations"] == 0 and o["last_iterations"] == n and o["last_status"] == 0
    assert qp.setOptions({"max_iterations": 3})
    assert not qp.solve() and qp.getOptions()["last_status"] == 2          # OSOT_STATUS_MAX_ITER
    assert qp.setOptions({"max_iterations": 0}) and qp.solve()
    assert not qp.setOptions({"max_iterations": -1})


@pytest.mark.parametrize("n,nc", [(24, 10), (36, 12), (50, 14)])     # osot_qp_kernel<32 / 40 / 56, HOT> (the 40- and 56-lane layouts: round 5)
def test_solve_hot_starts_from_the_previous_working_set(n, nc, gpu_device):
    """QPOasesBackEnd::solve hot-starts every call (QPOasesBackEnd.cpp:258-285): a repeated solve() of the same problem re-adds the
    previous working set without scans (never more iterations than the cold solve, the same x), a drifted g still lands on the
    cold answer, and initProblem / a changed row count forget the set (QPOasesBackEnd.cpp:229-244)"""
    rng = np.random.default_rng(31)
    H, g, A, lA, uA, l, u = [a[0] if a is not None else None for a in random_qp(rng, 1, n, nc, box=True, scale=1.0)]
    g = g * 4.0                                                   # push the minimiser well outside the box: many active bounds
    qp = BackEnd(n, nc, abi.HST_SEMIDEF, 1e6)
    assert qp.initProblem(H, g, A, lA, uA, l, u)
    x_cold, it_cold = qp.getSolution(), qp.getOptions()["last_iterations"]
    assert it_cold >= 4                                           # (the test needs a non-trivial active set)
    assert qp.solve()
    x_hot, it_hot = qp.getSolution(), qp.getOptions()["last_iterations"]
    np.testing.assert_allclose(x_hot, x_cold, atol=1e-10)
    # fewer iterations than the cold solve: the hot trips skip the scans, and what is re-added is (mostly) the final set.  (Not
    # necessarily exactly the size of the final set: the multipliers are looked at after every few hot additions, and a partial
    # working set can show a negative multiplier that the complete one does not -- such a member is taken out and found again.)
    print(f"[backend hot start] iterations cold {it_cold}, hot {it_hot}")
    assert it_hot < it_cold
    assert kkt_check(H, g, A, lA, uA, l, u, x_hot, qp.getEpsRegularisation()) < 1e-7
    g2 = g * (1.0 + 0.01 * rng.normal(size=n))                    # a drifting linear term: hot and cold agree
    assert qp.updateTask(H, g2) and qp.solve()
    x2 = qp.getSolution()
    cold = BackEnd(n, nc, abi.HST_SEMIDEF, 1e6)
    assert cold.initProblem(H, g2, A, lA, uA, l, u)
    np.testing.assert_allclose(x2, cold.getSolution(), atol=1e-10)
    # initProblem is a cold start again: the iteration count of the first solve comes back
    assert qp.initProblem(H, g, A, lA, uA, l, u)
    assert qp.getOptions()["last_iterations"] == it_cold
    # ... and so is a changed row count
    assert qp.updateConstraints(A[:6], lA[:6], uA[:6]) and qp.solve()
    c6 = BackEnd(n, 6, abi.HST_SEMIDEF, 1e6)
    assert c6.initProblem(H, g, A[:6], lA[:6], uA[:6], l, u)
    np.testing.assert_allclose(qp.getSolution(), c6.getSolution(), atol=1e-10)
    assert qp.getOptions()["last_iterations"] == c6.getOptions()["last_iterations"]
