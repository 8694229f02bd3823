"""Oracle restatement vs the committed golden vectors (generated from the reference's qpOASES, see
tests/golden/make_golden.py) and, when oracle/_ref is present, vs qpOASES live.  No GPU."""
import numpy as np
import pytest

from helpers import load_golden


@pytest.mark.parametrize("cfg", ["C2", "C3", "C4", "C5"])
def test_leaf_update_matches_golden(cfg, oracle):
    plan, leaf, z = load_golden(cfg)
    asm = oracle.assemble(plan, leaf)
    for k in range(plan.L):
        np.testing.assert_array_equal(asm["b"][k], z[f"asm_b{k}"])
        np.testing.assert_array_equal(asm["w"][k], z[f"asm_w{k}"])
    for name in ("l", "u", "C", "lo", "up"):
        if asm[name] is not None:
            np.testing.assert_array_equal(asm[name], z[f"asm_{name}"])


@pytest.mark.parametrize("cfg", ["C2", "C3", "C4", "C5"])
@pytest.mark.parametrize("backend", ["eq", "refform"])
def test_cascade_matches_qpoases_golden(cfg, backend, oracle):
    plan, leaf, z = load_golden(cfg)
    asm = oracle.assemble(plan, leaf)
    be = oracle.BE_EIQP_EQ if backend == "eq" else oracle.BE_EIQP_REFFORM
    r = oracle.ihqp_solve_batch(asm, be, nthreads=1)
    assert (r["status"] == 1).all()
    ok = z["ok_ref"].astype(bool)
    # north_star tolerance: 1e-6 on the solved joint velocities vs the reference qpOASES back-end
    assert np.abs(r["dq"][ok] - z["x_ref"][ok][:, -1]).max() < 1e-6
    okx = z["ok_exact"].astype(bool)
    # against qpOASES run to (almost) machine-precision termination the port agrees far tighter
    assert np.abs(r["dq"][okx] - z["x_exact"][okx][:, -1]).max() < 1e-8


def test_cartesian_error_analytic(oracle):
    """cartesian_utils::computeCartesianError (cartesian_utils.cpp:79-96): for a rotation of angle th about
    axis k between actual and desired, the quaternion error is sin(th/2) k (expressed in world)."""
    import ctypes as C
    L = oracle.lib()
    dp = C.POINTER(C.c_double)
    rng = np.random.default_rng(3)
    for _ in range(50):
        k = rng.normal(size=3); k /= np.linalg.norm(k)
        th = rng.uniform(-1.0, 1.0)
        K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        Rrel = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
        w = rng.normal(size=3); w *= rng.uniform(0, 3.0) / np.linalg.norm(w)
        Kw = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        a = np.linalg.norm(w)
        R = np.eye(3) + np.sin(a) / a * Kw + (1 - np.cos(a)) / a ** 2 * Kw @ Kw
        Rd = Rrel @ R            # desired = world-frame rotation of the actual
        p = rng.normal(size=3); pd = rng.normal(size=3)
        ep = np.zeros(3); eo = np.zeros(3)
        Rc, Rdc = np.ascontiguousarray(R), np.ascontiguousarray(Rd)
        L.orc_cartesian_error(Rc.ctypes.data_as(dp), p.ctypes.data_as(dp), Rdc.ctypes.data_as(dp),
                              pd.ctypes.data_as(dp), ep.ctypes.data_as(dp), eo.ctypes.data_as(dp))
        np.testing.assert_allclose(ep, pd - p, atol=0)
        # e = qd.w*eps - q.w*epsd + epsd x eps = -(vector part of qd * conj(q)) = -sin(th/2) k
        np.testing.assert_allclose(eo, -np.sin(th / 2) * k, atol=1e-12)


def test_rot_to_quat_branches(oracle):
    """Eigen's Quaterniond(Matrix3d): all four branches give a unit quaternion that reproduces R."""
    import ctypes as C
    L = oracle.lib(); dp = C.POINTER(C.c_double)
    def quat_to_rot(q):
        x, y, z, w = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    for axis in range(3):
        for ang in (0.3, 3.0, np.pi):    # small angle: trace > 0; near pi: the three diagonal branches
            k = np.zeros(3); k[axis] = 1.0
            K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
            R = np.ascontiguousarray(np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K)
            q = np.zeros(4)
            L.orc_rot_to_quat(R.ctypes.data_as(dp), q.ctypes.data_as(dp))
            assert abs(np.linalg.norm(q) - 1) < 1e-12
            np.testing.assert_allclose(quat_to_rot(q), R, atol=1e-12)


def test_joint_limits_and_collision_rows(oracle):
    """JointLimits.cpp:47-52 (bounds always contain 0) and CollisionAvoidance.cpp:96-152 (skip beyond the
    detection threshold, cap at 0, unused rows are zero with [-DBL_MAX, DBL_MAX])"""
    import ctypes as C
    L = oracle.lib(); dp = C.POINTER(C.c_double)
    p = lambda a: a.ctypes.data_as(dp)
    q = np.array([0.0, 2.0, -3.0]); qmin = np.array([-1.0, -1.0, -1.0]); qmax = np.array([1.0, 1.0, 1.0])
    l = np.zeros(3); u = np.zeros(3)
    L.orc_joint_limits(3, p(q), p(qmin), p(qmax), 1.0, p(l), p(u))
    np.testing.assert_array_equal(l, [-1, -3, 0]); np.testing.assert_array_equal(u, [1, 0, 4])
    n, P = 4, 4
    Jd = np.arange(P * n, dtype=float).reshape(P, n) + 1
    d = np.array([-0.01, 0.02, 0.2, 0.03])
    A = np.ones((P, n)); lo = np.zeros(P); up = np.zeros(P)
    L.orc_collision_rows(n, P, P, p(Jd), p(d), 0.0, 0.05, 2.0, p(A), p(lo), p(up))
    np.testing.assert_array_equal(A[0], -Jd[0]); np.testing.assert_array_equal(A[1], -Jd[1])
    np.testing.assert_array_equal(A[2], -Jd[3]); np.testing.assert_array_equal(A[3], 0 * Jd[0])
    np.testing.assert_array_equal(up, [0.0, 0.04, 0.06, np.finfo(float).max])
    assert (lo == -np.finfo(float).max).all()


def test_id_row_producers(oracle):
    """friction-cone pyramid (FrictionCone.cpp:35-56), torque-limit bounds (TorqueLimits.cpp:44-45) and the
    acceleration joint-limit bounds (constraints/acceleration/JointLimits.cpp:58-131) on hand-checkable inputs"""
    import ctypes as C
    L = oracle.lib(); dp = C.POINTER(C.c_double)
    p = lambda a: a.ctypes.data_as(dp)
    A = np.zeros(15)
    L.orc_friction_cone_rows(p(np.eye(3).ravel().copy()), 0.8, p(A))
    m = 0.8 / np.sqrt(2.0)
    np.testing.assert_allclose(A.reshape(5, 3), [[1, 0, -m], [-1, 0, -m], [0, 1, -m], [0, -1, -m], [0, 0, -1]])
    # a force along the contact normal satisfies A f <= 0; a tangential one beyond the cone does not
    Rz = np.array([[0.0, -1, 0], [1, 0, 0], [0, 0, 1]])
    L.orc_friction_cone_rows(p(Rz.ravel().copy()), 0.8, p(A))
    assert (A.reshape(5, 3) @ (Rz @ np.array([0, 0, 10.0])) <= 1e-12).all()
    assert (A.reshape(5, 3) @ (Rz @ np.array([9.0, 0, 10.0])) > 0).any()
    lo = np.zeros(2); up = np.zeros(2)
    L.orc_torque_limit_bounds(2, p(np.array([3.0, -4.0])), p(np.array([10.0, 10.0])), p(lo), p(up))
    np.testing.assert_array_equal(lo, [-13, -6]); np.testing.assert_array_equal(up, [7, 14])
    # joint in the middle of its range at rest: the admissible accelerations are symmetric
    lo = np.zeros(1); up = np.zeros(1)
    L.orc_acc_joint_limits(1, p(np.array([0.0])), p(np.array([0.0])), p(np.array([-1.0])), p(np.array([1.0])),
                           p(np.array([100.0])), 0.02, p(lo), p(up))
    assert up[0] > 0 and lo[0] == pytest.approx(-up[0])
    # joint at its upper limit moving towards it: no positive acceleration allowed
    L.orc_acc_joint_limits(1, p(np.array([1.0])), p(np.array([0.5])), p(np.array([-1.0])), p(np.array([1.0])),
                           p(np.array([100.0])), 0.02, p(lo), p(up))
    assert up[0] < 0
    L.orc_acc_velocity_limits(1, p(np.array([1.0])), p(np.array([3.0])), 0.001, 20.0, p(lo), p(up))
    assert up[0] == pytest.approx(100.0) and lo[0] == pytest.approx(-200.0)
