"""Batched kinematics producer: the numpy restatement against finite differences (CPU), the HIP kernel
against the restatement (GPU), and a closed loop: kinematics -> AutoStack::update -> cascade -> q += dq, all on the
device, the end effectors converging on their targets like examples/cpp/coman_ik.cpp:174-219."""
import numpy as np
import pytest

from opensot_amd import abi, kinematics as kin
from opensot_amd.plan import StackPlan, Task, Bound, eps_abs_from_factor
from oracle import pykin


def _fd_jacobians(m, q, h=1e-6):
    o = pykin.forward(m, q)
    Jf = [np.zeros((6, m.n)) for _ in m.frames]
    Jc = np.zeros((3, m.n))
    for j in range(m.n):
        dq = np.zeros(m.n); dq[j] = h
        a, b = pykin.forward(m, q + dq), pykin.forward(m, q - dq)
        Jc[:, j] = (a["com"] - b["com"]) / (2 * h)
        for f in range(len(m.frames)):
            Jf[f][:3, j] = (a["frame_p"][f] - b["frame_p"][f]) / (2 * h)
            S = (a["frame_R"][f] - b["frame_R"][f]) / (2 * h) @ o["frame_R"][f].T     # [w]x
            Jf[f][3:, j] = [S[2, 1], S[0, 2], S[1, 0]]
    return o, Jf, Jc


def test_restatement_matches_finite_differences():
    m = kin.humanoid32()
    assert m.n == 32 and abs(m.mass.sum() - 32.3) < 1e-12
    rng = np.random.default_rng(3)
    for _ in range(3):
        q = rng.uniform(-0.8, 0.8, m.n)
        o, Jf, Jc = _fd_jacobians(m, q)
        for f in range(len(m.frames)):
            assert np.abs(Jf[f] - o["J"][f]).max() < 1e-8
            assert np.abs(o["frame_R"][f] @ o["frame_R"][f].T - np.eye(3)).max() < 1e-14
        assert np.abs(Jc - o["Jcom"]).max() < 1e-8


def test_pair_distances_match_finite_differences():
    """self-collision pairs (what CollisionAvoidance.cpp:96-118 obtains from the collision module): the
    distance rows J_d of the restatement against central differences of its own distances; capsule-capsule,
    capsule-sphere and sphere-sphere pairs of the 32-DoF humanoid"""
    m = kin.humanoid32_pairs(kin.humanoid32())
    assert len(m.pairs) == 16
    rng = np.random.default_rng(12)
    h = 1e-6
    for _ in range(4):
        q = rng.uniform(-0.7, 0.7, m.n)
        d, J = pykin.pair_distances(m, q)
        Jfd = np.zeros_like(J)
        for j in range(m.n):
            e = np.zeros(m.n); e[j] = h
            Jfd[:, j] = (pykin.pair_distances(m, q + e)[0] - pykin.pair_distances(m, q - e)[0]) / (2 * h)
        assert np.abs(J - Jfd).max() < 5e-8
        assert np.abs(J[:, :3]).max() < 1e-12          # a rigid translation of the base changes no distance
    # known answers: two spheres, and two parallel capsules (degenerate closest pair: any point of the overlap)
    ca, cb = pykin.closest_segment_points(np.array([0.0, 0, 0]), np.array([0.0, 0, 0]), np.array([3.0, 4, 0]), np.array([3.0, 4, 0]))
    assert np.linalg.norm(ca - cb) == 5.0
    ca, cb = pykin.closest_segment_points(np.array([0.0, 0, 0]), np.array([1.0, 0, 0]), np.array([0.5, 2, 0]), np.array([3.0, 2, 0]))
    assert abs(np.linalg.norm(ca - cb) - 2.0) < 1e-15
    ca, cb = pykin.closest_segment_points(np.array([0.0, 0, 0]), np.array([1.0, 0, 0]), np.array([2.0, 1, 0]), np.array([2.0, 3, 0]))
    assert np.allclose(ca, [1, 0, 0]) and np.allclose(cb, [2, 1, 0])


def test_emulated_pair_distances_match_restatement():
    from helpers import emu_kinematics
    m = kin.humanoid32_pairs(kin.humanoid32())
    rng = np.random.default_rng(8)
    q = rng.uniform(-1.0, 1.0, (6, m.n))
    poses, J, com, pd, pJ = emu_kinematics(m, q)
    P = len(m.pairs)
    for i in range(6):
        d, Jd = pykin.pair_distances(m, q[i])
        assert np.abs(pd[i] - d).max() < 1e-14 and np.abs(pJ[i, :P] - Jd).max() < 1e-13
    assert (pJ[:, P] == 7.0).all()


def _coman():
    import os
    return kin.from_json(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "coman_tree.json"))


def test_coman_tree_fixture():
    """the reference's robot (tests/robots/coman_floating_base, reduced to a tree fixture by
    tests/golden/make_coman_tree.py): 6 virtual + 29 revolute joints, 31.46 kg; restatement vs finite differences and
    the emulated kernel vs the restatement on it (35 joints: more than half of the 64 lanes busy)"""
    from helpers import emu_kinematics
    m, lo, up = _coman()
    assert m.n == 35 and abs(m.mass.sum() - 31.463860196) < 1e-9 and [f[0] for f in m.frames] == ["l_wrist", "r_wrist", "l_sole", "r_sole"]
    assert np.isinf(lo[:6]).all() and np.isfinite(lo[6:]).all() and (up[6:] > lo[6:]).all()
    rng = np.random.default_rng(8)
    q = np.clip(rng.uniform(-0.6, 0.6, (3, m.n)), np.maximum(lo, -3.0), np.minimum(up, 3.0))
    o, Jf, Jc = _fd_jacobians(m, q[0])
    for f in range(4):
        assert np.abs(Jf[f] - o["J"][f]).max() < 1e-8
    assert np.abs(Jc - o["Jcom"]).max() < 1e-8
    # standing at q = 0 the soles are level and 0.5168 m (the URDF's reference-joint offset) plus the leg below the waist
    z0 = pykin.forward(m, np.zeros(m.n))
    assert abs(z0["frame_p"][2][2] - z0["frame_p"][3][2]) < 1e-12 and abs(z0["frame_p"][2][1] + z0["frame_p"][3][1]) < 1e-9
    poses, J, com = emu_kinematics(m, q)
    for i in range(3):
        oi = pykin.forward(m, q[i])
        for f in range(4):
            assert np.abs(J[i, 6 * f:6 * f + 6] - oi["J"][f]).max() < 1e-13
            assert np.abs(poses[f][i][9:] - oi["frame_p"][f]).max() < 1e-14
        assert np.abs(J[i, 24:27] - oi["Jcom"]).max() < 1e-14 and np.abs(com[i] - oi["com"]).max() < 1e-14


def test_kin_desc_checks_without_gpu():
    L = abi.lib()
    import ctypes as C
    d = kin.humanoid32().desc()
    h = C.c_void_p()
    d.parent[5] = 7      # not in tree order
    assert L.osot_kin_create(C.byref(d), 0, C.byref(h)) == abi.ERR_INVALID
    d = kin.humanoid32().desc(); d.n = 65
    assert L.osot_kin_create(C.byref(d), 0, C.byref(h)) == abi.ERR_INVALID
    d = kin.humanoid32().desc(); d.frame_joint[1] = 40
    assert L.osot_kin_create(C.byref(d), 0, C.byref(h)) == abi.ERR_INVALID
    assert L.osot_kinematics(None, None, None) == abi.ERR_INVALID


def test_emulated_kernel_matches_restatement():
    """the kernel body (pointer jumping, ancestor masks, subtree aggregates) run lane by lane on the host emulator"""
    from helpers import emu_kinematics
    forest = kin.humanoid32()       # (round 4: the CoM Jacobian comes from prefix sums over a depth-first numbering -- a second root,
    forest.parent = list(forest.parent)                       #  the left arm cut loose from the torso, must get its own range of it)
    forest.parent[forest.names.index("LShSag")] = -1
    for m in (kin.humanoid32(), forest):
        rng = np.random.default_rng(6)
        q = rng.uniform(-1.0, 1.0, (5, m.n))
        poses, J, com = emu_kinematics(m, q)
        for i in range(5):
            o = pykin.forward(m, q[i])
            for f in range(4):
                assert np.abs(J[i, 6 * f:6 * f + 6] - o["J"][f]).max() < 1e-13
                assert np.abs(poses[f][i][:9].reshape(3, 3) - o["frame_R"][f]).max() < 1e-14
                assert np.abs(poses[f][i][9:] - o["frame_p"][f]).max() < 1e-14
            assert np.abs(J[i, 24:27] - o["Jcom"]).max() < 1e-14
            assert np.abs(com[i] - o["com"]).max() < 1e-14


def _frame_options_model():
    m = kin.humanoid32()
    m.frame_body = {1: True, 2: True}
    m.frame_active_joints = {0: list(range(6, m.n)), 2: [j for j in range(m.n) if j % 3]}    # frame 0: the floating base masked out
    m.com_active_joints = list(range(0, m.n, 2))
    return m


def _expected_with_options(m, o, f):
    """the frame's Jacobian with the reference's post-processing restated on the world Jacobian of oracle/pykin.py: body
    frame = Ad(R') J with Ad(R) = blockdiag(R, R) (Cartesian.cpp:93-100), then the active-joints mask zeroes columns
    (Task::applyActiveJointsMask, Task.h:129-139)"""
    J = o["J"][f].copy()
    if m.frame_body.get(f):
        Rt = o["frame_R"][f].T
        J = np.concatenate([Rt @ J[:3], Rt @ J[3:]], axis=0)
    if f in m.frame_active_joints:
        keep = np.zeros(m.n, dtype=bool); keep[m.frame_active_joints[f]] = True
        J[:, ~keep] = 0.0
    return J


def test_emulated_frame_options():
    """body-frame Jacobians and active-joint masks written by the producer (the emulated kernel body)"""
    from helpers import emu_kinematics
    m = _frame_options_model()
    rng = np.random.default_rng(16)
    q = rng.uniform(-1.0, 1.0, (4, m.n))
    poses, J, com = emu_kinematics(m, q)
    for i in range(4):
        o = pykin.forward(m, q[i])
        for f in range(4):
            assert np.abs(J[i, 6 * f:6 * f + 6] - _expected_with_options(m, o, f)).max() < 1e-13
        Jc = o["Jcom"].copy(); Jc[:, 1::2] = 0.0
        assert np.abs(J[i, 24:27] - Jc).max() < 1e-14
    assert np.abs(J[:, 0:6, :6]).max() == 0.0 and np.abs(J[:, 6:12, :6]).max() > 0.1


def _relative_model():
    """the 32-DoF humanoid with DefaultHumanoidStack's relative Cartesian tasks (tests/DefaultHumanoidStack.cpp:24-35, 52):
    waist2LeftArm / waist2RightArm (the wrists relative to the waist link) and right2LeftLeg (here: l_sole relative to r_sole);
    r_sole stays a world task.  Frame 4 = the waist link frame (base link only: it has no outputs of its own)"""
    m = kin.humanoid32()
    m.frames = list(m.frames) + [("waist", m.names.index("WaistYaw"), kin._rpy(0.1, -0.2, 0.3), (0.01, 0.0, 0.05))]
    m.frame_base = {0: "waist", 1: 4, 2: 3}
    return m


def _expected_relative(m, o, f):
    """frame f as the producer writes it: relative to its base (oracle/pykin.relative, the textbook two-Jacobian route) or the
    world quantities; then BODY (Ad(R') with R the pose the task sees, Cartesian.cpp:93-100) and the column mask"""
    if f in m.frame_base:
        g = m.frame_base[f]
        g = m.frame_index(g) if isinstance(g, str) else g
        R, p, J = pykin.relative(o, f, g)
    else:
        R, p, J = o["frame_R"][f], o["frame_p"][f], o["J"][f].copy()
    if m.frame_body.get(f):
        J = np.concatenate([R.T @ J[:3], R.T @ J[3:]], axis=0)
    if f in m.frame_active_joints:
        keep = np.zeros(m.n, dtype=bool); keep[m.frame_active_joints[f]] = True
        J[:, ~keep] = 0.0
    return R, p, J


def test_relative_restatement_matches_finite_differences():
    """getRelativeJacobian / relative getPose (Cartesian.cpp:75-76, 80-81) restated: the relative Jacobian against central
    differences of the relative pose (linear: d p_rel; angular: d R_rel R_rel' = [w_rel]x, both in base coordinates)"""
    m = _relative_model()
    rng = np.random.default_rng(23)
    h = 1e-6
    for _ in range(3):
        q = rng.uniform(-0.8, 0.8, m.n)
        o = pykin.forward(m, q)
        for f, g in ((0, 4), (1, 4), (2, 3)):
            R, p, J = pykin.relative(o, f, g)
            Jfd = np.zeros((6, m.n))
            for j in range(m.n):
                e = np.zeros(m.n); e[j] = h
                Ra, pa, _ = pykin.relative(pykin.forward(m, q + e), f, g)
                Rb, pb, _ = pykin.relative(pykin.forward(m, q - e), f, g)
                Jfd[:3, j] = (pa - pb) / (2 * h)
                S = (Ra - Rb) / (2 * h) @ R.T
                Jfd[3:, j] = [S[2, 1], S[0, 2], S[1, 0]]
            assert np.abs(J - Jfd).max() < 1e-6      # (the verdict's bound; observed ~1e-9)
            assert np.abs(J - Jfd).max() < 5e-8
            # the floating base moves both links alike: no relative motion
            assert np.abs(J[:, :6]).max() < 1e-12
        # wrists relative to the waist: only waist-to-wrist joints (the arm chains) have columns
        arm = [m.names.index(s) for s in ("LShSag", "LShLat", "LShYaw", "LElbj", "LWrj")]
        J0 = pykin.relative(o, 0, 4)[2]
        assert np.abs(np.delete(J0, arm, axis=1)).max() < 1e-12 and np.abs(J0[:, arm]).max() > 0.1
        # l_sole relative to r_sole: BOTH legs move it (the right leg's joints with a minus sign), nothing else
        legs = [m.names.index(s + j) for s in "RL" for j in ("HipSag", "HipLat", "HipYaw", "KneeSag", "AnkLat", "AnkSag")]
        J2 = pykin.relative(o, 2, 3)[2]
        assert np.abs(np.delete(J2, legs, axis=1)).max() < 1e-12 and (np.abs(J2[:, legs]).max(axis=0) > 1e-3).all()


def test_emulated_relative_base_frames():
    """the producer's relative-base frames (the emulated kernel body; its route -- the difference of two ancestor tests on ONE column
    formula -- against the restatement's difference of two world Jacobians), with and without the BODY option and a column mask"""
    from helpers import emu_kinematics
    for opts in (False, True):
        m = _relative_model()
        if opts:
            m.frame_body = {1: True, 2: True}
            m.frame_active_joints = {0: [j for j in range(m.n) if j % 4]}
        rng = np.random.default_rng(29)
        q = rng.uniform(-1.0, 1.0, (5, m.n))
        poses, J, com = emu_kinematics(m, q)
        for i in range(5):
            o = pykin.forward(m, q[i])
            for f in range(5):
                R, p, Je = _expected_relative(m, o, f)
                assert np.abs(J[i, 6 * f:6 * f + 6] - Je).max() < 1e-12
                assert np.abs(poses[f][i][:9].reshape(3, 3) - R).max() < 1e-13
                assert np.abs(poses[f][i][9:] - p).max() < 1e-13
            assert np.abs(J[i, 30:33] - o["Jcom"]).max() < 1e-14


def test_kin_desc_rejects_a_bad_base_frame():
    L = abi.lib()
    import ctypes as C
    h = C.c_void_p()
    d = _relative_model().desc()
    assert [d.frame_base[f] for f in range(5)] == [5, 5, 4, 0, 0]
    d.frame_base[1] = 2      # its own index + 1: a frame cannot be its own base link (Cartesian.cpp:50-51)
    assert L.osot_kin_create(C.byref(d), 0, C.byref(h)) == abi.ERR_INVALID
    d = _relative_model().desc(); d.frame_base[0] = 6      # beyond the frames
    assert L.osot_kin_create(C.byref(d), 0, C.byref(h)) == abi.ERR_INVALID
    d = _relative_model().desc(); d.frame_base[0] = -1
    assert L.osot_kin_create(C.byref(d), 0, C.byref(h)) == abi.ERR_INVALID
    with pytest.raises(ValueError):
        mm = _relative_model(); mm.frame_base[3] = 3; mm.desc()


@pytest.mark.gpu
@pytest.mark.parametrize("opts", [False, True])
def test_relative_base_frames_gpu(opts, gpu_device):
    """osot_kin_kernel with relative base links (Cartesian.cpp:73-81) against the restatement, <= 1e-12"""
    import torch
    m = _relative_model()
    if opts:
        m.frame_body = {1: True, 2: True}
        m.frame_active_joints = {0: [j for j in range(m.n) if j % 4]}
    K = kin.Kinematics(m, device=0)
    B = 131
    rng = np.random.default_rng(31)
    q = rng.uniform(-1.0, 1.0, (B, m.n))
    dev = torch.device("cuda", 0)
    A = torch.full((B, 30 + 3 + 1, m.n), 7.0, dtype=torch.float64, device=dev)
    poses = {f: torch.zeros((B, 12), dtype=torch.float64, device=dev) for f in range(5)}
    K.forward(torch.as_tensor(q, device=dev), frame_pose=poses, frame_J={f: (A, 6 * f) for f in range(5)}, com_J=(A, 30))
    torch.cuda.synchronize()
    Ah = A.cpu().numpy()
    for i in list(range(0, B, 11)) + [B - 1]:
        o = pykin.forward(m, q[i])
        for f in range(5):
            R, p, Je = _expected_relative(m, o, f)
            assert np.abs(Ah[i, 6 * f:6 * f + 6] - Je).max() < 1e-12
            ph = poses[f][i].cpu().numpy()
            assert np.abs(ph[:9].reshape(3, 3) - R).max() < 1e-13 and np.abs(ph[9:] - p).max() < 1e-13
        assert np.abs(Ah[i, 30:33] - o["Jcom"]).max() < 1e-14
    assert (Ah[:, 33] == 7.0).all()
    # a base frame WITHOUT outputs (no pose, no Jacobian rows of its own): the relative frames are the same
    A2 = torch.full((B, 24, m.n), 7.0, dtype=torch.float64, device=dev)
    K.forward(torch.as_tensor(q, device=dev), frame_J={f: (A2, 6 * f) for f in range(4)})
    torch.cuda.synchronize()
    assert torch.equal(A2, A[:, :24])


@pytest.mark.gpu
def test_frame_options_gpu(gpu_device):
    import torch
    m = _frame_options_model()
    K = kin.Kinematics(m, device=0)
    B = 130
    rng = np.random.default_rng(17)
    q = rng.uniform(-1.0, 1.0, (B, m.n))
    dev = torch.device("cuda", 0)
    A = torch.full((B, 27, m.n), 7.0, dtype=torch.float64, device=dev)
    K.forward(torch.as_tensor(q, device=dev), frame_J={f: (A, 6 * f) for f in range(4)}, com_J=(A, 24))
    torch.cuda.synchronize()
    Ah = A.cpu().numpy()
    for i in range(0, B, 13):
        o = pykin.forward(m, q[i])
        for f in range(4):
            assert np.abs(Ah[i, 6 * f:6 * f + 6] - _expected_with_options(m, o, f)).max() < 1e-13
        Jc = o["Jcom"].copy(); Jc[:, 1::2] = 0.0
        assert np.abs(Ah[i, 24:27] - Jc).max() < 1e-14


@pytest.mark.gpu
def test_kernel_matches_restatement(gpu_device):
    import torch
    m = kin.humanoid32()
    K = kin.Kinematics(m, device=0)
    B = 257
    rng = np.random.default_rng(5)
    q = rng.uniform(-1.0, 1.0, (B, m.n))
    dev = torch.device("cuda", 0)
    tq = torch.as_tensor(q, device=dev)
    A = torch.full((B, 24 + 3 + 2, m.n), 7.0, dtype=torch.float64, device=dev)    # 4 frames + CoM rows + 2 spare rows
    poses = {f: torch.zeros((B, 12), dtype=torch.float64, device=dev) for f in range(4)}
    com = torch.zeros((B, 3), dtype=torch.float64, device=dev)
    K.forward(tq, frame_pose=poses, frame_J={f: (A, 6 * f) for f in range(4)}, com=com, com_J=(A, 24))
    torch.cuda.synchronize()
    Ah = A.cpu().numpy()
    for i in list(range(0, B, 7)) + [B - 2, B - 1]:      # (odd and even instances: two share a wavefront; B is odd)
        o = pykin.forward(m, q[i])
        for f in range(4):
            assert np.abs(Ah[i, 6 * f:6 * f + 6] - o["J"][f]).max() < 1e-13
            ph = poses[f][i].cpu().numpy()
            assert np.abs(ph[:9].reshape(3, 3) - o["frame_R"][f]).max() < 1e-14
            assert np.abs(ph[9:] - o["frame_p"][f]).max() < 1e-14
        assert np.abs(Ah[i, 24:27] - o["Jcom"]).max() < 1e-14
        assert np.abs(com[i].cpu().numpy() - o["com"]).max() < 1e-14
    assert (Ah[:, 27:] == 7.0).all()          # rows outside the producers' ranges are untouched


@pytest.mark.gpu
def test_closed_loop_ik_on_device(gpu_device):
    """coman_ik.cpp:174-219 with everything resident: q -> poses/Jacobians (osot_kinematics) -> b, box
    (osot_stack_update) -> dq (osot_ihqp_solve) -> q += dq.  Stack of BASELINE config 3."""
    import torch
    from opensot_amd.solver import BatchedStack
    m = kin.humanoid32()
    n, B = m.n, 128
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(9)
    q0 = np.zeros((B, n))
    q0[:, [m.names.index(s + "KneeSag") for s in "RL"]] = 0.5
    q0[:, [m.names.index(s + "HipSag") for s in "RL"]] = -0.25
    q0[:, [m.names.index(s + "AnkSag") for s in "RL"]] = -0.25
    q0[:, [m.names.index(s + "Elbj") for s in "RL"]] = -0.6
    q0 += rng.normal(0.0, 0.02, (B, n))
    levels = [[Task(abi.TASK_COM, 3, lam=0.1, name="com")],
              [Task(abi.TASK_CARTESIAN, 6, weight=0.1, lam=0.1, name="l_wrist"), Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="r_wrist"),
               Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="l_sole"), Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="r_sole")],
              [Task(abi.TASK_POSTURAL, n, lam=0.01, name="postural")]]
    bounds = [Bound(abi.BOUND_JOINT_LIMITS, scaling=1.0, name="jl"), Bound(abi.BOUND_VELOCITY_LIMITS, dT=0.01, name="vl")]
    plan = StackPlan(n=n, levels=levels, bounds=bounds, rowblocks=[], eps_abs=eps_abs_from_factor(1e6))
    st = BatchedStack(plan, B, device=0, want_levels=False)
    K = kin.Kinematics(m, device=0)
    f64 = dict(dtype=torch.float64, device=dev)
    q = torch.as_tensor(q0, **f64).contiguous()
    pose = [torch.zeros((B, 12), **f64) for _ in range(4)]
    com = torch.zeros((B, 3), **f64)

    def fk():
        K.forward(q, frame_pose={f: pose[f] for f in range(4)}, frame_J={f: (st.A[1], 6 * f) for f in range(4)},
                  com=com, com_J=(st.A[0], 0))
    fk()
    torch.cuda.synchronize()
    # targets: feet stay, wrists move 6 cm, CoM 2 cm
    pose_d = [p.clone() for p in pose]
    pose_d[0][:, 9:] += torch.as_tensor([0.04, 0.03, 0.03], **f64)
    pose_d[1][:, 9:] += torch.as_tensor([0.04, -0.03, 0.03], **f64)
    com_d = com.clone(); com_d[:, 0] += 0.02
    qmin = torch.full((B, n), -2.5, **f64); qmax = torch.full((B, n), 2.5, **f64)
    qdot_max = torch.full((B, n), 2.0, **f64)
    q_ref = q.clone()
    leaf = {"B": B, "task": [[(com, com_d, None)], [(pose[f], pose_d[f], None) for f in range(4)], [(q, q_ref, None)]],
            "bound": [(q, qmin, qmax), (qdot_max, None, None)], "rows": []}
    err0 = None
    for cycle in range(300):
        fk()
        st.update(leaf)
        st.solve(B)
        q += st.dq[:B]
        if cycle == 0:
            torch.cuda.synchronize()
            assert (st.status[:B] == 0).all()
            err0 = [float((pose_d[f][:, 9:] - pose[f][:, 9:]).norm(dim=1).max()) for f in range(4)]
    fk()
    torch.cuda.synchronize()
    assert (st.status[:B] == 0).all()
    err = [float((pose_d[f][:, 9:] - pose[f][:, 9:]).norm(dim=1).max()) for f in range(4)]
    assert err0[0] > 0.05 and err0[1] > 0.05
    assert err[1] < 0.05 * err0[1]                 # r_wrist (full weight) converged (shares its level with the feet)
    assert err[0] < 0.2 * err0[0]                  # l_wrist (weight 0.1, same level) follows
    assert err[2] < 5e-3 and err[3] < 5e-3         # the feet (same level as the wrists: a least-squares compromise) stayed
    assert float((com_d - com).norm(dim=1).max()) < 2e-3
    # the restatement agrees with the device state at the end
    o = pykin.forward(m, q[3].cpu().numpy())
    assert np.abs(pose[1][3].cpu().numpy()[9:] - o["frame_p"][1]).max() < 1e-12


@pytest.mark.gpu
def test_closed_loop_default_humanoid_stack_relative_tasks(gpu_device):
    """A DefaultHumanoidStack-shaped stack (tests/DefaultHumanoidStack.cpp:24-35, 52; the shape of
    tests/solvers/TestQPOases_AutoStack.cpp:80-83) closed on the device with RELATIVE Cartesian tasks produced by the kinematics
    kernel: (rightLeg + right2LeftLeg) / (waist2LeftArm + waist2RightArm) / postural << jointLimits << velocityLimits.  The wrists
    reach their targets GIVEN IN THE WAIST FRAME, the feet keep their relative pose, and the device state agrees with the
    restatement at the end."""
    import torch
    from opensot_amd.solver import BatchedStack
    m = _relative_model()
    # the base link of DefaultHumanoidStack is the robot's "Waist" = its PELVIS (the floating-base link): three waist joints and five arm
    # joints lie between it and a wrist (a 6-D task relative to the torso link would face the arm's five joints alone)
    m.frames[4] = ("waist", m.names.index("base_yaw"), kin._rpy(0.1, -0.2, 0.3), (0.01, 0.0, 0.05))
    n, B = m.n, 96
    dev = torch.device("cuda", 0)
    f64 = dict(dtype=torch.float64, device=dev)
    rng = np.random.default_rng(37)
    q0 = np.zeros((B, n))
    q0[:, [m.names.index(s + "KneeSag") for s in "RL"]] = 0.5
    q0[:, [m.names.index(s + "HipSag") for s in "RL"]] = -0.25
    q0[:, [m.names.index(s + "AnkSag") for s in "RL"]] = -0.25
    q0[:, [m.names.index(s + "Elbj") for s in "RL"]] = -0.6
    q0 += rng.normal(0.0, 0.02, (B, n))
    levels = [[Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="r_sole"), Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="right2LeftLeg")],
              [Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="waist2LeftArm"), Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="waist2RightArm")],
              [Task(abi.TASK_POSTURAL, n, lam=0.01, name="postural")]]
    bounds = [Bound(abi.BOUND_JOINT_LIMITS, scaling=1.0, name="jl"), Bound(abi.BOUND_VELOCITY_LIMITS, dT=0.01, name="vl")]
    plan = StackPlan(n=n, levels=levels, bounds=bounds, rowblocks=[], eps_abs=eps_abs_from_factor(1e6))
    st = BatchedStack(plan, B, device=0, want_levels=False)
    K = kin.Kinematics(m, device=0)
    q = torch.as_tensor(q0, **f64).contiguous()
    pose = [torch.zeros((B, 12), **f64) for _ in range(4)]
    # frames: 0 l_wrist|waist, 1 r_wrist|waist, 2 l_sole|r_sole, 3 r_sole (world); rows: level 0 = [r_sole; l_sole|r_sole], level 1 = wrists
    kw = dict(frame_pose={f: pose[f] for f in range(4)}, frame_J={3: (st.A[0], 0), 2: (st.A[0], 6), 0: (st.A[1], 0), 1: (st.A[1], 6)})
    K.forward(q, **kw)
    torch.cuda.synchronize()
    pose_d = [p.clone() for p in pose]
    pose_d[0][:, 9:] += torch.as_tensor([0.05, 0.02, 0.04], **f64)      # in WAIST coordinates
    pose_d[1][:, 9:] += torch.as_tensor([0.05, -0.02, 0.04], **f64)
    qmin = torch.full((B, n), -2.5, **f64); qmax = torch.full((B, n), 2.5, **f64)
    leaf = {"B": B, "task": [[(pose[3], pose_d[3], None), (pose[2], pose_d[2], None)], [(pose[0], pose_d[0], None), (pose[1], pose_d[1], None)],
                             [(q, q.clone(), None)]],
            "bound": [(q, qmin, qmax), (torch.full((B, n), 2.0, **f64), None, None)], "rows": []}
    kb = K.batch_args(q, **kw)
    err0 = [float((pose_d[f][:, 9:] - pose[f][:, 9:]).norm(dim=1).max()) for f in range(4)]
    for cycle in range(300):
        st.control_cycle(K, kb, leaf, q_integrate=q)       # (kinematics incl. the relative frames inside the launch)
        if cycle == 0:
            torch.cuda.synchronize()
            assert (st.status[:B] == 0).all()
    K.forward(q, **kw)
    torch.cuda.synchronize()
    assert (st.status[:B] == 0).all()
    err = [float((pose_d[f][:, 9:] - pose[f][:, 9:]).norm(dim=1).max()) for f in range(4)]
    assert err0[0] > 0.05 and err0[1] > 0.05
    assert err[0] < 0.1 * err0[0] and err[1] < 0.1 * err0[1]          # the wrists reached their waist-frame targets (12 rows, 13 joints, one level)
    assert err[2] < 1e-4 and err[3] < 1e-4                            # the feet (first level) kept their world / relative poses
    i = 5
    o = pykin.forward(m, q[i].cpu().numpy())
    for f, g in ((0, 4), (1, 4), (2, 3)):
        R, p, _ = pykin.relative(o, f, g)
        ph = pose[f][i].cpu().numpy()
        assert np.abs(ph[9:] - p).max() < 1e-12 and np.abs(ph[:9].reshape(3, 3) - R).max() < 1e-12
    # relative tasks have no floating-base columns (the base moves both links alike): nothing in this stack asks the base to move, and
    # it stayed where the Postural reference holds it -- the wrists were brought to their targets by the waist and arm joints alone
    assert float((q[:, :6] - torch.as_tensor(q0[:, :6], **f64)).abs().max()) < 1e-9
    assert float((q[:, 6:9] - torch.as_tensor(q0[:, 6:9], **f64)).abs().max()) > 1e-3      # (the waist joints did work)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["box", "rows", "collision_pairs", "hotstart", "inactive_task", "dense_weight", "relative_base"])
def test_control_cycle_in_one_launch_matches_the_three_calls(mode, gpu_device):
    """osot_control_cycle (per instance kinematics -> AutoStack::update -> cascade -> q += dq by the same wavefront,
    coman_ik.cpp:186-219 in ONE launch) against osot_kinematics + osot_cycle + the integration as three launches, over 25
    closed-loop steps from the same start: every array both write (poses, CoM, the Jacobian rows in A_k, b_k, the box, dq, q)
    is bit-identical.  Modes: "box" a plan without constraint rows (the BOX instantiation), "rows" with generic rows (the
    general one); round 5 -- every plan the three-call form takes: "collision_pairs" BASELINE config 4's shape, the
    CollisionAvoidance rows PRODUCED ON THE DEVICE inside the launch (capsule-pair distances and distance Jacobians,
    velocity/CollisionAvoidance.cpp:96-152), "hotstart" (osot_solver_set_hotstart), "inactive_task" (Task::setActive(false))
    and "dense_weight" (Task::setWeight(W)) through the EXTRA instantiation; round 6 -- "relative_base": DefaultHumanoidStack's
    waist2LeftArm / waist2RightArm / right2LeftLeg (Cartesian tasks with a base link, Cartesian.cpp:73-81) produced inside the launch.
    An odd batch."""
    import torch
    from opensot_amd.plan import Rows
    from opensot_amd.solver import BatchedStack
    pairs = mode == "collision_pairs"
    m = kin.humanoid32_pairs(kin.humanoid32()) if pairs else (_relative_model() if mode == "relative_base" else kin.humanoid32())
    n, B = m.n, 203
    P = len(m.pairs) if pairs else 0
    dev = torch.device("cuda", 0)
    f64 = dict(dtype=torch.float64, device=dev)
    rng = np.random.default_rng(19)
    q0 = np.zeros((B, n))
    q0[:, [m.names.index(s + "KneeSag") for s in "RL"]] = 0.5
    q0[:, [m.names.index(s + "HipSag") for s in "RL"]] = -0.25
    q0[:, [m.names.index(s + "AnkSag") for s in "RL"]] = -0.25
    q0[:, [m.names.index(s + "Elbj") for s in "RL"]] = -0.6 if not pairs else -0.9
    if pairs:
        q0[:, m.names.index("RShLat")] = -0.35; q0[:, m.names.index("LShLat")] = 0.35
    q0 += rng.normal(0.0, 0.02, (B, n))
    Wd = None
    if mode == "dense_weight":
        Mw = rng.normal(size=(6, 6)); Wd = Mw @ Mw.T / 6.0 + np.eye(6)
    levels = [[Task(abi.TASK_COM, 3, lam=0.1, name="com")],
              [Task(abi.TASK_CARTESIAN, 6, weight=0.1, lam=0.1, name="l_wrist", dense_weight=(mode == "dense_weight")),
               Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="r_wrist"),
               Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="l_sole"), Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="r_sole")],
              [Task(abi.TASK_POSTURAL, n, lam=0.01, name="postural")]]
    bounds = [Bound(abi.BOUND_JOINT_LIMITS, scaling=1.0, name="jl"), Bound(abi.BOUND_VELOCITY_LIMITS, dT=0.01, name="vl")]
    rowblocks = []
    if mode == "rows":
        rowblocks = [Rows(abi.ROWS_GENERIC, 2, name="rows")]
    if pairs:
        rowblocks = [Rows(abi.ROWS_COLLISION, P, d_threshold=0.02, detection_threshold=0.0, bound_scaling=0.2, name="self_collision")]
    plan = StackPlan(n=n, levels=levels, bounds=bounds, rowblocks=rowblocks, eps_abs=eps_abs_from_factor(1e6))
    K = kin.Kinematics(m, device=0)
    Crow = torch.as_tensor(rng.normal(size=(B, 2, n)), **f64).contiguous()
    rlo = torch.full((B, 2), -0.05, **f64); rup = torch.full((B, 2), 0.05, **f64)

    def make():
        st = BatchedStack(plan, B, device=0, want_levels=False)
        if mode == "hotstart":
            st.set_hotstart(True)
        if mode == "inactive_task":
            st.set_task_active(1, 1, False)
        q = torch.as_tensor(q0, **f64).contiguous()
        pose = [torch.zeros((B, 12), **f64) for _ in range(4)]
        com = torch.zeros((B, 3), **f64)
        kw = dict(frame_pose={f: pose[f] for f in range(4)}, frame_J={f: (st.A[1], 6 * f) for f in range(4)}, com=com, com_J=(st.A[0], 0))
        Jd = dist = None
        if pairs:
            Jd = torch.zeros((B, P, n), **f64); dist = torch.zeros((B, P), **f64)
            kw.update(pair_dist=dist, pair_J=(Jd, 0))
        K.forward(q, **kw)
        torch.cuda.synchronize()
        pose_d = [p.clone() for p in pose]
        if pairs:      # both hands to the same point: the capsule rows become active (test_closed_loop_self_collision_avoidance_on_device)
            mid = 0.5 * (pose[0][:, 9:] + pose[1][:, 9:])
            pose_d[0][:, 9:] = mid; pose_d[1][:, 9:] = mid
        else:
            pose_d[0][:, 9:] += torch.as_tensor([0.04, 0.03, 0.03], **f64)
            pose_d[1][:, 9:] += torch.as_tensor([0.04, -0.03, 0.03], **f64)
        com_d = com.clone(); com_d[:, 0] += 0.02
        qmin = torch.full((B, n), -2.5, **f64); qmax = torch.full((B, n), 2.5, **f64)
        qdot_max = torch.full((B, n), 2.0, **f64)
        Wt = None if Wd is None else torch.as_tensor(np.tile(Wd, (B, 1, 1)), **f64).contiguous()
        leaf = {"B": B, "task": [[(com, com_d, None)], [(pose[f], pose_d[f], None) for f in range(4)], [(q, q.clone(), None)]],
                "bound": [(q, qmin, qmax), (qdot_max, None, None)],
                "rows": [(Crow, rlo, rup)] if mode == "rows" else ([(Jd, dist, None)] if pairs else [])}
        if Wt is not None:
            leaf["W"] = [[None], [Wt, None, None, None], [None]]
        return st, q, pose, com, kw, leaf, dist

    a = make()
    b = make()
    c = make()
    sta, qa, posea, coma, kwa, leafa, dista = a
    stb, qb, poseb, comb, kwb, leafb, distb = b
    stc, qc, posec, comc, kwc, leafc, distc = c
    kb = K.batch_args(qb, **kwb)
    kc = K.batch_args(qc, **kwc)
    nsteps = 25 if not pairs else 120
    dq_hist = []
    for step in range(nsteps):
        K.forward(qa, **kwa)
        sta.cycle(leafa)
        qa += sta.dq[:B]
        stb.control_cycle(K, kb, leafb, q_integrate=qb)
        dq_hist.append(stb.dq[:B].clone())
    # round 5 -- osot_control_rollout: the same steps as ROLLOUTS (several control cycles of every robot per launch: 7, then 1, then
    # the rest), every cycle's dq and status recorded
    dq_steps = torch.zeros((nsteps, B, n), **f64)
    st_steps = torch.full((nsteps, B), -1, dtype=torch.int32, device=dev)
    done = 0
    for chunk in (7, 1, nsteps - 8):
        stc.control_rollout(K, kc, leafc, qc, chunk, dq_steps=dq_steps[done:done + chunk], status_steps=st_steps[done:done + chunk])
        done += chunk
    torch.cuda.synchronize()
    assert torch.equal(qc, qb) and torch.equal(stc.dq[:B], stb.dq[:B]) and (stc.status[:B] == 0).all()
    assert torch.equal(dq_steps, torch.stack(dq_hist)) and (st_steps == 0).all()
    assert torch.equal(comc, comb) and all(torch.equal(posec[f], poseb[f]) for f in range(4))
    assert (sta.status[:B] == 0).all() and (stb.status[:B] == 0).all()
    assert torch.equal(qa, qb) and torch.equal(sta.dq[:B], stb.dq[:B])
    assert torch.equal(coma, comb) and all(torch.equal(posea[f], poseb[f]) for f in range(4))
    for k in range(2):
        assert torch.equal(sta.A[k][:B], stb.A[k][:B]) and torch.equal(sta.b[k][:B], stb.b[k][:B])
    assert torch.equal(sta.l[:B], stb.l[:B]) and torch.equal(sta.u[:B], stb.u[:B])
    assert float(sta.dq[:B].abs().max()) > 1e-5            # (the loop is still moving: the comparison is not of zeros)
    if pairs:
        assert torch.equal(dista, distb) and torch.equal(sta.C[:B], stb.C[:B]) and torch.equal(sta.up[:B], stb.up[:B])
        assert float(distb.min()) < 0.03                    # (the capsule rows ARE what holds the hands apart by now)
    if mode == "hotstart":
        assert torch.equal(sta.iterations[:B], stb.iterations[:B])


@pytest.mark.gpu
def test_control_cycle_writes_pair_outputs_on_a_plan_without_rows(gpu_device):
    """a plan WITHOUT constraint rows picks the BOX instantiation, whose kinematics stage is compiled without the
    collision-pair stage; a model with pairs whose batch asks for pair_dist / pair_J must still get them written (the launch takes
    the general instantiation) -- they used to keep stale data with no error.  Compared with osot_kinematics' own outputs."""
    import torch
    from opensot_amd.solver import BatchedStack
    m = kin.humanoid32_pairs(kin.humanoid32())
    n, B, P = m.n, 37, len(m.pairs)
    dev = torch.device("cuda", 0)
    f64 = dict(dtype=torch.float64, device=dev)
    rng = np.random.default_rng(41)
    q = torch.as_tensor(rng.uniform(-0.4, 0.4, (B, n)), **f64).contiguous()
    levels = [[Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="r_wrist")], [Task(abi.TASK_POSTURAL, n, lam=0.01, name="postural")]]
    bounds = [Bound(abi.BOUND_VELOCITY_LIMITS, dT=0.01, name="vl")]
    plan = StackPlan(n=n, levels=levels, bounds=bounds, rowblocks=[], eps_abs=eps_abs_from_factor(1e6))      # nc == 0: BOX
    st = BatchedStack(plan, B, device=0, want_levels=False)
    K = kin.Kinematics(m, device=0)
    pose = torch.zeros((B, 12), **f64)
    dist = torch.full((B, P), 7.0, **f64); Jd = torch.full((B, P, n), 7.0, **f64)
    kw = dict(frame_pose={1: pose}, frame_J={1: (st.A[0], 0)}, pair_dist=dist, pair_J=(Jd, 0))
    dist_ref = torch.zeros((B, P), **f64); Jd_ref = torch.zeros((B, P, n), **f64)
    K.forward(q, frame_pose={1: pose}, frame_J={1: (st.A[0], 0)}, pair_dist=dist_ref, pair_J=(Jd_ref, 0))
    torch.cuda.synchronize()
    leaf = {"B": B, "task": [[(pose, pose.clone(), None)], [(q, q.clone(), None)]], "bound": [(torch.full((B, n), 2.0, **f64), None, None)], "rows": []}
    st.control_cycle(K, K.batch_args(q, **kw), leaf, q_integrate=None)
    torch.cuda.synchronize()
    assert (st.status[:B] == 0).all()
    assert torch.equal(dist, dist_ref) and torch.equal(Jd, Jd_ref)
    assert float(dist.max()) < 5.0          # (not the fill value: written)


@pytest.mark.gpu
def test_control_cycle_in_one_launch_on_the_35_coordinate_coman(gpu_device):
    """osot_control_cycle on the reference's own robot (35 coordinates: the 56-lane kernels, the 64-joint producer with one
    robot per wavefront): the stack of coman_ik.cpp:425-449, 20 closed-loop steps, bit-identical to osot_kinematics + osot_cycle
    + q += dq"""
    import torch
    from opensot_amd.solver import BatchedStack
    m, lo, up = _coman()
    n, B = m.n, 77
    dev = torch.device("cuda", 0)
    f64 = dict(dtype=torch.float64, device=dev)
    rng = np.random.default_rng(40)
    q0 = np.zeros((B, n))
    for s_ in "RL":
        q0[:, m.names.index(s_ + "HipSag")] = -0.3; q0[:, m.names.index(s_ + "KneeSag")] = 0.6
        q0[:, m.names.index(s_ + "AnkSag")] = -0.3; q0[:, m.names.index(s_ + "Elbj")] = -0.8
        q0[:, m.names.index(s_ + "ShSag")] = 0.2
    q0[:, m.names.index("LShLat")] = 0.3; q0[:, m.names.index("RShLat")] = -0.3
    q0[:, 6:] += rng.normal(0.0, 0.01, (B, n - 6))
    q0 = np.clip(q0, np.maximum(lo, -10.0) + 1e-3, np.minimum(up, 10.0) - 1e-3)
    levels = [[Task(abi.TASK_COM, 3, lam=0.1, name="com")],
              [Task(abi.TASK_CARTESIAN, 6, weight=0.1, lam=0.1, name="l_wrist"), Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="r_wrist"),
               Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="l_sole"), Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="r_sole")],
              [Task(abi.TASK_POSTURAL, n, lam=0.01, name="postural")]]
    bounds = [Bound(abi.BOUND_JOINT_LIMITS, scaling=1.0, name="jl"), Bound(abi.BOUND_VELOCITY_LIMITS, dT=0.01, name="vl")]
    plan = StackPlan(n=n, levels=levels, bounds=bounds, rowblocks=[], eps_abs=eps_abs_from_factor(1e6))
    K = kin.Kinematics(m, device=0)
    big = 1.0e3

    def make():
        st = BatchedStack(plan, B, device=0, want_levels=False)
        q = torch.as_tensor(q0, **f64).contiguous()
        pose = [torch.zeros((B, 12), **f64) for _ in range(4)]
        com = torch.zeros((B, 3), **f64)
        kw = dict(frame_pose={f: pose[f] for f in range(4)}, frame_J={f: (st.A[1], 6 * f) for f in range(4)}, com=com, com_J=(st.A[0], 0))
        K.forward(q, **kw)
        torch.cuda.synchronize()
        pose_d = [p.clone() for p in pose]
        pose_d[1][:, 9:] += torch.as_tensor([0.05, -0.02, 0.04], **f64)
        qmin = torch.as_tensor(np.tile(np.maximum(lo, -big), (B, 1)), **f64); qmax = torch.as_tensor(np.tile(np.minimum(up, big), (B, 1)), **f64)
        leaf = {"B": B, "task": [[(com, com.clone(), None)], [(pose[f], pose_d[f], None) for f in range(4)], [(q, q.clone(), None)]],
                "bound": [(q, qmin, qmax), (torch.full((B, n), 2.0, **f64), None, None)], "rows": []}
        return st, q, pose, com, kw, leaf

    sta, qa, posea, coma, kwa, leafa = make()
    stb, qb, poseb, comb, kwb, leafb = make()
    kb = K.batch_args(qb, **kwb)
    for step in range(20):
        K.forward(qa, **kwa)
        sta.cycle(leafa)
        qa += sta.dq[:B]
        stb.control_cycle(K, kb, leafb, q_integrate=qb)
    torch.cuda.synchronize()
    assert (sta.status[:B] == 0).all() and (stb.status[:B] == 0).all()
    assert torch.equal(qa, qb) and torch.equal(sta.dq[:B], stb.dq[:B])
    assert torch.equal(coma, comb) and all(torch.equal(posea[f], poseb[f]) for f in range(4))
    for k in range(2):
        assert torch.equal(sta.A[k][:B], stb.A[k][:B]) and torch.equal(sta.b[k][:B], stb.b[k][:B])
    assert float(sta.dq[:B].abs().max()) > 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["S1", "S2", "S3", "S4"])
def test_coman35_published_stacks_through_nhqp(which, gpu_device):
    """the reference's four published stacks (examples/cpp/coman_ik.cpp:425-449) on its own 35-coordinate robot
    through the null-space front-end, closed loop on the device.  S1 -- ONE level of 50 rows in 35 variables -- was refused until
    round 5 (min(rows, free variables) > 32: osot_nhqp_prepare_wide_kernel).  Every robot solved at every checked step, the URDF's
    joint limits held, the feet (TaskToConstraint rows) stayed, the right wrist moved towards its goal."""
    import sys, os
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from opensot_amd.solver import BatchedStack
    m, lo, up = _coman()
    n, B = m.n, 64
    plan = bench.coman_stack(which, n)
    dev = torch.device("cuda", 0)
    f64 = dict(dtype=torch.float64, device=dev)
    rng = np.random.default_rng(4)
    q0 = np.zeros((B, n))
    for s_ in "RL":
        q0[:, m.names.index(s_ + "HipSag")] = -0.3; q0[:, m.names.index(s_ + "KneeSag")] = 0.6
        q0[:, m.names.index(s_ + "AnkSag")] = -0.3; q0[:, m.names.index(s_ + "Elbj")] = -0.8
        q0[:, m.names.index(s_ + "ShSag")] = 0.2
    q0[:, m.names.index("LShLat")] = 0.3; q0[:, m.names.index("RShLat")] = -0.3
    q0[:, 6:] += rng.normal(0.0, 0.01, (B, n - 6))
    q0 = np.clip(q0, np.maximum(lo, -10.0) + 1e-3, np.minimum(up, 10.0) - 1e-3)
    st = BatchedStack(plan, B, device=0, want_levels=False)
    K = kin.Kinematics(m, device=0)
    q = torch.as_tensor(q0, **f64).contiguous()
    pose = [torch.zeros((B, 12), **f64) for _ in range(4)]
    com = torch.zeros((B, 3), **f64)
    where, off = {}, [0] * plan.L
    for k, lev in enumerate(plan.levels):
        for t in lev:
            if t.name in ("l_wrist", "r_wrist", "com"):
                where[t.name] = (st.A[k], off[k])
            if not t.implicit:
                off[k] += t.rows
    kw = dict(frame_pose={f: pose[f] for f in range(4)}, frame_J={0: where["l_wrist"], 1: where["r_wrist"], 2: (st.C, 0), 3: (st.C, 6)},
              com=com, com_J=where["com"])
    K.forward(q, **kw); torch.cuda.synchronize()
    pose_d = [p.clone() for p in pose]
    pose_d[1][:, 9:] += torch.as_tensor([0.05, -0.02, 0.04], **f64)
    big = 1.0e3
    qmin = torch.as_tensor(np.tile(np.maximum(lo, -big), (B, 1)), **f64); qmax = torch.as_tensor(np.tile(np.minimum(up, big), (B, 1)), **f64)
    leaf_of = {"l_wrist": (pose[0], pose_d[0], None), "r_wrist": (pose[1], pose_d[1], None), "com": (com, com.clone(), None), "postural": (q, q.clone(), None)}
    leaf = {"B": B, "task": [[leaf_of[t.name] for t in lev] for lev in plan.levels],
            "bound": [(q, qmin, qmax), (torch.full((B, n), 2.0, **f64), None, None)], "rows": [(pose[2], pose_d[2], None), (pose[3], pose_d[3], None)]}
    e0 = float((pose_d[1][:, 9:] - pose[1][:, 9:]).norm(dim=1).max())
    for cycle in range(150):
        K.forward(q, **kw); st.update(leaf); st.solve_nhqp(B); q += st.dq[:B]
        if cycle in (0, 75, 149):
            torch.cuda.synchronize()
            assert (st.status[:B] == 0).all(), (which, cycle, st.status[:B].cpu().numpy())
    K.forward(q, **kw); torch.cuda.synchronize()
    qh = q.cpu().numpy()
    assert (qh[:, 6:] >= lo[6:] - 1e-9).all() and (qh[:, 6:] <= up[6:] + 1e-9).all()
    assert float((pose_d[1][:, 9:] - pose[1][:, 9:]).norm(dim=1).max()) < 0.5 * e0
    for f in (2, 3):
        assert float((pose_d[f][:, 9:] - pose[f][:, 9:]).norm(dim=1).max()) < 5e-3


@pytest.mark.gpu
@pytest.mark.parametrize("front_end", ["iHQP", "eHQP", "nHQP"])
def test_closed_loop_ik_coman35(front_end, gpu_device):
    """the reference's own robot and stack -- examples/cpp/coman_ik.cpp:425-449: 35 coordinates,
    (com / 0.1*l_wrist + r_wrist + l_sole + r_sole / postural) << joint limits << velocity limits -- with kinematics,
    update and the (64-lane) cascade all on the device; joint limits from the URDF.  front_end eHQP (round 3: the QR kernel
    takes n <= 64): the same loop through the reference's equality-only front-end (eHQP.cpp:64-95), which ignores the
    bounds -- the targets are reached as well, the limit check does not apply.  front_end nHQP (round 4: 64-column level
    preparation, osot_nhqp_prepare64_kernel): the reference's null-space front-end (nHQP.cpp:155-204) on its own robot, 35 / 32 / 8
    free variables per level; the bounds hold (they are rows of every level's QP there)."""
    import torch
    from opensot_amd.solver import BatchedStack
    m, lo, up = _coman()
    n, B = m.n, 96
    dev = torch.device("cuda", 0)
    f64 = dict(dtype=torch.float64, device=dev)
    rng = np.random.default_rng(4)
    q0 = np.zeros((B, n))
    for s_ in "RL":                                   # a slightly crouched, arms-bent posture inside the limits
        q0[:, m.names.index(s_ + "HipSag")] = -0.3; q0[:, m.names.index(s_ + "KneeSag")] = 0.6
        q0[:, m.names.index(s_ + "AnkSag")] = -0.3; q0[:, m.names.index(s_ + "Elbj")] = -0.8
        q0[:, m.names.index(s_ + "ShSag")] = 0.2
    q0[:, m.names.index("LShLat")] = 0.3; q0[:, m.names.index("RShLat")] = -0.3
    q0[:, 6:] += rng.normal(0.0, 0.01, (B, n - 6))
    q0 = np.clip(q0, np.maximum(lo, -10.0) + 1e-3, np.minimum(up, 10.0) - 1e-3)
    levels = [[Task(abi.TASK_COM, 3, lam=0.1, name="com")],
              [Task(abi.TASK_CARTESIAN, 6, weight=0.1, lam=0.1, name="l_wrist"), Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="r_wrist"),
               Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="l_sole"), Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="r_sole")],
              [Task(abi.TASK_POSTURAL, n, lam=0.01, name="postural")]]
    bounds = [Bound(abi.BOUND_JOINT_LIMITS, scaling=1.0, name="jl"), Bound(abi.BOUND_VELOCITY_LIMITS, dT=0.01, name="vl")]
    plan = StackPlan(n=n, levels=levels, bounds=bounds, rowblocks=[], eps_abs=eps_abs_from_factor(1e6))
    st = BatchedStack(plan, B, device=0, want_levels=False)
    K = kin.Kinematics(m, device=0)
    q = torch.as_tensor(q0, **f64).contiguous()
    pose = [torch.zeros((B, 12), **f64) for _ in range(4)]
    com = torch.zeros((B, 3), **f64)

    def fk():
        K.forward(q, frame_pose={f: pose[f] for f in range(4)}, frame_J={f: (st.A[1], 6 * f) for f in range(4)},
                  com=com, com_J=(st.A[0], 0))
    fk(); torch.cuda.synchronize()
    pose_d = [p.clone() for p in pose]
    pose_d[1][:, 9:] += torch.as_tensor([0.05, -0.02, 0.04], **f64)          # r_wrist target
    com_d = com.clone()
    big = 1.0e3                                                              # "no limit" for the virtual joints
    qmin = torch.as_tensor(np.tile(np.maximum(lo, -big), (B, 1)), **f64); qmax = torch.as_tensor(np.tile(np.minimum(up, big), (B, 1)), **f64)
    qdot_max = torch.full((B, n), 2.0, **f64)
    q_ref = q.clone()
    leaf = {"B": B, "task": [[(com, com_d, None)], [(pose[f], pose_d[f], None) for f in range(4)], [(q, q_ref, None)]],
            "bound": [(q, qmin, qmax), (qdot_max, None, None)], "rows": []}
    e0 = float((pose_d[1][:, 9:] - pose[1][:, 9:]).norm(dim=1).max())
    for cycle in range(200):
        fk(); st.update(leaf)
        if front_end == "eHQP":
            st.solve_ehqp(B)
        elif front_end == "nHQP":
            st.solve_nhqp(B)
        else:
            st.solve(B)
        q += st.dq[:B]
        if cycle in (0, 199):
            torch.cuda.synchronize()
            assert (st.status[:B] == 0).all()
    fk(); torch.cuda.synchronize()
    qh = q.cpu().numpy()
    if front_end in ("iHQP", "nHQP"):
        assert (qh[:, 6:] >= lo[6:] - 1e-9).all() and (qh[:, 6:] <= up[6:] + 1e-9).all()    # the URDF's limits held
    # (nHQP's default A / b regularisation lifts the small singular values of a level, nHQP.cpp:236-279: a slower last approach)
    assert float((pose_d[1][:, 9:] - pose[1][:, 9:]).norm(dim=1).max()) < (0.1 if front_end == "nHQP" else 0.05) * e0    # r_wrist reached its target
    for f in (2, 3):
        assert float((pose_d[f][:, 9:] - pose[f][:, 9:]).norm(dim=1).max()) < 5e-3           # the feet stayed
    assert float((com_d - com).norm(dim=1).max()) < 2e-3                                      # and so did the CoM
    o = pykin.forward(m, qh[5])
    assert np.abs(pose[1][5].cpu().numpy()[9:] - o["frame_p"][1]).max() < 1e-12


@pytest.mark.gpu
def test_pair_distances_kernel_matches_restatement(gpu_device):
    import torch
    m = kin.humanoid32_pairs(kin.humanoid32())
    K = kin.Kinematics(m, device=0)
    B, P = 300, len(m.pairs)
    rng = np.random.default_rng(15)
    q = rng.uniform(-1.0, 1.0, (B, m.n))
    dev = torch.device("cuda", 0)
    tq = torch.as_tensor(q, device=dev)
    Jd = torch.full((B, P + 2, m.n), 7.0, dtype=torch.float64, device=dev)
    dist = torch.zeros((B, P), dtype=torch.float64, device=dev)
    K.forward(tq, pair_dist=dist, pair_J=(Jd, 1))
    torch.cuda.synchronize()
    Jh, dh = Jd.cpu().numpy(), dist.cpu().numpy()
    for i in range(0, B, 13):
        d, J = pykin.pair_distances(m, q[i])
        assert np.abs(dh[i] - d).max() < 1e-14 and np.abs(Jh[i, 1:P + 1] - J).max() < 1e-13
    assert (Jh[:, 0] == 7.0).all() and (Jh[:, P + 1] == 7.0).all()


@pytest.mark.gpu
def test_closed_loop_self_collision_avoidance_on_device(gpu_device):
    """SURVEY 8f-3 end to end: q -> capsule-pair distances and rows J_d (osot_kinematics) -> CollisionAvoidance rows
    (osot_stack_update, CollisionAvoidance.cpp:119-147) -> dq (osot_ihqp_solve) -> q += dq.  Both hands are sent to the
    SAME point: without the constraint they end up inside each other, with it every pair stops at the threshold."""
    import torch
    from opensot_amd.plan import Rows, subtask
    from opensot_amd.solver import BatchedStack
    m = kin.humanoid32_pairs(kin.humanoid32())
    n, B, P = m.n, 64, len(m.pairs)
    dev = torch.device("cuda", 0)
    f64 = dict(dtype=torch.float64, device=dev)
    rng = np.random.default_rng(2)
    q0 = np.zeros((B, n))
    q0[:, [m.names.index(s + "Elbj") for s in "RL"]] = -0.9
    q0[:, m.names.index("RShLat")] = -0.35; q0[:, m.names.index("LShLat")] = 0.35
    q0 += rng.normal(0.0, 0.01, (B, n))
    d_min = 0.02

    def run(with_constraint):
        wrist = lambda nm: subtask(Task(abi.TASK_CARTESIAN, 6, lam=0.1, name=nm), [0, 1, 2])      # position only
        levels = [[Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="l_sole"), Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="r_sole")],
                  [wrist("l_wrist"), wrist("r_wrist")],
                  [Task(abi.TASK_POSTURAL, n, lam=0.01, name="postural")]]
        bounds = [Bound(abi.BOUND_VELOCITY_LIMITS, dT=0.01, name="vl")]
        rows = [Rows(abi.ROWS_COLLISION, P, d_threshold=d_min, detection_threshold=0.0, bound_scaling=0.2, name="self_collision")] if with_constraint else []
        plan = StackPlan(n=n, levels=levels, bounds=bounds, rowblocks=rows, eps_abs=eps_abs_from_factor(1e6))
        st = BatchedStack(plan, B, device=0, want_levels=False)
        K = kin.Kinematics(m, device=0)
        q = torch.as_tensor(q0, **f64).contiguous()
        pose = [torch.zeros((B, 12), **f64) for _ in range(4)]
        Jd = torch.zeros((B, P, n), **f64); dist = torch.zeros((B, P), **f64)
        Jw = torch.zeros((B, 12, n), **f64)

        def fk():
            K.forward(q, frame_pose={f: pose[f] for f in range(4)}, frame_J={2: (st.A[0], 0), 3: (st.A[0], 6), 0: (Jw, 0), 1: (Jw, 6)},
                      pair_dist=dist, pair_J=(Jd, 0))
            st.A[1][:B, 0:3].copy_(Jw[:, 0:3]); st.A[1][:B, 3:6].copy_(Jw[:, 6:9])    # the sub-tasks keep the linear rows
        fk(); torch.cuda.synchronize()
        pose_d = [p.clone() for p in pose]
        mid = 0.5 * (pose[0][:, 9:] + pose[1][:, 9:])
        pose_d[0][:, 9:] = mid; pose_d[1][:, 9:] = mid
        q_ref = q.clone()
        qdot_max = torch.full((B, n), 2.0, **f64)
        leaf = {"B": B, "task": [[(pose[2], pose_d[2], None), (pose[3], pose_d[3], None)],
                                 [(pose[0], pose_d[0], None), (pose[1], pose_d[1], None)], [(q, q_ref, None)]],
                "bound": [(qdot_max, None, None)], "rows": [(Jd, dist, None)] if with_constraint else []}
        dmin_seen = np.inf
        for cycle in range(400):
            fk()
            st.update(leaf); st.solve(B)
            q += st.dq[:B]
            if cycle % 20 == 0:
                torch.cuda.synchronize()
                assert (st.status[:B] == 0).all()
                dmin_seen = min(dmin_seen, float(dist.min()))
        fk(); torch.cuda.synchronize()
        gap = float((pose[0][:, 9:] - pose[1][:, 9:]).norm(dim=1).max())
        return float(dist.min()), dmin_seen, dist.cpu().numpy(), q.cpu().numpy(), gap

    free_end, free_seen, _, _, free_gap = run(False)
    safe_end, safe_seen, dist, q, safe_gap = run(True)
    assert free_gap < 5e-3 and free_end < -0.05           # unconstrained: the wrists meet, the hand spheres overlap
    assert safe_end > d_min - 2e-3 and safe_seen > d_min - 2e-3     # constrained: every pair stays at the threshold (first-order rows)
    assert safe_gap > 0.05                                # ... so the wrists cannot meet
    d, _ = pykin.pair_distances(m, q[5])
    assert np.abs(d - dist[5]).max() < 1e-12


@pytest.mark.gpu
def test_closed_loop_coman_ik_stack_with_contact_constraints(oracle, gpu_device):
    """examples/cpp/coman_ik.cpp:437-442, the reference example's own three-level stack,
        (com / (0.1*l_wrist + r_wrist) / postural) << joint_limits << vel_limits << (l_sole + r_sole),
    with the feet as constraints::TaskToConstraint rows (TaskToConstraint.cpp:59-68): their Jacobians go from the
    kinematics kernel straight into C, their bounds b +- 0 come out of the update kernel; first cycle against the
    oracle's assembly and the witnesses, then the closed loop."""
    import torch
    from opensot_amd.plan import Rows
    from opensot_amd.solver import BatchedStack
    m = kin.humanoid32()
    n, B = m.n, 96
    dev = torch.device("cuda", 0)
    f64 = dict(dtype=torch.float64, device=dev)
    rng = np.random.default_rng(19)
    q0 = np.zeros((B, n))
    q0[:, [m.names.index(s + "KneeSag") for s in "RL"]] = 0.5
    q0[:, [m.names.index(s + "HipSag") for s in "RL"]] = -0.25
    q0[:, [m.names.index(s + "AnkSag") for s in "RL"]] = -0.25
    q0[:, [m.names.index(s + "Elbj") for s in "RL"]] = -0.6
    q0 += rng.normal(0.0, 0.02, (B, n))
    levels = [[Task(abi.TASK_COM, 3, lam=0.1, name="com")],
              [Task(abi.TASK_CARTESIAN, 6, weight=0.1, lam=0.1, name="l_wrist"), Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="r_wrist")],
              [Task(abi.TASK_POSTURAL, n, lam=0.01, name="postural")]]
    bounds = [Bound(abi.BOUND_JOINT_LIMITS, scaling=1.0, name="jl"), Bound(abi.BOUND_VELOCITY_LIMITS, dT=0.01, name="vl")]
    rows = [Rows(abi.ROWS_TASK_CARTESIAN, 6, lam=0.1, name="l_sole"), Rows(abi.ROWS_TASK_CARTESIAN, 6, lam=0.1, name="r_sole")]
    plan = StackPlan(n=n, levels=levels, bounds=bounds, rowblocks=rows, eps_abs=eps_abs_from_factor(1e6))
    st = BatchedStack(plan, B, device=0)
    K = kin.Kinematics(m, device=0)
    q = torch.as_tensor(q0, **f64).contiguous()
    pose = [torch.zeros((B, 12), **f64) for _ in range(4)]
    com = torch.zeros((B, 3), **f64)

    def fk():
        K.forward(q, frame_pose={f: pose[f] for f in range(4)},
                  frame_J={0: (st.A[1], 0), 1: (st.A[1], 6), 2: (st.C, 0), 3: (st.C, 6)}, com=com, com_J=(st.A[0], 0))
    fk(); torch.cuda.synchronize()
    pose_d = [p.clone() for p in pose]
    pose_d[0][:, 9:] += torch.as_tensor([0.04, 0.03, 0.03], **f64)
    pose_d[1][:, 9:] += torch.as_tensor([0.04, -0.03, 0.03], **f64)
    pose_d[2][:, 9:] += torch.as_tensor(rng.normal(0, 2e-4, (B, 3)), **f64)      # the soles start a fraction of a mm off
    com_d = com.clone(); com_d[:, 0] += 0.02
    qmin = torch.full((B, n), -2.5, **f64); qmax = torch.full((B, n), 2.5, **f64)
    qdot_max = torch.full((B, n), 2.0, **f64)
    q_ref = q.clone()
    leaf = {"B": B, "task": [[(com, com_d, None)], [(pose[0], pose_d[0], None), (pose[1], pose_d[1], None)], [(q, q_ref, None)]],
            "bound": [(q, qmin, qmax), (qdot_max, None, None)], "rows": [(pose[2], pose_d[2], None), (pose[3], pose_d[3], None)]}
    st.update(leaf); st.solve(B); torch.cuda.synchronize()
    # first cycle: the update kernel's rows against the oracle's assembly, the cascade against the witnesses
    h = lambda t: None if t is None else t.cpu().numpy()
    np_leaf = {"B": B, "A": [h(st.A[0]), h(st.A[1]), None],
               "task": [[tuple(h(x) for x in t) for t in lev] for lev in leaf["task"]],
               "bound": [tuple(h(x) for x in t) for t in leaf["bound"]], "rows": [tuple(h(x) for x in t) for t in leaf["rows"]],
               "C": [h(st.C[:, 0:6]), h(st.C[:, 6:12])]}
    asm = oracle.assemble(plan, np_leaf)
    np.testing.assert_allclose(h(st.lo), asm["lo"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(h(st.up), asm["up"], rtol=0, atol=1e-15)
    assert (asm["lo"] == asm["up"]).all() and np.abs(asm["lo"]).max() > 1e-6       # equalities with a non-trivial right-hand side
    dq = h(st.dq[:B])
    assert (h(st.status[:B]) == 0).all()
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
    okr = ref["status"] == 1
    assert np.abs(dq[okr] - ref["dq"][okr]).max(initial=0.0) < 1e-9
    wit = {"eiQuadProg": ref}
    if oracle.ref_available():
        rq = wit["qpOASES"] = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1)
        rx = wit["qpOASES exact"] = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1, termination_tolerance=10 * 2.221e-16)
        e = np.minimum(np.where(rq["status"] == 1, np.abs(dq - rq["dq"]).max(axis=1), np.inf),
                       np.where(rx["status"] == 1, np.abs(dq - rx["dq"]).max(axis=1), np.inf))
        assert e[np.isfinite(e)].max(initial=0.0) < 1e-6
    from helpers import judge_remainder
    judge_remainder(asm, dq, wit, label="coman_ik stack with contact constraints")    # (no instance goes unjudged)
    # the feet rows hold exactly: J_sole dq = b_sole
    Cs = h(st.C[:B]); los = h(st.lo[:B])
    assert np.abs(np.einsum("bri,bi->br", Cs, dq) - los).max() < 1e-10
    q += st.dq[:B]
    for cycle in range(299):
        fk(); st.update(leaf); st.solve(B)
        q += st.dq[:B]
    fk(); torch.cuda.synchronize()
    assert (st.status[:B] == 0).all()
    err = [float((pose_d[f][:, 9:] - pose[f][:, 9:]).norm(dim=1).max()) for f in range(4)]
    assert err[1] < 3e-3 and err[0] < 2e-2           # r_wrist converged, l_wrist (weight 0.1, same level) follows
    assert err[2] < 5e-3 and err[3] < 5e-3           # the feet are CONSTRAINTS here: velocity-level rows, lambda = 0.1 on the drift
    assert float((com_d - com).norm(dim=1).max()) < 1e-3


# ---- environment shapes and boxes (addCollisionShape / moveCollisionShape / setLinksVsEnvironment) ------------
def _humanoid_with_environment():
    """the 32-DoF humanoid's self-collision pairs plus its hands / forearms against a static world: a box (table edge in
    front of the robot), a sphere and a capsule (a post), and a rotated box carried by the TORSO link"""
    m = kin.humanoid32_pairs(kin.humanoid32())
    m.self_pairs = m.pairs[:4]; m.pairs = m.pairs[:4]; m.pair_names = m.pair_names[:4]
    Rz = lambda a: np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    Rx = lambda a: np.array([[1.0, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    assert m.add_collision_shape("table", "world", ("box", (0.30, 0.80, 0.04)), (Rz(0.3) @ Rx(0.1), (0.33, 0.05, 0.10)))
    assert m.add_collision_shape("ball", "world", ("sphere", 0.08), (np.eye(3), (0.25, -0.30, 0.30)))
    assert m.add_collision_shape("post", "world", ("capsule", 0.60, 0.03), (Rx(0.2), (0.20, 0.35, 0.10)))
    assert m.add_collision_shape("chest_plate", "WaistYaw", ("box", (0.06, 0.20, 0.16)), (Rz(0.1), (0.11, 0.0, 0.14)))
    assert not m.add_collision_shape("ball", "world", ("sphere", 0.1))           # the name is taken
    assert not m.add_collision_shape("x", "no_such_link", ("sphere", 0.1))
    m.set_links_vs_environment(["Lhand", "Rhand", "Lforearm", "Rforearm"])
    assert len(m.pairs) == 4 + 4 * 4 and m.pair_names[4] == ("Lhand", "table")
    return m


def test_environment_pairs_match_finite_differences():
    """link-vs-environment pairs of the restatement (sphere, capsule and BOX, world-fixed and link-carried): the rows against
    central differences of the distances; a known-answer segment / box case; moveCollisionShape moves the distances"""
    m = _humanoid_with_environment()
    rng = np.random.default_rng(21)
    h = 1e-6
    for _ in range(3):
        q = rng.uniform(-0.6, 0.6, m.n)
        d, J = pykin.pair_distances(m, q)
        Jfd = np.zeros_like(J)
        for j in range(m.n):
            e = np.zeros(m.n); e[j] = h
            Jfd[:, j] = (pykin.pair_distances(m, q + e)[0] - pykin.pair_distances(m, q - e)[0]) / (2 * h)
        ok = d + np.array([p[3] + p[7] for p in m.pairs]) > 1e-3      # (a segment inside a box has no normal: zero row)
        assert ok.sum() >= 18 and np.abs(J[ok] - Jfd[ok]).max() < 2e-7
    # known answers in the box frame: a point above a face, beside an edge, and a segment that passes over a corner
    half = np.array([1.0, 2.0, 3.0])
    ca, cb = pykin.segment_box_closest(np.array([0.5, 0.5, 5.0]), np.array([0.5, 0.5, 5.0]), half)
    assert np.allclose(cb, [0.5, 0.5, 3.0]) and abs(np.linalg.norm(ca - cb) - 2.0) < 1e-15
    ca, cb = pykin.segment_box_closest(np.array([4.0, 6.0, 0.0]), np.array([4.0, 6.0, 0.0]), half)
    assert np.allclose(cb, [1.0, 2.0, 0.0]) and abs(np.linalg.norm(ca - cb) - 5.0) < 1e-15
    ca, cb = pykin.segment_box_closest(np.array([3.0, 0.0, 4.0]), np.array([-1.0, 4.0, 4.0]), half)
    assert np.allclose(cb, [1.0, 2.0, 3.0]) and abs(np.linalg.norm(ca - cb) - 1.0) < 1e-12
    q = rng.uniform(-0.4, 0.4, m.n)
    d0, _ = pykin.pair_distances(m, q)
    assert m.move_collision_shape("ball", (np.eye(3), (0.25, -0.30, 0.80))) and not m.move_collision_shape("chest_plate", (np.eye(3), (0, 0, 0)))
    d1, _ = pykin.pair_distances(m, q)
    moved = [k for k, nm in enumerate(m.pair_names) if nm[1] == "ball"]
    same = [k for k in range(len(m.pairs)) if k not in moved]
    assert np.abs(d1[moved] - d0[moved]).min() > 1e-2 and np.abs(d1[same] - d0[same]).max() == 0.0


def test_emulated_environment_pairs_match_restatement():
    from helpers import emu_kinematics
    m = _humanoid_with_environment()
    rng = np.random.default_rng(22)
    B = 5
    q = rng.uniform(-0.8, 0.8, (B, m.n))
    P = len(m.pairs)
    poses, J, com, pd, pJ = emu_kinematics(m, q)
    for i in range(B):
        d, Jd = pykin.pair_distances(m, q[i])
        assert np.abs(pd[i] - d).max() < 1e-12 and np.abs(pJ[i, :P] - Jd).max() < 1e-12
    # one world per instance (a [B][n_env][12] pose array): instance i sees the ball shifted by 0.1 i
    env = np.repeat(m.env_pose_array()[None], B, axis=0)
    ball = [e["name"] for e in m.env_shapes if e["link"] == "world"].index("ball")
    env[:, ball, 11] += 0.1 * np.arange(B)
    poses, J, com, pd, pJ = emu_kinematics(m, q, env_pose=env)
    for i in range(B):
        d, Jd = pykin.pair_distances(m, q[i], env_pose=env[i])
        assert np.abs(pd[i] - d).max() < 1e-12 and np.abs(pJ[i, :P] - Jd).max() < 1e-12


@pytest.mark.gpu
def test_environment_pairs_kernel_matches_restatement(gpu_device):
    import torch
    m = _humanoid_with_environment()
    K = kin.Kinematics(m, device=0)
    B, P = 200, len(m.pairs)
    rng = np.random.default_rng(23)
    q = rng.uniform(-0.8, 0.8, (B, m.n))
    dev = torch.device("cuda", 0)
    tq = torch.as_tensor(q, device=dev)
    Jd = torch.full((B, P + 1, m.n), 7.0, dtype=torch.float64, device=dev)
    dist = torch.zeros((B, P), dtype=torch.float64, device=dev)
    K.forward(tq, pair_dist=dist, pair_J=(Jd, 0))
    torch.cuda.synchronize()
    Jh, dh = Jd.cpu().numpy(), dist.cpu().numpy()
    for i in range(0, B, 9):
        d, J = pykin.pair_distances(m, q[i])
        assert np.abs(dh[i] - d).max() < 1e-12 and np.abs(Jh[i, :P] - J).max() < 1e-12
    assert (Jh[:, P] == 7.0).all()
    # moveCollisionShape: the new pose is a runtime input of the next launch
    assert m.move_collision_shape("table", (np.eye(3), (0.40, 0.0, 0.25)))
    K.forward(tq, pair_dist=dist, pair_J=(Jd, 0))
    torch.cuda.synchronize()
    d, J = pykin.pair_distances(m, q[7])
    assert np.abs(dist[7].cpu().numpy() - d).max() < 1e-12 and np.abs(Jd[7, :P].cpu().numpy() - J).max() < 1e-12


@pytest.mark.gpu
def test_closed_loop_hand_stops_at_a_world_box(gpu_device):
    """SURVEY 8f-3 end to end with an ENVIRONMENT shape: the right wrist is sent to a point behind a world box
    (addCollisionShape + setLinksVsEnvironment); q -> distances and rows (osot_kinematics) -> CollisionAvoidance rows
    (osot_stack_update, CollisionAvoidance.cpp:119-147) -> dq -> q += dq.  Without the constraint the hand ends inside the
    box, with it the hand sphere stops at the distance threshold from the box surface."""
    import torch
    from opensot_amd.plan import Rows, subtask
    from opensot_amd.solver import BatchedStack
    dev = torch.device("cuda", 0)
    f64 = dict(dtype=torch.float64, device=dev)
    B, d_min = 32, 0.02
    rng = np.random.default_rng(3)

    def run(with_constraint):
        m = kin.humanoid32_pairs(kin.humanoid32())
        m.self_pairs = []; m.pairs = []; m.pair_names = []
        n = m.n
        q0 = np.zeros((B, n))
        q0[:, [m.names.index(s + "Elbj") for s in "RL"]] = -0.9
        q0 += rng.normal(0.0, 0.01, (B, n))
        fk0 = pykin.forward(m, q0[0])
        hand = fk0["frame_p"][m.frame_index("r_wrist")]
        # a wall 12 cm in front of the hand, the target 10 cm behind its near face
        wall_c = hand + np.array([0.12 + 0.05, 0.0, 0.0])
        assert m.add_collision_shape("wall", "world", ("box", (0.10, 0.60, 0.60)), (np.eye(3), wall_c))
        m.set_links_vs_environment(["Rhand", "Rforearm"])
        P = len(m.pairs)
        wrist = lambda nm: subtask(Task(abi.TASK_CARTESIAN, 6, lam=0.1, name=nm), [0, 1, 2])
        levels = [[Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="l_sole"), Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="r_sole")],
                  [wrist("r_wrist")], [Task(abi.TASK_POSTURAL, n, lam=0.01, name="postural")]]
        rows = [Rows(abi.ROWS_COLLISION, P, d_threshold=d_min, detection_threshold=0.0, bound_scaling=0.2, name="env")] if with_constraint else []
        plan = StackPlan(n=n, levels=levels, bounds=[Bound(abi.BOUND_VELOCITY_LIMITS, dT=0.01, name="vl")], rowblocks=rows,
                         eps_abs=eps_abs_from_factor(1e6))
        st = BatchedStack(plan, B, device=0, want_levels=False)
        K = kin.Kinematics(m, device=0)
        q = torch.as_tensor(q0, **f64).contiguous()
        pose = [torch.zeros((B, 12), **f64) for _ in range(4)]
        Jd = torch.zeros((B, P, n), **f64); dist = torch.zeros((B, P), **f64)
        Jw = torch.zeros((B, 6, n), **f64)
        fr = m.frame_index

        def fk():
            K.forward(q, frame_pose={f: pose[f] for f in range(4)},
                      frame_J={fr("l_sole"): (st.A[0], 0), fr("r_sole"): (st.A[0], 6), fr("r_wrist"): (Jw, 0)}, pair_dist=dist, pair_J=(Jd, 0))
            st.A[1][:B, 0:3].copy_(Jw[:, 0:3])
        fk(); torch.cuda.synchronize()
        pose_d = [p.clone() for p in pose]
        pose_d[fr("r_wrist")][:, 9] += 0.22          # 10 cm behind the wall's near face
        q_ref = q.clone()
        leaf = {"B": B, "task": [[(pose[fr("l_sole")], pose_d[fr("l_sole")], None), (pose[fr("r_sole")], pose_d[fr("r_sole")], None)],
                                 [(pose[fr("r_wrist")], pose_d[fr("r_wrist")], None)], [(q, q_ref, None)]],
                "bound": [(torch.full((B, n), 2.0, **f64), None, None)], "rows": [(Jd, dist, None)] if with_constraint else []}
        for cycle in range(300):
            fk()
            st.update(leaf); st.solve(B)
            q += st.dq[:B]
        fk(); torch.cuda.synchronize()
        assert (st.status[:B] == 0).all()
        return dist.cpu().numpy(), (pose_d[fr("r_wrist")][:, 9:] - pose[fr("r_wrist")][:, 9:]).norm(dim=1).cpu().numpy(), q.cpu().numpy(), m

    free_d, free_err, _, _ = run(False)
    safe_d, safe_err, q, m = run(True)
    hand_pair = m.pair_names.index(("Rhand", "wall"))
    assert free_err.max() < 5e-3 and free_d[:, hand_pair].max() < 0.0      # unconstrained: the wrist reaches the target INSIDE the wall
    assert safe_d.min() > d_min - 2e-3                                     # constrained: nothing comes closer than the threshold
    assert np.abs(safe_d[:, hand_pair] - d_min).max() < 5e-3 and safe_err.min() > 0.05    # the hand rests against the wall
    d, _ = pykin.pair_distances(m, q[3])
    assert np.abs(d - safe_d[3]).max() < 1e-12
