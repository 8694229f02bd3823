"""Host side of the batched kinematics producer (include/osot_mi355x.h: osot_kin_*): model description,
a 32-DoF humanoid tree for the benchmarks/tests, and the ctypes plumbing that points the producer at the row
ranges of a BatchedStack's A_k buffers.  Plumbing only: the arithmetic is in csrc/osot_kin.h."""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np
import torch

from . import abi


@dataclass
class KinModel:
    parent: list
    jtype: list
    axis: np.ndarray            # [n][3]
    R0: np.ndarray              # [n][3][3]
    p0: np.ndarray              # [n][3]
    mass: np.ndarray            # [n]
    com: np.ndarray             # [n][3]
    names: list = field(default_factory=list)
    frames: list = field(default_factory=list)     # (name, joint index, R[3][3], p[3])
    # self-collision pairs: (joint a, a0[3], a1[3], radius a, joint b, b0[3], b1[3], radius b): two capsules (axis end
    # points in the joint frames; a0 == a1: a sphere)
    pairs: list = field(default_factory=list)
    # per-frame options of the Jacobian the producer writes: frame index -> True for a BODY Jacobian Ad(R_f') J
    # (Cartesian::setIsBodyJacobian, Cartesian.cpp:93-100); frame index -> list of ACTIVE joint columns, the others are
    # written as zero (Task::setActiveJointsMask, Task.h:129-139); the same for the CoM Jacobian
    frame_body: dict = field(default_factory=dict)
    frame_active_joints: dict = field(default_factory=dict)
    # frame index -> index (or name) of the frame that is its BASE LINK: pose and Jacobian of the frame relative to that frame, in its
    # coordinates (velocity::Cartesian with base_link != "world", Cartesian.cpp:73-81); absent = the world
    frame_base: dict = field(default_factory=dict)
    com_active_joints: list = None
    # environment shapes and link-vs-environment pairs (CollisionAvoidance::addCollisionShape / moveCollisionShape /
    # setLinksVsEnvironment, CollisionAvoidance.h:115-144).  link_shapes: link name -> (joint, a0, a1, radius), the capsule
    # a link is checked with; env_shapes: dicts (name, link, kind, data ..., pose); self_pairs: the link-link pairs that
    # stay in front of the generated link-environment pairs
    link_shapes: dict = field(default_factory=dict)
    env_shapes: list = field(default_factory=list)
    self_pairs: list = None
    env_links: list = field(default_factory=list)

    # ---- the reference's three calls ------------------------------------------------------------------------------
    def add_collision_shape(self, name, link, shape, link_T_shape=None):
        """CollisionAvoidance::addCollisionShape(name, link, shape, link_T_shape) (CollisionAvoidance.cpp:166-173).  link:
        "world" (an environment shape, movable afterwards) or a joint / link name of the model.  shape: ("sphere", r),
        ("capsule", length, r) (axis = z of the shape frame, centred, like XBot::Collision::Shape::Capsule) or
        ("box", (sx, sy, sz)) (full sizes).  link_T_shape: (R[3][3], p[3]), default identity.  False if the name is taken
        or the link unknown (the reference returns false as well)."""
        if any(e["name"] == name for e in self.env_shapes):
            return False
        if link != "world" and link not in self.names:
            return False
        R, p = (np.eye(3), np.zeros(3)) if link_T_shape is None else (np.asarray(link_T_shape[0], float).reshape(3, 3),
                                                                      np.asarray(link_T_shape[1], float).reshape(3))
        kind = shape[0]
        if kind == "sphere":
            e = dict(kind="capsule", a0=np.zeros(3), a1=np.zeros(3), r=float(shape[1]))
        elif kind == "capsule":
            L, r = float(shape[1]), float(shape[2])
            e = dict(kind="capsule", a0=np.array([0, 0, -0.5 * L]), a1=np.array([0, 0, 0.5 * L]), r=r)
        elif kind == "box":
            e = dict(kind="box", half=0.5 * np.asarray(shape[1], float).reshape(3), r=0.0)
        else:
            raise ValueError("shape must be a sphere, a capsule or a box (meshes / box-box pairs are not offered)")
        e.update(name=name, link=link, R=R, p=p)
        self.env_shapes.append(e)
        self._rebuild_pairs()
        return True

    def move_collision_shape(self, name, new_pose):
        """CollisionAvoidance::moveCollisionShape(id, new_pose) (CollisionAvoidance.cpp:175-178): new world pose (R, p) of
        an environment shape; takes effect at the next Kinematics.forward (the pose is a runtime input of the kernel)"""
        for e in self.env_shapes:
            if e["name"] == name and e["link"] == "world":
                e["R"] = np.asarray(new_pose[0], float).reshape(3, 3); e["p"] = np.asarray(new_pose[1], float).reshape(3)
                return True
        return False

    def set_links_vs_environment(self, links):
        """CollisionAvoidance::setLinksVsEnvironment(links): the links (names with an entry in link_shapes) that are
        checked against every environment shape; the pair list becomes self_pairs + links x environment"""
        missing = [l for l in links if l not in self.link_shapes]
        if missing:
            raise KeyError(f"no collision capsule known for {missing}")
        self.env_links = list(links)
        self._rebuild_pairs()

    def env_pose_array(self):
        """[n_env][12] world_T_shape of the WORLD-carried shapes, in env-slot order (osot_kin_batch.env_pose)"""
        W = [e for e in self.env_shapes if e["link"] == "world"]
        out = np.zeros((len(W), 12))
        for k, e in enumerate(W):
            out[k, :9] = e["R"].reshape(9); out[k, 9:] = e["p"]
        return out

    def _rebuild_pairs(self):
        if self.self_pairs is None:
            self.self_pairs = list(self.pairs)
        pairs = list(self.self_pairs)
        names = list(getattr(self, "pair_names", [])[:len(self.self_pairs)])
        world = [e for e in self.env_shapes if e["link"] == "world"]
        for ln in self.env_links:
            ja, a0, a1, ra = self.link_shapes[ln]
            for e in self.env_shapes:
                if e["link"] == "world":
                    slot = [w["name"] for w in world].index(e["name"])
                    jb, env, R, p = -1, slot + 1, np.eye(3), np.zeros(3)     # runtime pose; data in the shape frame
                else:
                    jb, env, R, p = self.names.index(e["link"]), 0, e["R"], e["p"]
                    if jb == ja:
                        continue
                if e["kind"] == "box":
                    pairs.append((ja, a0, a1, ra, jb, (0, 0, 0), (0, 0, 0), e["r"],
                                  dict(kind=abi.SHAPE_BOX, env=env, half=e["half"], R=R, p=p)))
                else:
                    b0, b1 = R @ e["a0"] + p, R @ e["a1"] + p
                    pairs.append((ja, a0, a1, ra, jb, b0, b1, e["r"], dict(kind=abi.SHAPE_CAPSULE, env=env)))
                names.append((ln, e["name"]))
        if len(pairs) > abi.KIN_MAX_PAIRS:
            raise ValueError(f"{len(pairs)} collision pairs, the producer holds {abi.KIN_MAX_PAIRS}")
        self.pairs = pairs
        self.pair_names = names

    @property
    def n(self):
        return len(self.parent)

    def frame_index(self, name):
        return [f[0] for f in self.frames].index(name)

    def desc(self):
        d = abi.KinDesc()
        d.n = self.n
        for j in range(self.n):
            d.parent[j] = int(self.parent[j]); d.type[j] = int(self.jtype[j]); d.mass[j] = float(self.mass[j])
            for i in range(3):
                d.axis[j][i] = float(self.axis[j][i]); d.p0[j][i] = float(self.p0[j][i]); d.com[j][i] = float(self.com[j][i])
            for i in range(9):
                d.R0[j][i] = float(self.R0[j].reshape(9)[i])
        d.n_frames = len(self.frames)
        for f, (_, jf, R, p) in enumerate(self.frames):
            d.frame_joint[f] = int(jf)
            for i in range(9):
                d.frame_R[f][i] = float(np.asarray(R).reshape(9)[i])
            for i in range(3):
                d.frame_p[f][i] = float(p[i])
        mask = lambda cols: sum(1 << int(c) for c in cols)
        for f in range(len(self.frames)):
            d.frame_body[f] = 1 if self.frame_body.get(f) else 0
            d.frame_col_mask[f] = mask(self.frame_active_joints[f]) if f in self.frame_active_joints else 0
        for f, g in self.frame_base.items():
            g = self.frame_index(g) if isinstance(g, str) else int(g)
            if g == f or not 0 <= g < len(self.frames):
                raise ValueError(f"frame {f}: its base link frame must be another frame of the model")
            d.frame_base[f] = g + 1
        d.com_col_mask = mask(self.com_active_joints) if self.com_active_joints is not None else 0
        d.n_pairs = len(self.pairs)
        d.n_env = sum(1 for e in self.env_shapes if e["link"] == "world")
        for k, pr in enumerate(self.pairs):
            ja, a0, a1, ra, jb, b0, b1, rb = pr[:8]
            ex = pr[8] if len(pr) > 8 else {}
            d.pair_joint[k][0], d.pair_joint[k][1] = int(ja), int(jb)
            d.pair_radius[k][0], d.pair_radius[k][1] = float(ra), float(rb)
            for i in range(3):
                d.pair_seg[k][0][i], d.pair_seg[k][0][3 + i] = float(a0[i]), float(a1[i])
                d.pair_seg[k][1][i], d.pair_seg[k][1][3 + i] = float(b0[i]), float(b1[i])
            d.pair_kind[k] = int(ex.get("kind", abi.SHAPE_CAPSULE))
            d.pair_env[k] = int(ex.get("env", 0))
            R = np.asarray(ex.get("R", np.eye(3)), float).reshape(9)
            pp = np.asarray(ex.get("p", np.zeros(3)), float)
            hf = np.asarray(ex.get("half", np.zeros(3)), float)
            for i in range(9):
                d.pair_shape_R[k][i] = float(R[i])
            for i in range(3):
                d.pair_shape_p[k][i] = float(pp[i]); d.pair_box[k][i] = float(hf[i])
        return d


def _rpy(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def humanoid32():
    """a 32-DoF floating-base humanoid tree: 6 virtual joints (x, y, z, roll, pitch, yaw), 3 waist, 2 x 6 leg,
    2 x 5 arm, 1 neck.  Link lengths / masses are COMAN-like round numbers (the reference robot,
    tests/robots/coman_floating_base, has 6 + 29 = 35 coordinates; BASELINE's configs are quoted for 32)."""
    P, R = abi.JOINT_PRISMATIC, abi.JOINT_REVOLUTE
    X, Y, Z = [1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]
    rows = []   # (name, parent name, type, axis, xyz, rpy, mass, com)

    def add(name, parent, t, ax, xyz=(0, 0, 0), rpy=(0, 0, 0), mass=0.0, com=(0, 0, 0)):
        rows.append((name, parent, t, ax, xyz, rpy, mass, com))
    add("base_x", None, P, X); add("base_y", "base_x", P, Y); add("base_z", "base_y", P, Z)
    add("base_roll", "base_z", R, X); add("base_pitch", "base_roll", R, Y)
    add("base_yaw", "base_pitch", R, Z, mass=4.0, com=(0.0, 0.0, 0.02))            # pelvis
    add("WaistLat", "base_yaw", R, X, xyz=(0.02, 0, 0.12), mass=0.8, com=(0, 0, 0.02))
    add("WaistSag", "WaistLat", R, Y, mass=0.8, com=(0, 0, 0.03))
    add("WaistYaw", "WaistSag", R, Z, xyz=(0, 0, 0.04), mass=6.0, com=(0, 0, 0.12))   # torso
    for s, sy in (("R", -1.0), ("L", 1.0)):
        add(s + "HipSag", "base_yaw", R, Y, xyz=(0, sy * 0.07, -0.03), mass=0.9, com=(0, 0, -0.02))
        add(s + "HipLat", s + "HipSag", R, X, mass=0.9, com=(0, 0, -0.03))
        add(s + "HipYaw", s + "HipLat", R, Z, xyz=(0, 0, -0.10), mass=1.6, com=(0, 0, -0.06))
        add(s + "KneeSag", s + "HipYaw", R, Y, xyz=(0, 0, -0.12), mass=1.3, com=(0, 0, -0.09))
        add(s + "AnkLat", s + "KneeSag", R, X, xyz=(0, 0, -0.20), mass=0.4, com=(0, 0, -0.01))
        add(s + "AnkSag", s + "AnkLat", R, Y, mass=0.7, com=(0.02, 0, -0.05))
    for s, sy in (("R", -1.0), ("L", 1.0)):
        add(s + "ShSag", "WaistYaw", R, Y, xyz=(0.0, sy * 0.15, 0.22), rpy=(sy * 0.17, 0, 0), mass=0.7, com=(0, sy * 0.02, 0))
        add(s + "ShLat", s + "ShSag", R, X, mass=0.7, com=(0, 0, -0.03))
        add(s + "ShYaw", s + "ShLat", R, Z, xyz=(0, 0, -0.05), mass=1.0, com=(0, 0, -0.08))
        add(s + "Elbj", s + "ShYaw", R, Y, xyz=(0.015, 0, -0.13), mass=0.9, com=(0, 0, -0.08))
        add(s + "Wrj", s + "Elbj", R, Z, xyz=(0, 0, -0.16), mass=0.5, com=(0, 0, -0.04))
    add("NeckYaw", "WaistYaw", R, Z, xyz=(0, 0, 0.28), mass=1.5, com=(0, 0, 0.08))
    names = [r[0] for r in rows]
    n = len(rows)
    assert n == 32, n
    m = KinModel(parent=[-1 if r[1] is None else names.index(r[1]) for r in rows], jtype=[r[2] for r in rows],
                 axis=np.array([r[3] for r in rows], dtype=float), R0=np.array([_rpy(*r[5]) for r in rows]),
                 p0=np.array([r[4] for r in rows], dtype=float), mass=np.array([r[6] for r in rows], dtype=float),
                 com=np.array([r[7] for r in rows], dtype=float), names=names)
    I = np.eye(3)
    m.frames = [("l_wrist", names.index("LWrj"), I, (0, 0, -0.08)), ("r_wrist", names.index("RWrj"), I, (0, 0, -0.08)),
                ("l_sole", names.index("LAnkSag"), I, (0.02, 0, -0.10)), ("r_sole", names.index("RAnkSag"), I, (0.02, 0, -0.10))]
    return m


def humanoid32_pairs(m):
    """capsules on the forearms, hands, torso, pelvis and thighs of humanoid32() and the pairs a self-collision
    constraint would watch (hands / forearms against torso, pelvis, thighs and each other)"""
    nm = m.names.index
    shape = m.link_shapes          # (kept on the model: setLinksVsEnvironment pairs these capsules with the environment)
    shape.update({"torso": (nm("WaistYaw"), (0, 0, 0.05), (0, 0, 0.22), 0.10), "pelvis": (nm("base_yaw"), (0, -0.05, 0.0), (0, 0.05, 0.0), 0.09),
                  "head": (nm("NeckYaw"), (0, 0, 0.10), (0, 0, 0.10), 0.09)})
    for s in "LR":
        shape[s + "forearm"] = (nm(s + "Elbj"), (0, 0, -0.02), (0, 0, -0.15), 0.035)
        shape[s + "hand"] = (nm(s + "Wrj"), (0, 0, -0.06), (0, 0, -0.06), 0.045)
        shape[s + "thigh"] = (nm(s + "HipYaw"), (0, 0, 0.0), (0, 0, -0.12), 0.055)
        shape[s + "shin"] = (nm(s + "KneeSag"), (0, 0, -0.02), (0, 0, -0.19), 0.045)
    names = [("Lhand", "torso"), ("Rhand", "torso"), ("Lforearm", "torso"), ("Rforearm", "torso"), ("Lhand", "pelvis"),
             ("Rhand", "pelvis"), ("Lhand", "Lthigh"), ("Rhand", "Rthigh"), ("Lhand", "Rhand"), ("Lforearm", "Rforearm"),
             ("Lhand", "Rforearm"), ("Rhand", "Lforearm"), ("Lhand", "head"), ("Rhand", "head"), ("Lshin", "Rshin"), ("Lthigh", "Rthigh")]
    m.pairs = [shape[a] + shape[b] for a, b in names]
    m.pair_names = names
    m.self_pairs = list(m.pairs)
    if m.env_links:
        m._rebuild_pairs()
    return m


def from_json(path):
    """a tree fixture as written by tests/golden/make_coman_tree.py (the reference's COMAN, 6 virtual + 29 revolute
    joints): returns (KinModel, lower[n], upper[n]) -- the limits are +-inf for the virtual joints."""
    import json
    d = json.load(open(path))
    J = d["joints"]
    m = KinModel(parent=[j["parent"] for j in J], jtype=[j["type"] for j in J],
                 axis=np.array([j["axis"] for j in J], dtype=float), R0=np.array([j["R0"] for j in J], dtype=float).reshape(-1, 3, 3),
                 p0=np.array([j["p0"] for j in J], dtype=float), mass=np.array([j["mass"] for j in J], dtype=float),
                 com=np.array([j["com"] for j in J], dtype=float), names=[j["name"] for j in J])
    m.frames = [(f["name"], f["joint"], np.array(f["R"], dtype=float).reshape(3, 3), tuple(f["p"])) for f in d["frames"]]
    lo = np.array([-np.inf if j["lower"] is None else j["lower"] for j in J])
    up = np.array([np.inf if j["upper"] is None else j["upper"] for j in J])
    return m, lo, up


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class Kinematics:
    """osot_kin handle + the binding of its outputs to device buffers."""

    def __init__(self, model: KinModel, device=0):
        self.model = model
        self._lib = abi.lib()
        self._h = C.c_void_p()
        d = model.desc()
        abi.check(self._lib.osot_kin_create(C.byref(d), int(device), C.byref(self._h)), "osot_kin_create")
        self.device = torch.device("cuda", device)
        self._env_host = self._env_dev = self._env_keep = None

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                self._lib.osot_kin_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def forward(self, q, frame_pose=None, frame_J=None, com=None, com_J=None, pair_dist=None, pair_J=None, env_pose=None):
        """osot_kinematics for the batch `batch_args` describes, stream-ordered on torch's current stream"""
        kb = self.batch_args(q, frame_pose, frame_J, com, com_J, pair_dist, pair_J, env_pose)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        abi.check(self._lib.osot_kinematics(self._h, C.byref(kb), stream), "osot_kinematics")

    def batch_args(self, q, frame_pose=None, frame_J=None, com=None, com_J=None, pair_dist=None, pair_J=None, env_pose=None):
        """the osot_kin_batch of a call (pointers and strides only; the tensors must outlive its use).  q [B][n] (device).  frame_pose: {frame index: tensor [B][12]}; frame_J: {frame index: (A_k tensor [B][ma][n],
        first row)}; com: tensor [B][3]; com_J: (A_k tensor, first row).
        env_pose: poses of the environment shapes, a device tensor [n_env][12] (one world for all instances) or
        [B][n_env][12]; default: the model's own (add_collision_shape / move_collision_shape), uploaded when they change"""
        B, n = q.shape
        assert n == self.model.n and q.is_contiguous()
        kb = abi.KinBatch()
        kb.B = B
        kb.q = q.data_ptr()
        for f, t in (frame_pose or {}).items():
            assert t.shape[0] >= B and t.is_contiguous()
            kb.frame_pose[f] = t.data_ptr()
        for f, (A, row) in (frame_J or {}).items():
            assert A.is_contiguous() and A.shape[2] == n and row + 6 <= A.shape[1]
            kb.frame_J[f] = A.data_ptr() + 8 * row * n
            kb.frame_J_stride[f] = A.shape[1] * n
        if com is not None:
            kb.com = com.data_ptr()
        if com_J is not None:
            A, row = com_J
            assert A.is_contiguous() and A.shape[2] == n and row + 3 <= A.shape[1]
            kb.com_J = A.data_ptr() + 8 * row * n
            kb.com_J_stride = A.shape[1] * n
        if pair_dist is not None:      # [B][n_pairs]: the OSOT_ROWS_COLLISION leaf p1
            assert pair_dist.is_contiguous() and pair_dist.shape[1] == len(self.model.pairs)
            kb.pair_dist = pair_dist.data_ptr()
        if pair_J is not None:         # (tensor [B][rows][n], first row): the OSOT_ROWS_COLLISION leaf p0
            A, row = pair_J
            assert A.is_contiguous() and A.shape[2] == n and row + len(self.model.pairs) <= A.shape[1]
            kb.pair_J = A.data_ptr() + 8 * row * n
            kb.pair_J_stride = A.shape[1] * n
        n_env = sum(1 for e in self.model.env_shapes if e["link"] == "world")
        if n_env and (pair_dist is not None or pair_J is not None):
            if env_pose is None:
                host = self.model.env_pose_array()
                if self._env_host is None or not np.array_equal(host, self._env_host):
                    self._env_host = host.copy()
                    self._env_dev = torch.as_tensor(host, dtype=torch.float64).to(self.device)
                env_pose = self._env_dev
            assert env_pose.is_contiguous() and env_pose.shape[-2:] == (n_env, 12)
            kb.env_pose = env_pose.data_ptr()
            kb.env_pose_stride = 12 * n_env if env_pose.dim() == 3 else 0
            self._env_keep = env_pose
        return kb
