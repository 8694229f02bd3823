"""oracle/pykin.py -- TEST INFRASTRUCTURE ONLY: CPU restatement (numpy) of the batched kinematics producer.

What the reference's leaf tasks obtain from XBot::ModelInterface each cycle (velocity/Cartesian.cpp:73-81
getJacobian/getPose, velocity/CoM.cpp:59-74 getCOM/getCOMJacobian).  xbot2_interface is not vendored, so parity at
this boundary is UNPINNED against the reference; this restatement is checked against finite differences of its own
forward kinematics (tests/test_kinematics.py) and the HIP kernel is checked against it."""
import numpy as np


def _rodrigues(ax, q):
    x, y, z = ax
    c, s = np.cos(q), np.sin(q)
    v = 1.0 - c
    return np.array([[c + x * x * v, x * y * v - z * s, x * z * v + y * s],
                     [y * x * v + z * s, c + y * y * v, y * z * v - x * s],
                     [z * x * v - y * s, z * y * v + x * s, c + z * z * v]])


def forward(model, q):
    """one configuration q[n] -> dict(Rw [n][3][3], pw [n][3], frame_R, frame_p, com, J (per frame 6 x n), Jcom 3 x n)"""
    n = model.n
    Rw = np.zeros((n, 3, 3)); pw = np.zeros((n, 3))
    for j in range(n):
        if model.jtype[j] == 0:
            Rl = model.R0[j] @ _rodrigues(model.axis[j], q[j]); pl = model.p0[j]
        else:
            Rl = model.R0[j]; pl = model.p0[j] + model.R0[j] @ model.axis[j] * q[j]
        a = model.parent[j]
        if a < 0:
            Rw[j], pw[j] = Rl, pl
        else:
            Rw[j] = Rw[a] @ Rl; pw[j] = Rw[a] @ pl + pw[a]
    z = np.einsum("jab,jb->ja", Rw, model.axis)
    cw = np.einsum("jab,jb->ja", Rw, model.com) + pw
    M = model.mass.sum()
    anc = []
    for j in range(n):
        s = {j} | (anc[model.parent[j]] if model.parent[j] >= 0 else set())
        anc.append(s)
    out = {"Rw": Rw, "pw": pw, "com": (model.mass[:, None] * cw).sum(0) / M, "frame_R": [], "frame_p": [], "J": []}
    for (_, jf, Rf, pf) in model.frames:
        R = Rw[jf] @ np.asarray(Rf, dtype=float); p = pw[jf] + Rw[jf] @ np.asarray(pf, dtype=float)
        J = np.zeros((6, n))
        for j in anc[jf]:
            if model.jtype[j] == 0:
                J[:3, j] = np.cross(z[j], p - pw[j]); J[3:, j] = z[j]
            else:
                J[:3, j] = z[j]
        out["frame_R"].append(R); out["frame_p"].append(p); out["J"].append(J)
    Jc = np.zeros((3, n))
    for l in range(n):
        for j in anc[l]:
            if model.jtype[j] == 0:
                Jc[:, j] += model.mass[l] * np.cross(z[j], cw[l] - pw[j])
            else:
                Jc[:, j] += model.mass[l] * z[j]
    out["Jcom"] = Jc / M
    return out


def _skew(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def relative(fk, f, g):
    """frame f RELATIVE to the base frame g, from the WORLD quantities of forward(): what velocity::Cartesian with a base link
    asks the model for (Cartesian.cpp:75-76, 80-81: getRelativeJacobian(distal, base, A), getPose(distal, base, T)).
    Textbook route, on purpose not the kernel's: the twist of the distal frame relative to the base frame in base coordinates,
        v_rel = R_b'(v_d - v_b - w_b x (p_d - p_b)),   w_rel = R_b'(w_d - w_b),
    from the two world Jacobians.  Returns (R_rel, p_rel, J_rel 6 x n)."""
    Rd, pd, Jd = fk["frame_R"][f], fk["frame_p"][f], fk["J"][f]
    Rb, pb, Jb = fk["frame_R"][g], fk["frame_p"][g], fk["J"][g]
    lin = Jd[:3] - Jb[:3] + _skew(pd - pb) @ Jb[3:]
    ang = Jd[3:] - Jb[3:]
    return Rb.T @ Rd, Rb.T @ (pd - pb), np.concatenate([Rb.T @ lin, Rb.T @ ang], axis=0)


def closest_segment_points(p1, q1, p2, q2):
    """closest points of two segments (either may be a point): the standard clamped two-parameter minimisation"""
    d1, d2, r = q1 - p1, q2 - p2, p1 - p2
    a, e, f, c = d1 @ d1, d2 @ d2, d2 @ r, d1 @ r
    tiny = 1e-18
    cl = lambda v: min(1.0, max(0.0, v))
    s = t = 0.0
    if a <= tiny and e <= tiny:
        pass
    elif a <= tiny:
        t = cl(f / e)
    elif e <= tiny:
        s = cl(-c / a)
    else:
        b = d1 @ d2
        den = a * e - b * b
        s = cl((b * f - c * e) / den) if den > tiny * a * e else 0.0
        t = (b * s + f) / e
        if t < 0.0:
            t, s = 0.0, cl(-c / a)
        elif t > 1.0:
            t, s = 1.0, cl((b - c) / a)
    return p1 + s * d1, p2 + t * d2


def segment_box_closest(p0, p1, half):
    """closest points of the segment [p0, p1] and the box [-half, half] (box frame), by an INDEPENDENT route from the
    kernel's piecewise-quadratic pieces: dist^2 from P(t) to the box is convex in t and its derivative
    g(t) = 2 sum_i excess_i(t) v_i is continuous and monotone, so the parameter is the root of g (or an end point):
    bisection to the last bit"""
    half = np.asarray(half, float)
    v = p1 - p0
    P = lambda t: p0 + t * v
    g = lambda t: float(2.0 * ((P(t) - np.clip(P(t), -half, half)) @ v))
    if g(0.0) >= 0.0:
        t = 0.0
    elif g(1.0) <= 0.0:
        t = 1.0
    else:
        a, b = 0.0, 1.0
        for _ in range(200):
            m = 0.5 * (a + b)
            if g(m) < 0.0:
                a = m
            else:
                b = m
        t = 0.5 * (a + b)
    ca = P(t)
    return ca, np.clip(ca, -half, half)


def _carrier(model, fk, jb, ex, env_pose):
    """world pose (R, p) of the frame side b's data are expressed in: its link, the world, or an environment shape's
    runtime pose (moveCollisionShape)"""
    if jb >= 0:
        return fk["Rw"][jb], fk["pw"][jb]
    env = int(ex.get("env", 0)) - 1
    if env >= 0:
        E = np.asarray(env_pose, float).reshape(-1, 12)[env]
        return E[:9].reshape(3, 3), E[9:]
    return np.eye(3), np.zeros(3)


def pair_distances(model, q, fk=None, env_pose=None):
    """self-collision pairs (what CollisionAvoidance.cpp:96-118 obtains from the collision module): surface distances
    d[P] of the capsule pairs and the rows J_d[P][n] with delta d = J_d dq"""
    fk = forward(model, q) if fk is None else fk
    n = model.n
    Rw, pw = fk["Rw"], fk["pw"]
    z = np.einsum("jab,jb->ja", Rw, model.axis)
    anc = []
    for j in range(n):
        anc.append({j} | (anc[model.parent[j]] if model.parent[j] >= 0 else set()))
    d = np.zeros(len(model.pairs)); J = np.zeros((len(model.pairs), n))
    if env_pose is None and getattr(model, "env_shapes", None):
        env_pose = model.env_pose_array()
    for k, pr in enumerate(model.pairs):
        ja, a0, a1, ra, jb, b0, b1, rb = pr[:8]
        ex = pr[8] if len(pr) > 8 else {}
        wa0, wa1 = Rw[ja] @ np.asarray(a0, float) + pw[ja], Rw[ja] @ np.asarray(a1, float) + pw[ja]
        Rc, pc = _carrier(model, fk, jb, ex, env_pose)
        if ex.get("kind", 0) == 1:      # box: carrier o link_T_shape is the box frame (CollisionAvoidance.h:115-119)
            Rb = Rc @ np.asarray(ex["R"], float).reshape(3, 3); pb = pc + Rc @ np.asarray(ex["p"], float)
            la, lb = segment_box_closest(Rb.T @ (wa0 - pb), Rb.T @ (wa1 - pb), ex["half"])
            ca, cb = Rb @ la + pb, Rb @ lb + pb
        else:
            wb0, wb1 = Rc @ np.asarray(b0, float) + pc, Rc @ np.asarray(b1, float) + pc
            ca, cb = closest_segment_points(wa0, wa1, wb0, wb1)
        dv = ca - cb
        ln = np.linalg.norm(dv)
        d[k] = ln - ra - rb
        if not ln > 1e-12:
            continue
        nn = dv / ln
        for j in range(n):
            va = (np.cross(z[j], ca - pw[j]) if model.jtype[j] == 0 else z[j]) if j in anc[ja] else np.zeros(3)
            vb = (np.cross(z[j], cb - pw[j]) if model.jtype[j] == 0 else z[j]) if (jb >= 0 and j in anc[jb]) else np.zeros(3)
            J[k, j] = nn @ (va - vb)
    return d, J
