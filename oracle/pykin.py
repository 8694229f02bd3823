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
