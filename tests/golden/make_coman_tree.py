"""Generates tests/golden/coman_tree.json from the reference's robot description
(/root/reference/tests/robots/coman_floating_base/coman_floating_base.urdf).  Run in the build container only.

The fixture is DATA: the kinematic tree reduced to what the batched kinematics producer needs -- for every moving joint
its parent, type, axis, the fixed transform from the parent joint frame, and the mass / centre of mass of everything
rigidly attached to it (links behind fixed joints are merged into their moving ancestor); the floating joint becomes
the usual chain x, y, z, roll, pitch, yaw of virtual joints; the four end-effector frames of examples/cpp/coman_ik.cpp
(l_wrist, r_wrist, l_sole, r_sole) with their fixed offsets.  Limits come along for the joint-limit bounds."""
import json, os, sys
import xml.etree.ElementTree as ET
import numpy as np

URDF = "/root/reference/tests/robots/coman_floating_base/coman_floating_base.urdf"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "coman_tree.json")


def rpy(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def vec(s, d=(0.0, 0.0, 0.0)):
    return np.array([float(v) for v in s.split()]) if s else np.array(d, dtype=float)


root = ET.parse(URDF).getroot()
links = {}
for l in root.findall("link"):
    m, c = 0.0, np.zeros(3)
    ine = l.find("inertial")
    if ine is not None:
        m = float(ine.find("mass").get("value"))
        o = ine.find("origin")
        c = vec(o.get("xyz")) if o is not None else np.zeros(3)
    links[l.get("name")] = (m, c)
children = {}
for j in root.findall("joint"):
    o = j.find("origin")
    xyz = vec(o.get("xyz")) if o is not None and o.get("xyz") else np.zeros(3)
    R = rpy(*vec(o.get("rpy"))) if o is not None and o.get("rpy") else np.eye(3)
    ax = j.find("axis")
    lim = j.find("limit")
    children.setdefault(j.find("parent").get("link"), []).append(
        dict(name=j.get("name"), type=j.get("type"), child=j.find("child").get("link"), R=R, p=xyz,
             axis=vec(ax.get("xyz"), (1, 0, 0)) if ax is not None else np.array([1.0, 0, 0]),
             lower=float(lim.get("lower")) if lim is not None and lim.get("lower") else None,
             upper=float(lim.get("upper")) if lim is not None and lim.get("upper") else None))

joints, frames = [], {}
WANT = {"l_wrist", "r_wrist", "l_sole", "r_sole"}


def add_joint(name, parent, jtype, axis, R, p, lower=None, upper=None):
    joints.append(dict(name=name, parent=parent, type=jtype, axis=[float(v) for v in axis], R0=[float(v) for v in R.reshape(9)],
                       p0=[float(v) for v in p], mass=0.0, mcom=[0.0, 0.0, 0.0], lower=lower, upper=upper))
    return len(joints) - 1


def attach(link, jidx, R, p):
    """link is rigidly attached to joint jidx; (R, p) = its frame in the joint frame.  Merge its inertia, note frames,
    recurse into its child joints."""
    m, c = links[link]
    J = joints[jidx]
    J["mass"] += m
    for i in range(3):
        J["mcom"][i] += m * float((R @ c + p)[i])
    if link in WANT:
        frames[link] = dict(joint=jidx, R=[float(v) for v in R.reshape(9)], p=[float(v) for v in p])
    for ch in children.get(link, []):
        Rj, pj = R @ ch["R"], R @ ch["p"] + p          # child joint frame in jidx's frame
        if ch["type"] == "fixed":
            attach(ch["child"], jidx, Rj, pj)
        elif ch["type"] in ("revolute", "continuous", "prismatic"):
            k = add_joint(ch["name"], jidx, 1 if ch["type"] == "prismatic" else 0, ch["axis"] / np.linalg.norm(ch["axis"]), Rj, pj,
                          ch["lower"], ch["upper"])
            attach(ch["child"], k, np.eye(3), np.zeros(3))
        else:
            raise SystemExit("unsupported joint type " + ch["type"])


fl = [c for c in children["world"] if c["type"] == "floating"]
assert len(fl) == 1
I3, Z3 = np.eye(3), np.zeros(3)
k = add_joint("VIRTUALJOINT_1", -1, 1, [1, 0, 0], fl[0]["R"], fl[0]["p"])
k = add_joint("VIRTUALJOINT_2", k, 1, [0, 1, 0], I3, Z3)
k = add_joint("VIRTUALJOINT_3", k, 1, [0, 0, 1], I3, Z3)
k = add_joint("VIRTUALJOINT_4", k, 0, [1, 0, 0], I3, Z3)
k = add_joint("VIRTUALJOINT_5", k, 0, [0, 1, 0], I3, Z3)
k = add_joint("VIRTUALJOINT_6", k, 0, [0, 0, 1], I3, Z3)
attach(fl[0]["child"], k, I3, Z3)
for J in joints:
    m = J["mass"]
    J["com"] = [v / m if m > 0 else 0.0 for v in J.pop("mcom")]
assert set(frames) == WANT, frames.keys()
doc = {"source": "tests/robots/coman_floating_base/coman_floating_base.urdf (ADVRHumanoids/OpenSoT @2024-10-24), reduced by tests/golden/make_coman_tree.py",
       "n": len(joints), "total_mass": sum(J["mass"] for J in joints), "joints": joints,
       "frames": [dict(name=n, **frames[n]) for n in ("l_wrist", "r_wrist", "l_sole", "r_sole")]}
json.dump(doc, open(OUT, "w"), indent=0)
print(len(joints), "joints, mass %.4f kg ->" % doc["total_mass"], OUT, os.path.getsize(OUT), "bytes")
