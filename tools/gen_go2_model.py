#!/usr/bin/env python3
"""Derive the collapsed Go2 rigid-body model from the robot description and emit C headers.

Runs only in the build container (it reads the URDF under /root/reference); the emitted
headers are committed, so nothing on the GPU box reads the reference tree.

What the reference does with this file: Isaac Gym loads it with collapse_fixed_joints=True
(bbc/legged_gym/envs/base/legged_robot_config.py:85) so massless/fixed children fold into
their parents; links flagged dont_collapse (heads, feet) stay separate *bodies* but are
rigidly attached, which is dynamically the same as folding them.  We therefore produce
13 moving bodies (base + 4 x {hip, thigh, calf}) with composite inertias, plus the list of
collision primitives reduced to spheres/points (capsule end-spheres, box corners).

Emits the same header to the product tree and to oracle/ (data, not algorithm).
"""
import math
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np

URDF = "/root/reference/bbc/resources/robots/go2/urdf/go2.urdf"
OUTS = [
    os.path.join(os.path.dirname(__file__), "..", "quadrupedal_agility_amd", "csrc", "qa_go2_model.h"),
    os.path.join(os.path.dirname(__file__), "..", "oracle", "qa_go2_model.h"),
]

LEGS = ["FL", "FR", "RL", "RR"]
# body index convention of this build (the reference gets its order from Isaac Gym at run time)
BODY_NAMES = ["base", "Head_upper", "Head_lower"] + [f"{l}_{p}" for l in LEGS for p in ("hip", "thigh", "calf", "foot")]


def rpy_to_R(rpy):
    r, p, y = rpy
    Rx = np.array([[1, 0, 0], [0, math.cos(r), -math.sin(r)], [0, math.sin(r), math.cos(r)]])
    Ry = np.array([[math.cos(p), 0, math.sin(p)], [0, 1, 0], [-math.sin(p), 0, math.cos(p)]])
    Rz = np.array([[math.cos(y), -math.sin(y), 0], [math.sin(y), math.cos(y), 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def vec(s):
    return np.array([float(x) for x in s.split()])


def parse():
    root = ET.parse(URDF).getroot()
    links, joints = {}, {}
    for l in root.findall("link"):
        d = {"mass": 0.0, "com": np.zeros(3), "I": np.zeros((3, 3)), "cols": []}
        i = l.find("inertial")
        if i is not None:
            o = i.find("origin")
            d["mass"] = float(i.find("mass").get("value"))
            d["com"] = vec(o.get("xyz")) if o is not None else np.zeros(3)
            assert o is None or np.allclose(vec(o.get("rpy", "0 0 0")), 0)
            a = i.find("inertia").attrib
            ixx, iyy, izz = float(a["ixx"]), float(a["iyy"]), float(a["izz"])
            ixy, ixz, iyz = float(a["ixy"]), float(a["ixz"]), float(a["iyz"])
            d["I"] = np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]])
        for c in l.findall("collision"):
            o = c.find("origin")
            xyz = vec(o.get("xyz", "0 0 0")) if o is not None else np.zeros(3)
            rpy = vec(o.get("rpy", "0 0 0")) if o is not None else np.zeros(3)
            g = list(c.find("geometry"))[0]
            d["cols"].append((g.tag, dict(g.attrib), xyz, rpy_to_R(rpy)))
        links[l.get("name")] = d
    for j in root.findall("joint"):
        o = j.find("origin")
        lim = j.find("limit")
        joints[j.get("name")] = {
            "type": j.get("type"), "parent": j.find("parent").get("link"), "child": j.find("child").get("link"),
            "xyz": vec(o.get("xyz", "0 0 0")), "R": rpy_to_R(vec(o.get("rpy", "0 0 0"))),
            "axis": vec(j.find("axis").get("xyz")) if j.find("axis") is not None else np.zeros(3),
            "limit": {k: float(v) for k, v in lim.attrib.items()} if lim is not None else None,
        }
    return links, joints


def fold(links, joints, name, T_p, T_R, out_mass, out_cols, body_of):
    """Accumulate (mass, com, I about parent-frame origin) of `name` and all fixed descendants,
    expressed in the frame of the moving ancestor; collect collision prims tagged by body."""
    l = links[name]
    if l["mass"] > 0:
        c = T_p + T_R @ l["com"]
        I = T_R @ l["I"] @ T_R.T
        out_mass.append((l["mass"], c, I))
    for (tag, attr, xyz, R) in l["cols"]:
        out_cols.append((body_of(name), tag, attr, T_p + T_R @ xyz, T_R @ R))
    for jn, j in joints.items():
        if j["parent"] == name and j["type"] == "fixed":
            fold(links, joints, j["child"], T_p + T_R @ j["xyz"], T_R @ j["R"], out_mass, out_cols, body_of)


def composite(parts):
    m = sum(p[0] for p in parts)
    c = sum(p[0] * p[1] for p in parts) / m
    I = np.zeros((3, 3))
    for (mi, ci, Ii) in parts:
        d = ci - c
        I += Ii + mi * (d @ d * np.eye(3) - np.outer(d, d))
    return m, c, I


def prims_to_points(cols):
    """capsule (cylinder, replace_cylinder_with_capsule=True: legged_robot_config.py:89) -> its two
    end spheres; sphere -> itself; box -> 8 corner points of radius 0."""
    pts = []
    for (body, tag, attr, p, R) in cols:
        if tag == "sphere":
            pts.append((body, p, float(attr["radius"])))
        elif tag == "cylinder":
            hl = 0.5 * float(attr["length"])
            ax = R @ np.array([0, 0, 1.0])
            pts.append((body, p + hl * ax, float(attr["radius"])))
            pts.append((body, p - hl * ax, float(attr["radius"])))
        elif tag == "box":
            hx, hy, hz = 0.5 * vec(attr["size"])
            for sx in (-1, 1):
                for sy in (-1, 1):
                    for sz in (-1, 1):
                        pts.append((body, p + R @ np.array([sx * hx, sy * hy, sz * hz]), 0.0))
        else:
            raise ValueError(tag)
    return pts


def fmt(x):
    s = f"{float(x):.9g}"
    if "." not in s and "e" not in s and "n" not in s:
        s += ".0"
    return s + "f"


def arr(a):
    return "{" + ", ".join(fmt(x) for x in np.asarray(a).ravel()) + "}"


def sym6(I):
    return [I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]]


def main():
    links, joints = parse()

    def body_of(name):
        if name in BODY_NAMES:
            return BODY_NAMES.index(name)
        for l in LEGS:  # calflower* fold into the calf body
            if name.startswith(l + "_calflower"):
                return BODY_NAMES.index(l + "_calf")
        return 0  # imu, radar -> base

    base_parts, base_cols = [], []
    fold(links, joints, "base", np.zeros(3), np.eye(3), base_parts, base_cols, body_of)
    bm, bc, bI = composite(base_parts)
    base_pts = prims_to_points(base_cols)

    leg = {k: [] for k in ("hip_org", "thigh_org", "calf_org", "foot_org", "mass", "com", "I", "lim_lo", "lim_hi", "effort", "vel")}
    leg_pts = []
    total = bm
    for l in LEGS:
        jh, jt, jc = joints[f"{l}_hip_joint"], joints[f"{l}_thigh_joint"], joints[f"{l}_calf_joint"]
        assert np.allclose(jh["axis"], [1, 0, 0]) and np.allclose(jt["axis"], [0, 1, 0]) and np.allclose(jc["axis"], [0, 1, 0])
        leg["hip_org"].append(jh["xyz"]); leg["thigh_org"].append(jt["xyz"]); leg["calf_org"].append(jc["xyz"])
        leg["foot_org"].append(joints[f"{l}_foot_joint"]["xyz"])
        ms, cs, Is, pts_l = [], [], [], []
        for k, part in enumerate(("hip", "thigh", "calf")):
            parts, cols = [], []
            fold(links, joints, f"{l}_{part}", np.zeros(3), np.eye(3), parts, cols, body_of)
            m, c, I = composite(parts)
            total += m
            ms.append(m); cs.append(c); Is.append(sym6(I))
            for (body, p, r) in prims_to_points(cols):
                pts_l.append((k, body, p, r))
        leg["mass"].append(ms); leg["com"].append(cs); leg["I"].append(Is)
        leg["lim_lo"].append([j["limit"]["lower"] for j in (jh, jt, jc)])
        leg["lim_hi"].append([j["limit"]["upper"] for j in (jh, jt, jc)])
        leg["effort"].append([j["limit"]["effort"] for j in (jh, jt, jc)])
        leg["vel"].append([j["limit"]["velocity"] for j in (jh, jt, jc)])
        leg_pts.append(pts_l)
    npl = len(leg_pts[0])
    assert all(len(p) == npl for p in leg_pts)
    # foot sphere must be the last calf prim listed first for convenience: reorder so index 0 is the foot
    for pts_l in leg_pts:
        fi = [i for i, (k, body, p, r) in enumerate(pts_l) if BODY_NAMES[body].endswith("_foot")]
        assert len(fi) == 1
        pts_l.insert(0, pts_l.pop(fi[0]))

    L = []
    L.append("/* GENERATED by tools/gen_go2_model.py from bbc/resources/robots/go2/urdf/go2.urdf -- do not edit.")
    L.append(" * Collapsed Unitree Go2 model: 13 moving bodies, 12 revolute DoF (order FL,FR,RL,RR x hip,thigh,calf),")
    L.append(" * collision primitives reduced to points/spheres.  Units SI.  Inertia order xx,yy,zz,xy,xz,yz about the CoM,")
    L.append(" * in the link frame. */")
    L.append("#ifndef QA_GO2_MODEL_H")
    L.append("#define QA_GO2_MODEL_H")
    L.append(f"#define QA_NUM_BODIES {len(BODY_NAMES)}")
    L.append(f"#define QA_NUM_LEG_PTS {npl}   /* per leg; index 0 is the foot sphere */")
    L.append(f"#define QA_NUM_BASE_PTS {len(base_pts)}")
    L.append(f"#define QA_TOTAL_MASS {fmt(total)}")
    L.append(f"static const float QA_BASE_MASS = {fmt(bm)};")
    L.append(f"static const float QA_BASE_COM[3] = {arr(bc)};")
    L.append(f"static const float QA_BASE_I[6] = {arr(sym6(bI))};")
    L.append(f"static const float QA_HIP_ORG[4][3] = {arr_2d(leg['hip_org'])};   /* base -> hip joint */")
    L.append(f"static const float QA_THIGH_ORG[4][3] = {arr_2d(leg['thigh_org'])}; /* hip -> thigh joint */")
    L.append(f"static const float QA_CALF_ORG[4][3] = {arr_2d(leg['calf_org'])};  /* thigh -> calf joint */")
    L.append(f"static const float QA_FOOT_ORG[4][3] = {arr_2d(leg['foot_org'])};  /* calf -> foot body origin */")
    L.append(f"static const float QA_LINK_MASS[4][3] = {arr_2d(leg['mass'])};")
    L.append(f"static const float QA_LINK_COM[4][3][3] = {arr_3d(leg['com'])};")
    L.append(f"static const float QA_LINK_I[4][3][6] = {arr_3d(leg['I'])};")
    L.append(f"static const float QA_DOF_LOWER[4][3] = {arr_2d(leg['lim_lo'])};")
    L.append(f"static const float QA_DOF_UPPER[4][3] = {arr_2d(leg['lim_hi'])};")
    L.append(f"static const float QA_DOF_EFFORT[4][3] = {arr_2d(leg['effort'])};")
    L.append(f"static const float QA_DOF_VELLIM[4][3] = {arr_2d(leg['vel'])};")
    L.append("/* leg collision points: link (0 hip,1 thigh,2 calf), body id, local position, radius */")
    L.append(f"static const int QA_LEG_PT_LINK[4][QA_NUM_LEG_PTS] = {{" + ", ".join("{" + ", ".join(str(k) for (k, b, p, r) in pl) + "}" for pl in leg_pts) + "};")
    L.append(f"static const int QA_LEG_PT_BODY[4][QA_NUM_LEG_PTS] = {{" + ", ".join("{" + ", ".join(str(b) for (k, b, p, r) in pl) + "}" for pl in leg_pts) + "};")
    L.append(f"static const float QA_LEG_PT_POS[4][QA_NUM_LEG_PTS][3] = {{" + ", ".join("{" + ", ".join(arr(p) for (k, b, p, r) in pl) + "}" for pl in leg_pts) + "};")
    L.append(f"static const float QA_LEG_PT_RAD[4][QA_NUM_LEG_PTS] = {{" + ", ".join("{" + ", ".join(fmt(r) for (k, b, p, r) in pl) + "}" for pl in leg_pts) + "};")
    L.append(f"static const int QA_BASE_PT_BODY[QA_NUM_BASE_PTS] = {{" + ", ".join(str(b) for (b, p, r) in base_pts) + "};")
    L.append(f"static const float QA_BASE_PT_POS[QA_NUM_BASE_PTS][3] = {{" + ", ".join(arr(p) for (b, p, r) in base_pts) + "};")
    L.append(f"static const float QA_BASE_PT_RAD[QA_NUM_BASE_PTS] = {{" + ", ".join(fmt(r) for (b, p, r) in base_pts) + "};")
    L.append("#endif")
    text = "\n".join(L) + "\n"
    for o in OUTS:
        with open(o, "w") as f:
            f.write(text)
    print(f"total mass {total:.4f} kg; base composite m={bm:.4f} com={bc}; {npl} pts/leg, {len(base_pts)} base pts")
    print("body names:", BODY_NAMES)


def arr_2d(a):
    return "{" + ", ".join(arr(x) for x in a) + "}"


def arr_3d(a):
    return "{" + ", ".join(arr_2d(x) for x in a) + "}"


if __name__ == "__main__":
    main()
