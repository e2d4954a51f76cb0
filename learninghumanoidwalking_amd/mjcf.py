"""MJCF-subset compiler: XML -> ``model.Model`` (host-side, numpy).

Replaces, for the subset of MJCF the reference's robots use, what the reference
delegates to ``mujoco.MjSpec.from_file(path).compile()``
(reference envs/common/mujoco_env.py:24-25): element defaults, body tree,
free/slide/hinge joints, ``inertiafromgeom``, plane/sphere/capsule/box geoms,
``<exclude>``, joint motors, sites, and the compile-time constants the constraint
solver needs (``dof_invweight0``, ``body_invweight0``, ``stat.meaninertia`` --
MuJoCo's ``mj_setConst``).

Only what the hot path needs is implemented; anything else in the file raises
``MjcfError`` unless it is known to be inert for dynamics (cameras, lights,
sensors, keyframes, custom, visual, asset, size, statistic).
"""

from __future__ import annotations

import math
import xml.etree.ElementTree as ET

import numpy as np

from .model import (GEOM_BOX, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_ELLIPSOID, GEOM_PLANE, GEOM_SPHERE, JNT_FREE, JNT_HINGE, JNT_SLIDE, Model)


class MjcfError(ValueError):
    pass


_GEOM_TYPES = {"plane": GEOM_PLANE, "sphere": GEOM_SPHERE, "capsule": GEOM_CAPSULE, "ellipsoid": GEOM_ELLIPSOID, "cylinder": GEOM_CYLINDER,
               "box": GEOM_BOX}
_JNT_TYPES = {"free": JNT_FREE, "slide": JNT_SLIDE, "hinge": JNT_HINGE}
_INERT_TAGS = {"camera", "light", "sensor", "keyframe", "custom", "visual", "asset", "size", "statistic"}

# narrow-phase functions that exist in the stepper (geom types ordered type1 <= type2)
SUPPORTED_PAIRS = {
    (GEOM_PLANE, GEOM_SPHERE), (GEOM_PLANE, GEOM_CAPSULE), (GEOM_PLANE, GEOM_BOX),
    (GEOM_SPHERE, GEOM_SPHERE), (GEOM_SPHERE, GEOM_CAPSULE), (GEOM_CAPSULE, GEOM_CAPSULE), (GEOM_BOX, GEOM_BOX),
    (GEOM_SPHERE, GEOM_BOX), (GEOM_CAPSULE, GEOM_BOX),
    # round 6: the two cylinder pairs MuJoCo resolves analytically (mjc_PlaneCylinder, mjc_SphereCylinder); capsule- / cylinder- / box-cylinder go
    # through its general convex collider there and are refused here
    (GEOM_PLANE, GEOM_CYLINDER), (GEOM_SPHERE, GEOM_CYLINDER),
    (GEOM_PLANE, GEOM_ELLIPSOID),     # mjc_PlaneConvex on a smooth geom: one contact at the support point
}
_TYPE_NAMES = {GEOM_PLANE: "plane", 1: "hfield", GEOM_SPHERE: "sphere", GEOM_CAPSULE: "capsule", 4: "ellipsoid", GEOM_CYLINDER: "cylinder", GEOM_BOX: "box", 7: "mesh"}

_DEF_SOLREF = (0.02, 1.0)
_DEF_SOLIMP = (0.9, 0.95, 0.001, 0.5, 2.0)


# ----------------------------------------------------------------------------- small math
def _floats(s, n=None):
    v = [float(x) for x in s.replace(",", " ").split()]
    if n is not None and len(v) != n:
        raise MjcfError(f"expected {n} numbers, got {s!r}")
    return v


def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([
        aw * bw - ax * bx - ay * by - az * bz,
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
    ])


def quat2mat(q):
    w, x, y, z = q
    return np.array([
        [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z],
    ])


def mat2quat(R):
    """Rotation matrix -> unit quaternion (w,x,y,z), w >= 0."""
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = math.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    if q[0] < 0:
        q = -q
    return q / np.linalg.norm(q)


def axisangle2quat(axis, angle):
    s = math.sin(angle * 0.5)
    return np.array([math.cos(angle * 0.5), axis[0] * s, axis[1] * s, axis[2] * s])


def z2quat(vec):
    """Quaternion rotating (0,0,1) onto vec (MuJoCo ``mjuu_z2quat``)."""
    v = np.asarray(vec, dtype=float)
    v = v / np.linalg.norm(v)
    axis = np.cross([0.0, 0.0, 1.0], v)
    s = np.linalg.norm(axis)
    if s < 1e-10:
        return np.array([1.0, 0.0, 0.0, 0.0])
    axis /= s
    ang = math.atan2(s, v[2])
    return axisangle2quat(axis, ang)


# ----------------------------------------------------------------------------- defaults
class _Defaults:
    """<default> class tree; attribute lookup falls back to parent classes."""

    def __init__(self):
        self.classes = {"main": {}}
        self.parent = {"main": None}

    def load(self, elem, cls="main"):
        for child in elem:
            if child.tag == "default":
                name = child.get("class")
                if name is None:  # nested unnamed default at top level == main
                    self.load(child, cls)
                    continue
                self.classes.setdefault(name, {})
                self.parent[name] = cls
                self.load(child, name)
            else:
                self.classes[cls].setdefault(child.tag, {}).update(child.attrib)

    def get(self, cls, tag):
        chain = []
        c = cls or "main"
        if c not in self.classes:
            raise MjcfError(f"unknown default class {c!r}")
        while c is not None:
            chain.append(self.classes[c].get(tag, {}))
            c = self.parent[c]
        out = {}
        for d in reversed(chain):
            out.update(d)
        return out


# ----------------------------------------------------------------------------- compiler
class _Compiler:
    def __init__(self, root):
        if root.tag != "mujoco":
            raise MjcfError("root element must be <mujoco>")
        self.root = root
        comp = root.find("compiler")
        comp = comp.attrib if comp is not None else {}
        if comp.get("coordinate", "local") != "local":
            raise MjcfError("only coordinate='local' is supported")
        self.degrees = comp.get("angle", "degree") == "degree"
        self.eulerseq = comp.get("eulerseq", "xyz")
        self.inertiafromgeom = comp.get("inertiafromgeom", "auto")
        self.autolimits = comp.get("autolimits", "true") == "true"
        self.defaults = _Defaults()
        for d in root.findall("default"):
            self.defaults.load(d)
        self.bodies = []  # dicts
        self.joints = []
        self.geoms = []
        self.sites = []

    # -- attribute helpers
    def _angle(self, a):
        return math.radians(a) if self.degrees else a

    def _orientation(self, attrib):
        if "quat" in attrib:
            q = np.array(_floats(attrib["quat"], 4))
            return q / np.linalg.norm(q)
        if "euler" in attrib:
            e = [self._angle(x) for x in _floats(attrib["euler"], 3)]
            q = np.array([1.0, 0, 0, 0])
            for ch, ang in zip(self.eulerseq, e):
                ax = np.zeros(3)
                ax["xyz".index(ch.lower())] = 1.0
                r = axisangle2quat(ax, ang)
                q = quat_mul(q, r) if ch.islower() else quat_mul(r, q)
            return q / np.linalg.norm(q)
        if "axisangle" in attrib:
            v = _floats(attrib["axisangle"], 4)
            ax = np.array(v[:3]) / np.linalg.norm(v[:3])
            return axisangle2quat(ax, self._angle(v[3]))
        if "zaxis" in attrib:
            return z2quat(_floats(attrib["zaxis"], 3))
        if "xyaxes" in attrib:
            v = _floats(attrib["xyaxes"], 6)
            x = np.array(v[:3]) / np.linalg.norm(v[:3])
            y = np.array(v[3:]) - np.dot(x, v[3:]) * x
            y /= np.linalg.norm(y)
            return mat2quat(np.stack([x, y, np.cross(x, y)], axis=1))
        return np.array([1.0, 0, 0, 0])

    def _merged(self, elem, tag, childclass):
        cls = elem.get("class", childclass)
        out = dict(self.defaults.get(cls, tag))
        out.update(elem.attrib)
        return out

    # -- tree walk
    def walk(self):
        wb = self.root.find("worldbody")
        if wb is None:
            raise MjcfError("missing <worldbody>")
        self.bodies.append(dict(name="world", parent=0, pos=np.zeros(3), quat=np.array([1.0, 0, 0, 0]),
                                inertial=None, jnts=[], geoms=[]))
        self._body_children(wb, 0, None)

    def _body_children(self, elem, bid, childclass):
        for ch in elem:
            if ch.tag == "body":
                cc = ch.get("childclass", childclass)
                b = dict(name=ch.get("name", f"body{len(self.bodies)}"), parent=bid,
                         pos=np.array(_floats(ch.get("pos", "0 0 0"), 3)), quat=self._orientation(ch.attrib),
                         inertial=None, jnts=[], geoms=[])
                self.bodies.append(b)
                nid = len(self.bodies) - 1
                self._body_children(ch, nid, cc)
            elif ch.tag == "inertial":
                a = ch.attrib
                inert = dict(pos=np.array(_floats(a.get("pos", "0 0 0"), 3)), quat=self._orientation(a),
                             mass=float(a["mass"]))
                if "diaginertia" in a:
                    inert["diag"] = np.array(_floats(a["diaginertia"], 3))
                elif "fullinertia" in a:
                    f = _floats(a["fullinertia"], 6)
                    inert["full"] = np.array([[f[0], f[3], f[4]], [f[3], f[1], f[5]], [f[4], f[5], f[2]]])
                else:
                    raise MjcfError("<inertial> needs diaginertia or fullinertia")
                self.bodies[bid]["inertial"] = inert
            elif ch.tag in ("joint", "freejoint"):
                if bid == 0:
                    raise MjcfError("joint in worldbody")
                a = self._merged(ch, "joint", childclass) if ch.tag == "joint" else dict(ch.attrib, type="free")
                jt = _JNT_TYPES.get(a.get("type", "hinge"))
                if jt is None:
                    raise MjcfError(f"unsupported joint type {a.get('type')!r}")
                rng = _floats(a["range"], 2) if "range" in a else [0.0, 0.0]
                lim = a.get("limited", "auto")
                limited = (lim == "true") or (lim == "auto" and self.autolimits and "range" in a)
                if jt == JNT_HINGE:
                    rng = [self._angle(r) for r in rng]
                axis = np.array(_floats(a.get("axis", "0 0 1"), 3))
                axis = axis / np.linalg.norm(axis)
                ref = float(a.get("ref", 0.0))
                if jt == JNT_HINGE:
                    ref = self._angle(ref)
                j = dict(name=a.get("name", f"joint{len(self.joints)}"), type=jt, body=bid,
                         pos=np.array(_floats(a.get("pos", "0 0 0"), 3)), axis=axis, range=rng,
                         limited=int(limited and jt != JNT_FREE), damping=float(a.get("damping", 0.0)),
                         armature=float(a.get("armature", 0.0)), frictionloss=float(a.get("frictionloss", 0.0)),
                         stiffness=float(a.get("stiffness", 0.0)), ref=ref,
                         solreflimit=_floats(a.get("solreflimit", "0.02 1"), 2),
                         solimplimit=self._solimp(a.get("solimplimit")),
                         solreffriction=_floats(a.get("solreffriction", "0.02 1"), 2),
                         solimpfriction=self._solimp(a.get("solimpfriction")),
                         margin=float(a.get("margin", 0.0)))
                if j["stiffness"] != 0.0:
                    raise MjcfError("joint stiffness is not supported")
                self.joints.append(j)
                self.bodies[bid]["jnts"].append(len(self.joints) - 1)
            elif ch.tag == "geom":
                a = self._merged(ch, "geom", childclass)
                self._add_geom(a, bid)
            elif ch.tag == "site":
                a = self._merged(ch, "site", childclass)
                self.sites.append(dict(name=a.get("name", f"site{len(self.sites)}"), body=bid,
                                       pos=np.array(_floats(a.get("pos", "0 0 0"), 3)),
                                       quat=self._orientation(a)))
            elif ch.tag in _INERT_TAGS:
                continue
            else:
                raise MjcfError(f"unsupported element <{ch.tag}> in body")

    @staticmethod
    def _solimp(s):
        v = list(_DEF_SOLIMP)
        if s is not None:
            u = _floats(s)
            v[: len(u)] = u
        return v

    def _add_geom(self, a, bid):
        tname = a.get("type", "sphere")
        if tname not in _GEOM_TYPES:
            name = a.get("name", f"geom{len(self.geoms)}")
            if tname == "mesh":
                raise MjcfError(f"geom {name!r}: mesh geoms are not supported (replace it by the primitive that encloses it: box, capsule, sphere or cylinder)")
            raise MjcfError(f"geom {name!r}: unsupported geom type {tname!r}")
        gt = _GEOM_TYPES[tname]
        size = _floats(a.get("size", "0 0 0"))
        size = (size + [0.0, 0.0, 0.0])[:3]
        pos = np.array(_floats(a.get("pos", "0 0 0"), 3))
        quat = self._orientation(a)
        if "fromto" in a:
            if gt not in (GEOM_CAPSULE, GEOM_CYLINDER):
                raise MjcfError("fromto only supported for capsules and cylinders")
            ft = np.array(_floats(a["fromto"], 6))
            p0, p1 = ft[:3], ft[3:]
            pos = 0.5 * (p0 + p1)
            quat = z2quat(p1 - p0)
            size = [size[0], 0.5 * float(np.linalg.norm(p1 - p0)), 0.0]
        fr = _floats(a.get("friction", "1 0.005 0.0001"))
        fr = (fr + [0.005, 0.0001])[:3] if len(fr) < 3 else fr[:3]
        g = dict(name=a.get("name", f"geom{len(self.geoms)}"), type=gt, body=bid, pos=pos, quat=quat,
                 size=np.array(size, dtype=float), contype=int(a.get("contype", 1)),
                 conaffinity=int(a.get("conaffinity", 1)), condim=int(a.get("condim", 3)),
                 priority=int(a.get("priority", 0)), friction=np.array(fr, dtype=float),
                 solmix=float(a.get("solmix", 1.0)), solref=_floats(a.get("solref", "0.02 1"), 2),
                 solimp=self._solimp(a.get("solimp")), margin=float(a.get("margin", 0.0)),
                 gap=float(a.get("gap", 0.0)), density=float(a.get("density", 1000.0)),
                 mass=float(a["mass"]) if "mass" in a else None, group=int(a.get("group", 0)))
        if g["condim"] not in (1, 3):
            raise MjcfError("only condim 1 and 3 are supported")
        self.geoms.append(g)
        self.bodies[bid]["geoms"].append(len(self.geoms) - 1)

    # -- geom inertia (MuJoCo user_objects.cc GetVolume/SetInertia; formulas SURVEY.md A.2)
    @staticmethod
    def _geom_inertia(g):
        t, s = g["type"], g["size"]
        if t == GEOM_SPHERE:
            vol = 4.0 / 3.0 * math.pi * s[0] ** 3
            m = g["mass"] if g["mass"] is not None else vol * g["density"]
            i = 0.4 * m * s[0] ** 2
            return m, np.array([i, i, i])
        if t == GEOM_BOX:
            vol = 8 * s[0] * s[1] * s[2]
            m = g["mass"] if g["mass"] is not None else vol * g["density"]
            return m, m / 3.0 * np.array([s[1] ** 2 + s[2] ** 2, s[0] ** 2 + s[2] ** 2, s[0] ** 2 + s[1] ** 2])
        if t == GEOM_CAPSULE:
            r, l = s[0], s[1]
            vc = math.pi * r * r * 2 * l
            vs = 4.0 / 3.0 * math.pi * r ** 3
            vol = vc + vs
            m = g["mass"] if g["mass"] is not None else vol * g["density"]
            mc, ms = m * vc / vol, m * vs / vol
            izz = mc * r * r / 2 + ms * 0.4 * r * r
            ixx = mc * (r * r / 4 + l * l / 3) + ms * (0.4 * r * r + l * l + 0.75 * r * l)
            return m, np.array([ixx, ixx, izz])
        if t == GEOM_ELLIPSOID:     # semi-axes s
            vol = 4.0 / 3.0 * math.pi * s[0] * s[1] * s[2]
            m = g["mass"] if g["mass"] is not None else vol * g["density"]
            return m, m / 5.0 * np.array([s[1] ** 2 + s[2] ** 2, s[0] ** 2 + s[2] ** 2, s[0] ** 2 + s[1] ** 2])
        if t == GEOM_CYLINDER:      # radius s[0], half length s[1] along z
            r, l = s[0], s[1]
            vol = math.pi * r * r * 2 * l
            m = g["mass"] if g["mass"] is not None else vol * g["density"]
            ixx = m * (3 * r * r + (2 * l) ** 2) / 12.0
            return m, np.array([ixx, ixx, m * r * r / 2])
        return 0.0, np.zeros(3)

    def _body_inertial(self, b):
        use_geoms = self.inertiafromgeom == "true" or (self.inertiafromgeom == "auto" and b["inertial"] is None)
        if not use_geoms:
            it = b["inertial"]
            if it is None:
                return 0.0, np.zeros(3), np.array([1.0, 0, 0, 0]), np.zeros(3)
            if "diag" in it:
                return it["mass"], it["pos"], it["quat"], it["diag"]
            full = quat2mat(it["quat"]) @ it["full"] @ quat2mat(it["quat"]).T
            return (it["mass"], it["pos"]) + self._principal(full)
        ms, cs, Is = [], [], []
        for gi in b["geoms"]:
            g = self.geoms[gi]
            m, diag = self._geom_inertia(g)
            if m <= 0:
                continue
            R = quat2mat(g["quat"])
            ms.append(m)
            cs.append(g["pos"])
            Is.append(R @ np.diag(diag) @ R.T)
        if not ms:
            return 0.0, np.zeros(3), np.array([1.0, 0, 0, 0]), np.zeros(3)
        M = float(sum(ms))
        com = sum(m * c for m, c in zip(ms, cs)) / M
        full = np.zeros((3, 3))
        for m, c, I in zip(ms, cs, Is):
            d = c - com
            full += I + m * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
        return (M, com) + self._principal(full)

    @staticmethod
    def _principal(full):
        if np.allclose(full, np.diag(np.diag(full)), atol=1e-14 * max(1.0, float(np.abs(full).max()))):
            return np.array([1.0, 0, 0, 0]), np.diag(full).copy()
        w, V = np.linalg.eigh(full)
        if np.linalg.det(V) < 0:
            V[:, 2] = -V[:, 2]
        return mat2quat(V), w

    # -- assemble Model
    def build(self) -> Model:
        self.walk()
        opt = self.root.find("option")
        opt = opt.attrib if opt is not None else {}
        flag = self.root.find("option/flag")
        m = Model()
        m.timestep = float(opt.get("timestep", 0.002))
        m.gravity = tuple(_floats(opt.get("gravity", "0 0 -9.81"), 3))
        m.tolerance = float(opt.get("tolerance", 1e-8))
        m.ls_tolerance = float(opt.get("ls_tolerance", 0.01))
        m.iterations = int(opt.get("iterations", 100))
        m.ls_iterations = int(opt.get("ls_iterations", 50))
        m.impratio = float(opt.get("impratio", 1.0))
        m.o_margin = float(opt.get("o_margin", 0.0))
        if opt.get("cone", "pyramidal") != "pyramidal":
            raise MjcfError("only the pyramidal friction cone is supported")
        if opt.get("integrator", "Euler") != "Euler":
            raise MjcfError("only the Euler integrator is supported")
        if opt.get("solver", "Newton") != "Newton":
            raise MjcfError("only the Newton solver is supported")
        if flag is not None:
            from .model import DSBL_EULERDAMP, DSBL_REFSAFE, DSBL_WARMSTART
            for key, bit in (("eulerdamp", DSBL_EULERDAMP), ("refsafe", DSBL_REFSAFE), ("warmstart", DSBL_WARMSTART)):
                if flag.get(key, "enable") == "disable":
                    m.disableflags |= bit

        nb = len(self.bodies)
        a = m.arrays
        m.nbody, m.njnt, m.ngeom, m.nsite = nb, len(self.joints), len(self.geoms), len(self.sites)
        m.body_names = [b["name"] for b in self.bodies]
        m.jnt_names = [j["name"] for j in self.joints]
        m.geom_names = [g["name"] for g in self.geoms]
        m.site_names = [s["name"] for s in self.sites]

        a["body_parentid"] = np.array([b["parent"] for b in self.bodies], dtype=np.int32)
        a["body_pos"] = np.array([b["pos"] for b in self.bodies], dtype=float).reshape(nb, 3)
        a["body_quat"] = np.array([b["quat"] for b in self.bodies], dtype=float).reshape(nb, 4)
        mass = np.zeros(nb)
        ipos = np.zeros((nb, 3))
        iquat = np.tile(np.array([1.0, 0, 0, 0]), (nb, 1))
        inertia = np.zeros((nb, 3))
        for i, b in enumerate(self.bodies):
            if i == 0:
                continue
            mass[i], ipos[i], iquat[i], inertia[i] = self._body_inertial(b)
            if b["jnts"] and mass[i] <= 0:
                raise MjcfError(f"moving body {b['name']!r} has no mass")
        a["body_mass"], a["body_ipos"], a["body_iquat"], a["body_inertia"] = mass, ipos, iquat, inertia

        # joints / dofs
        jq, jd = [], []
        nq = nv = 0
        dof_body, dof_jnt, dof_parent = [], [], []
        dof_arm, dof_damp, dof_fl, dof_solref, dof_solimp = [], [], [], [], []
        qpos0 = []
        body_jntadr = np.full(nb, -1, dtype=np.int32)
        body_jntnum = np.zeros(nb, dtype=np.int32)
        body_dofadr = np.full(nb, -1, dtype=np.int32)
        body_dofnum = np.zeros(nb, dtype=np.int32)
        last_dof_of_body = np.full(nb, -1, dtype=np.int64)
        # joints were appended in body order because bodies are walked depth-first and each
        # body's joints are contiguous; verify
        order = [ji for b in self.bodies for ji in b["jnts"]]
        if order != list(range(len(self.joints))):
            raise MjcfError("internal: joint order")
        for bi, b in enumerate(self.bodies):
            if bi == 0:
                continue
            # dof parent: last dof of the nearest ancestor that has dofs
            p = b["parent"]
            while p > 0 and last_dof_of_body[p] < 0:
                p = self.bodies[p]["parent"]
            prev = int(last_dof_of_body[p]) if p > 0 else -1
            if b["jnts"]:
                body_jntadr[bi] = b["jnts"][0]
                body_jntnum[bi] = len(b["jnts"])
                body_dofadr[bi] = nv
            for ji in b["jnts"]:
                j = self.joints[ji]
                jq.append(nq)
                jd.append(nv)
                if j["type"] == JNT_FREE:
                    if len(b["jnts"]) != 1 or b["parent"] != 0:
                        raise MjcfError("free joint must be alone on a top-level body")
                    nd, nqj = 6, 7
                    qpos0 += list(b["pos"]) + list(b["quat"])
                else:
                    nd, nqj = 1, 1
                    qpos0.append(j["ref"])
                    if j["ref"] != 0.0:
                        raise MjcfError("joint ref != 0 is not supported")
                for k in range(nd):
                    dof_body.append(bi)
                    dof_jnt.append(ji)
                    dof_parent.append(prev)
                    prev = nv + k
                    dof_arm.append(j["armature"])
                    dof_damp.append(j["damping"])
                    dof_fl.append(j["frictionloss"])
                    dof_solref.append(j["solreffriction"])
                    dof_solimp.append(j["solimpfriction"])
                nq += nqj
                nv += nd
            if b["jnts"]:
                body_dofnum[bi] = nv - body_dofadr[bi]
                last_dof_of_body[bi] = nv - 1
        m.nq, m.nv = nq, nv
        a["body_jntadr"], a["body_jntnum"] = body_jntadr, body_jntnum
        a["body_dofadr"], a["body_dofnum"] = body_dofadr, body_dofnum
        nj = m.njnt
        a["jnt_type"] = np.array([j["type"] for j in self.joints], dtype=np.int32)
        a["jnt_bodyid"] = np.array([j["body"] for j in self.joints], dtype=np.int32)
        a["jnt_qposadr"] = np.array(jq, dtype=np.int32)
        a["jnt_dofadr"] = np.array(jd, dtype=np.int32)
        a["jnt_limited"] = np.array([j["limited"] for j in self.joints], dtype=np.int32)
        a["jnt_pos"] = np.array([j["pos"] for j in self.joints], dtype=float).reshape(nj, 3)
        a["jnt_axis"] = np.array([j["axis"] for j in self.joints], dtype=float).reshape(nj, 3)
        a["jnt_range"] = np.array([j["range"] for j in self.joints], dtype=float).reshape(nj, 2)
        a["jnt_solref"] = np.array([j["solreflimit"] for j in self.joints], dtype=float).reshape(nj, 2)
        a["jnt_solimp"] = np.array([j["solimplimit"] for j in self.joints], dtype=float).reshape(nj, 5)
        a["jnt_margin"] = np.array([j["margin"] for j in self.joints], dtype=float)
        a["dof_bodyid"] = np.array(dof_body, dtype=np.int32)
        a["dof_jntid"] = np.array(dof_jnt, dtype=np.int32)
        a["dof_parentid"] = np.array(dof_parent, dtype=np.int32)
        a["dof_armature"] = np.array(dof_arm, dtype=float)
        a["dof_damping"] = np.array(dof_damp, dtype=float)
        a["dof_frictionloss"] = np.array(dof_fl, dtype=float)
        a["dof_solref"] = np.array(dof_solref, dtype=float).reshape(nv, 2)
        a["dof_solimp"] = np.array(dof_solimp, dtype=float).reshape(nv, 5)
        a["qpos0"] = np.array(qpos0, dtype=float)

        # rootid / weldid
        rootid = np.zeros(nb, dtype=np.int32)
        weldid = np.zeros(nb, dtype=np.int32)
        for i in range(1, nb):
            p = self.bodies[i]["parent"]
            rootid[i] = i if p == 0 else rootid[p]
            weldid[i] = i if self.bodies[i]["jnts"] else weldid[p]
        a["body_rootid"], a["body_weldid"] = rootid, weldid

        # geoms
        ng = m.ngeom
        G = self.geoms
        a["geom_type"] = np.array([g["type"] for g in G], dtype=np.int32)
        a["geom_bodyid"] = np.array([g["body"] for g in G], dtype=np.int32)
        a["geom_contype"] = np.array([g["contype"] for g in G], dtype=np.int32)
        a["geom_conaffinity"] = np.array([g["conaffinity"] for g in G], dtype=np.int32)
        a["geom_condim"] = np.array([g["condim"] for g in G], dtype=np.int32)
        a["geom_priority"] = np.array([g["priority"] for g in G], dtype=np.int32)
        a["geom_pos"] = np.array([g["pos"] for g in G], dtype=float).reshape(ng, 3)
        a["geom_quat"] = np.array([g["quat"] for g in G], dtype=float).reshape(ng, 4)
        a["geom_size"] = np.array([g["size"] for g in G], dtype=float).reshape(ng, 3)
        a["geom_friction"] = np.array([g["friction"] for g in G], dtype=float).reshape(ng, 3)
        a["geom_solmix"] = np.array([g["solmix"] for g in G], dtype=float)
        a["geom_solref"] = np.array([g["solref"] for g in G], dtype=float).reshape(ng, 2)
        a["geom_solimp"] = np.array([g["solimp"] for g in G], dtype=float).reshape(ng, 5)
        a["geom_margin"] = np.array([g["margin"] for g in G], dtype=float)
        a["geom_gap"] = np.array([g["gap"] for g in G], dtype=float)

        # sites
        ns = m.nsite
        a["site_bodyid"] = np.array([s["body"] for s in self.sites], dtype=np.int32)
        a["site_pos"] = np.array([s["pos"] for s in self.sites], dtype=float).reshape(ns, 3)
        a["site_quat"] = np.array([s["quat"] for s in self.sites], dtype=float).reshape(ns, 4)

        # actuators
        acts = []
        act = self.root.find("actuator")
        if act is not None:
            for e in act:
                if e.tag != "motor":
                    raise MjcfError(f"unsupported actuator <{e.tag}> (only <motor>)")
                at = dict(self.defaults.get(e.get("class"), "motor"))
                at.update(e.attrib)
                if "joint" not in at:
                    raise MjcfError("motor needs a joint transmission")
                jid = m.jnt_names.index(at["joint"])
                if self.joints[jid]["type"] == JNT_FREE:
                    raise MjcfError("motor on a free joint")
                gear = _floats(at.get("gear", "1"))[0]
                cr = _floats(at["ctrlrange"], 2) if "ctrlrange" in at else [0.0, 0.0]
                fr = _floats(at["forcerange"], 2) if "forcerange" in at else [0.0, 0.0]
                cl = at.get("ctrllimited", "auto")
                fl = at.get("forcelimited", "auto")
                acts.append(dict(name=at.get("name", f"act{len(acts)}"), jnt=jid, gear=gear, ctrlrange=cr,
                                 forcerange=fr,
                                 ctrllimited=int(cl == "true" or (cl == "auto" and self.autolimits and "ctrlrange" in at)),
                                 forcelimited=int(fl == "true" or (fl == "auto" and self.autolimits and "forcerange" in at))))
        m.nu = len(acts)
        m.actuator_names = [x["name"] for x in acts]
        a["actuator_trnid"] = np.array([x["jnt"] for x in acts], dtype=np.int32)
        a["actuator_ctrllimited"] = np.array([x["ctrllimited"] for x in acts], dtype=np.int32)
        a["actuator_forcelimited"] = np.array([x["forcelimited"] for x in acts], dtype=np.int32)
        a["actuator_gear"] = np.array([x["gear"] for x in acts], dtype=float)
        a["actuator_ctrlrange"] = np.array([x["ctrlrange"] for x in acts], dtype=float).reshape(m.nu, 2)
        a["actuator_forcerange"] = np.array([x["forcerange"] for x in acts], dtype=float).reshape(m.nu, 2)

        # excludes + candidate pairs
        excludes = set()
        con = self.root.find("contact")
        if con is not None:
            for e in con:
                if e.tag == "exclude":
                    b1, b2 = m.body_id(e.get("body1")), m.body_id(e.get("body2"))
                    excludes.add((min(b1, b2), max(b1, b2)))
                else:
                    raise MjcfError(f"unsupported <contact> element <{e.tag}>")
        for t in self.root:
            if t.tag in ("equality", "tendon"):
                if len(t):
                    raise MjcfError(f"<{t.tag}> is not supported")
        g1, g2 = build_pairs(m, excludes)
        a["pair_geom1"], a["pair_geom2"] = g1, g2
        m.npair = len(g1)

        m.totalmass = float(mass.sum())
        set_const(m)
        return m


def build_pairs(m: Model, excludes: set) -> tuple[np.ndarray, np.ndarray]:
    """Static part of MuJoCo's collision filtering, in body-pair order.

    Restates engine_collision_driver.c ``mj_collision`` / ``filterBodyPair`` /
    ``mj_collideGeoms`` filters [MJ-recall]: same weld body, weld-parent (unless one side
    is static), ``<exclude>``, contype/conaffinity; the narrow phase is called with
    geom types ordered type1 <= type2.
    """
    a = m.arrays
    parent, weld = a["body_parentid"], a["body_weldid"]
    gb, gt = a["geom_bodyid"], a["geom_type"]
    ct, ca = a["geom_contype"], a["geom_conaffinity"]
    by_body = [[] for _ in range(m.nbody)]
    for g in range(m.ngeom):
        by_body[gb[g]].append(g)
    p1, p2 = [], []
    for b1 in range(m.nbody):
        for b2 in range(b1 + 1, m.nbody):
            if not by_body[b1] or not by_body[b2]:
                continue
            if (b1, b2) in excludes:
                continue
            w1, w2 = int(weld[b1]), int(weld[b2])
            if w1 == w2:
                continue
            wp1, wp2 = int(weld[parent[w1]]), int(weld[parent[w2]])
            if w1 != 0 and w2 != 0 and (w1 == wp2 or w2 == wp1):
                continue
            for ga in by_body[b1]:
                for gc in by_body[b2]:
                    if not ((ct[ga] & ca[gc]) or (ct[gc] & ca[ga])):
                        continue
                    x, y = (ga, gc) if gt[ga] <= gt[gc] else (gc, ga)
                    if gt[x] == GEOM_PLANE and gt[y] == GEOM_PLANE:
                        continue
                    if (int(gt[x]), int(gt[y])) not in SUPPORTED_PAIRS:
                        tn = lambda t: _TYPE_NAMES.get(int(t), str(int(t)))
                        raise MjcfError(
                            f"collision pair {m.geom_names[x]} ({tn(gt[x])}) / {m.geom_names[y]} ({tn(gt[y])}) needs a narrow phase that is not "
                            f"implemented (cylinders collide with planes and spheres, ellipsoids with planes only: MuJoCo resolves their other pairs "
                            f"through its general convex collider); mask the pair with contype / conaffinity or <exclude>, or replace the geom by a capsule")
                    p1.append(x)
                    p2.append(y)
    return np.array(p1, dtype=np.int32), np.array(p2, dtype=np.int32)


# ----------------------------------------------------------------------------- mj_setConst subset
def _kinematics0(m: Model):
    """Body frames, inertial frames and dof axes at qpos0 (numpy; compile-time only)."""
    a = m.arrays
    nb = m.nbody
    xpos = np.zeros((nb, 3))
    xquat = np.tile(np.array([1.0, 0, 0, 0]), (nb, 1))
    for i in range(1, nb):
        p = a["body_parentid"][i]
        R = quat2mat(xquat[p])
        xpos[i] = xpos[p] + R @ a["body_pos"][i]
        xquat[i] = quat_mul(xquat[p], a["body_quat"][i])
        xquat[i] /= np.linalg.norm(xquat[i])
    xipos = np.array([xpos[i] + quat2mat(xquat[i]) @ a["body_ipos"][i] for i in range(nb)])
    return xpos, xquat, xipos


def _jac_point(m: Model, xpos, xquat, body: int, point):
    """3 x nv translational and rotational Jacobians of ``point`` fixed to ``body`` at qpos0."""
    a = m.arrays
    Jt = np.zeros((3, m.nv))
    Jr = np.zeros((3, m.nv))
    b = body
    while b > 0:
        for k in range(a["body_jntnum"][b]):
            j = a["body_jntadr"][b] + k
            d = a["jnt_dofadr"][j]
            R = quat2mat(xquat[b])
            t = a["jnt_type"][j]
            if t == JNT_FREE:
                Jt[:, d:d + 3] = np.eye(3)
                for c in range(3):
                    ax = R[:, c]
                    Jr[:, d + 3 + c] = ax
                    Jt[:, d + 3 + c] = np.cross(ax, point - xpos[b])
            else:
                ax = R @ a["jnt_axis"][j]
                anchor = xpos[b] + R @ a["jnt_pos"][j]
                if t == JNT_HINGE:
                    Jr[:, d] = ax
                    Jt[:, d] = np.cross(ax, point - anchor)
                else:
                    Jt[:, d] = ax
        b = a["body_parentid"][b]
    return Jt, Jr


def mass_matrix0(m: Model) -> np.ndarray:
    """Joint-space inertia at qpos0 (incl. armature) via sum_b J^T I J."""
    a = m.arrays
    xpos, xquat, xipos = _kinematics0(m)
    M = np.zeros((m.nv, m.nv))
    for b in range(1, m.nbody):
        if a["body_mass"][b] <= 0:
            continue
        Jt, Jr = _jac_point(m, xpos, xquat, b, xipos[b])
        R = quat2mat(quat_mul(xquat[b], a["body_iquat"][b]))
        I = R @ np.diag(a["body_inertia"][b]) @ R.T
        M += a["body_mass"][b] * Jt.T @ Jt + Jr.T @ I @ Jr
    M += np.diag(a["dof_armature"])
    return M


def set_const(m: Model) -> None:
    """dof_invweight0, body_invweight0, meaninertia at qpos0 (MuJoCo ``mj_setConst``/``set0`` [MJ-recall])."""
    a = m.arrays
    nv = m.nv
    a["dof_invweight0"] = np.zeros(nv)
    a["body_invweight0"] = np.zeros((m.nbody, 2))
    a["geom_invweight0"] = np.zeros((m.ngeom, 2))
    if nv == 0:
        m.meaninertia = 1.0
        return
    M = mass_matrix0(m)
    m.meaninertia = float(np.trace(M) / nv)
    Minv = np.linalg.inv(M)
    diag = np.diag(Minv).copy()
    for j in range(m.njnt):
        d = a["jnt_dofadr"][j]
        if a["jnt_type"][j] == JNT_FREE:
            diag[d:d + 3] = diag[d:d + 3].mean()
            diag[d + 3:d + 6] = diag[d + 3:d + 6].mean()
    a["dof_invweight0"] = diag
    xpos, xquat, xipos = _kinematics0(m)
    for b in range(1, m.nbody):
        if a["body_weldid"][b] == 0:
            continue
        Jt, Jr = _jac_point(m, xpos, xquat, b, xipos[b])
        a["body_invweight0"][b, 0] = np.trace(Jt @ Minv @ Jt.T) / 3.0
        a["body_invweight0"][b, 1] = np.trace(Jr @ Minv @ Jr.T) / 3.0
    a["geom_invweight0"] = a["body_invweight0"][a["geom_bodyid"]].copy() if m.ngeom else np.zeros((0, 2))


# ----------------------------------------------------------------------------- public API
def compile_file(path: str, timestep: float | None = None) -> Model:
    """Compile an MJCF file. ``timestep`` overrides <option timestep> the way the reference
    does after compiling (reference envs/common/mujoco_env.py:33)."""
    tree = ET.parse(path)
    m = _Compiler(tree.getroot()).build()
    if timestep is not None:
        m.timestep = float(timestep)
    return m


def compile_string(xml: str, timestep: float | None = None) -> Model:
    m = _Compiler(ET.fromstring(xml)).build()
    if timestep is not None:
        m.timestep = float(timestep)
    return m
