"""Minimal stand-in for the `transforms3d` package (absent from this image), used ONLY by gen_golden.py so that the
reference's tasks/stepping_task.py can be imported and executed.  Implements, from transforms3d's documented conventions
(quaternions w,x,y,z; default Euler axes 'sxyz' = static x, y, z), exactly the six functions that file calls:
euler.euler2quat, euler.quat2euler, euler.euler2mat, euler.mat2euler, quaternions.quat2mat, affines.compose.
The stepping goldens therefore pin "reference task code + this stand-in", which is stated in DESIGN.md."""
import math
import types

import numpy as np

_EPS4 = np.finfo(float).eps * 4.0


def quat2mat(q):
    w, x, y, z = q
    Nq = w * w + x * x + y * y + z * z
    if Nq < np.finfo(float).eps:
        return np.eye(3)
    s = 2.0 / Nq
    X, Y, Z = x * s, y * s, z * s
    wX, wY, wZ = w * X, w * Y, w * Z
    xX, xY, xZ = x * X, x * Y, x * Z
    yY, yZ, zZ = y * Y, y * Z, z * Z
    return np.array([[1.0 - (yY + zZ), xY - wZ, xZ + wY], [xY + wZ, 1.0 - (xX + zZ), yZ - wX], [xZ - wY, yZ + wX, 1.0 - (xX + yY)]])


def euler2mat(ai, aj, ak):
    si, sj, sk = math.sin(ai), math.sin(aj), math.sin(ak)
    ci, cj, ck = math.cos(ai), math.cos(aj), math.cos(ak)
    cc, cs, sc, ss = ci * ck, ci * sk, si * ck, si * sk
    return np.array([[cj * ck, sj * sc - cs, sj * cc + ss], [cj * sk, sj * ss + cc, sj * cs - sc], [-sj, cj * si, cj * ci]])


def mat2euler(M):
    M = np.asarray(M, dtype=float)[:3, :3]
    cy = math.sqrt(M[0, 0] * M[0, 0] + M[1, 0] * M[1, 0])
    if cy > _EPS4:
        return math.atan2(M[2, 1], M[2, 2]), math.atan2(-M[2, 0], cy), math.atan2(M[1, 0], M[0, 0])
    return math.atan2(-M[1, 2], M[1, 1]), math.atan2(-M[2, 0], cy), 0.0


def euler2quat(ai, aj, ak):
    ai, aj, ak = ai / 2.0, aj / 2.0, ak / 2.0
    ci, si, cj, sj, ck, sk = math.cos(ai), math.sin(ai), math.cos(aj), math.sin(aj), math.cos(ak), math.sin(ak)
    cc, cs, sc, ss = ci * ck, ci * sk, si * ck, si * sk
    return np.array([cj * cc + sj * ss, cj * sc - sj * cs, cj * ss + sj * cc, cj * cs - sj * sc])


def quat2euler(q):
    return mat2euler(quat2mat(q))


def compose(T, R, Z):
    A = np.eye(4)
    A[:3, :3] = np.asarray(R) * np.asarray(Z)[None, :]
    A[:3, 3] = T
    return A


def install(sys_modules):
    tf3 = types.ModuleType("transforms3d")
    tf3.euler = types.ModuleType("transforms3d.euler")
    tf3.quaternions = types.ModuleType("transforms3d.quaternions")
    tf3.affines = types.ModuleType("transforms3d.affines")
    tf3.euler.euler2quat, tf3.euler.quat2euler, tf3.euler.euler2mat, tf3.euler.mat2euler = euler2quat, quat2euler, euler2mat, mat2euler
    tf3.quaternions.quat2mat = quat2mat
    tf3.affines.compose = compose
    for name, mod in (("transforms3d", tf3), ("transforms3d.euler", tf3.euler), ("transforms3d.quaternions", tf3.quaternions),
                      ("transforms3d.affines", tf3.affines)):
        sys_modules[name] = mod
