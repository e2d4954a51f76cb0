"""TEST INFRASTRUCTURE: independent cross-checks of the CPU oracle's physics (oracle/mjc_oracle.c).

The oracle restates MuJoCo's pipeline from recall and cannot be run against MuJoCo in this image ("parity unpinned",
DESIGN.md section 2).  What CAN be checked here is that its pieces agree with formulations that share no code and no
derivation path with them:

* `dual_pgs`       -- MuJoCo's soft-constraint problem solved through its DUAL by projected Gauss-Seidel (the formulation of
                      MuJoCo's PGS solver option): strictly convex, hence the same optimum as the primal
                      Newton method the oracle and the HIP kernels implement.  A wrong gradient / Hessian / line search /
                      active-set rule in the Newton code shows up as a different optimum.
* `kkt_residuals`  -- optimality conditions of the primal problem evaluated directly from (M, J, aref, D, frictionloss).
* `aba_qacc`       -- qacc_smooth by Featherstone's articulated-body algorithm written in world-origin spatial (Pluecker)
                      coordinates with its own kinematics, versus the oracle's com-based CRBA + RNE + dense factorisation.

None of this pins the *model* of contact (impedance, reference acceleration, pyramid regulariser): only a real MuJoCo can
(scripts/pin_vs_mujoco.py).  Only tests/ imports this module.
"""
from __future__ import annotations

import ctypes

import numpy as np

from .physics import OracleSim, lib

EFC_FRICTION, EFC_LIMIT, EFC_CONTACT = 0, 1, 2


def constraint_problem(s: OracleSim):
    """Matrices of the constraint problem of the oracle's last forward pass."""
    n = s.nefc
    return dict(M=np.array(s.M), J=s.efc("efc_J").reshape(n, -1), aref=s.efc("efc_aref"), R=s.efc("efc_R"), D=s.efc("efc_D"),
                floss=s.efc("efc_frictionloss"), kind=s.efc_types(), qacc_smooth=np.array(s.qacc_smooth),
                qfrc_smooth=np.array(s.qfrc_smooth), qacc=np.array(s.qacc), force=s.efc("efc_force"))


def dual_pgs(p, max_sweeps=200000, tol=1e-12):
    """Solve the dual by projected Gauss-Seidel.  Returns (force, qacc, sweeps)."""
    n = len(p["aref"])
    MinvJt = np.linalg.solve(p["M"], p["J"].T)
    AR = np.ascontiguousarray(p["J"] @ MinvJt + np.diag(p["R"]))
    b = np.ascontiguousarray(p["J"] @ p["qacc_smooth"] - p["aref"])
    f = np.zeros(n)
    floss = np.ascontiguousarray(p["floss"], dtype=np.float64)
    isf = np.ascontiguousarray(p["kind"] == EFC_FRICTION, dtype=np.int32)
    sweeps = lib().orc_dual_pgs(n, AR.ctypes.data, b.ctypes.data, floss.ctypes.data, isf.ctypes.data, f.ctypes.data, int(max_sweeps),
                                ctypes.c_double(tol))
    return f, p["qacc_smooth"] + MinvJt @ f, sweeps


def force_law(p, qacc):
    """efc_force implied by an acceleration: f = -D r on the active side of a limit / contact row, the Huber clamp for
    frictionloss rows (mj_constraintUpdate)."""
    r = p["J"] @ qacc - p["aref"]
    f = -p["D"] * r
    fr = p["kind"] == EFC_FRICTION
    return np.where(fr, np.clip(f, -p["floss"], p["floss"]), np.maximum(f, 0.0))


def kkt_residuals(p, qacc=None, force=None):
    """(stationarity, force-law) residuals: || M qacc - qfrc_smooth - J^T f ||_inf and || f - force_law(qacc) ||_inf."""
    qacc = p["qacc"] if qacc is None else qacc
    force = p["force"] if force is None else force
    stat = np.abs(p["M"] @ qacc - p["qfrc_smooth"] - p["J"].T @ force).max()
    law = np.abs(force - force_law(p, qacc)).max() if len(force) else 0.0
    return stat, law


# ---------------------------------------------------------------------------------------------------- articulated-body algorithm
def _quat2mat(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _axis_angle(ax, ang):
    ax = ax / np.linalg.norm(ax)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def _crm(v):   # spatial cross product for motion vectors [w; vO]
    out = np.zeros((6, 6))
    out[:3, :3] = _skew(v[:3]); out[3:, 3:] = _skew(v[:3]); out[3:, :3] = _skew(v[3:])
    return out


def aba_qacc(m, qpos, qvel, tau):
    """Forward dynamics qacc = FD(qpos, qvel, tau) of the kinematic tree of `m` (bodies with at most one joint: free, hinge
    or slide) under gravity, with joint armature, by the articulated-body algorithm.  Spatial vectors are [angular; linear
    velocity of the point at the world origin]; inertias are taken about the world origin.  `tau` holds every generalised
    force except the bias force (passive + actuator + applied)."""
    nb = m.nbody
    R = [np.eye(3)] * nb
    pos = [np.zeros(3)] * nb
    S = [np.zeros((6, 0))] * nb          # motion subspace of the body's joint
    dofs = [[] for _ in range(nb)]
    Sdot_qd = [np.zeros(6)] * nb         # sum_k dS_k/dt qd_k (velocity-product acceleration of the joint)
    v = [np.zeros(6)] * nb
    I = [np.zeros((6, 6))] * nb
    for b in range(1, nb):
        par = int(m.body_parentid[b])
        Rb = R[par] @ _quat2mat(m.body_quat[b])
        pb = pos[par] + R[par] @ m.body_pos[b]
        vb = v[par].copy()
        cb = np.zeros(6)
        if m.body_jntnum[b] == 1:
            j = int(m.body_jntadr[b]); t = int(m.jnt_type[j]); qa = int(m.jnt_qposadr[j]); da = int(m.jnt_dofadr[j])
            if t == 0:    # free: world-frame translations, then rotations about the body-frame axes through the body origin
                pb = np.array(qpos[qa:qa + 3], dtype=float)
                Rb = _quat2mat(np.array(qpos[qa + 3:qa + 7], dtype=float))
                cols = [np.concatenate([np.zeros(3), e]) for e in np.eye(3)]
                cols += [np.concatenate([Rb[:, k], np.cross(pb, Rb[:, k])]) for k in range(3)]
                Sb = np.array(cols).T
                dofs[b] = list(range(da, da + 6))
                qd = np.array(qvel[da:da + 6], dtype=float)
                omega = Rb @ qd[3:]
                cb = np.concatenate([np.zeros(3), np.cross(qd[:3], omega)])   # sum_k dS_k/dt qd_k = [0; pdot x omega]
                vb = v[par] + Sb @ qd
            else:
                anchor = pb + Rb @ m.jnt_pos[j]
                axis = Rb @ m.jnt_axis[j]
                q = qpos[qa] - m.qpos0[qa]
                if t == 3:   # hinge: rotate about the axis through the anchor
                    Rb = _axis_angle(axis, q) @ Rb
                    pb = anchor - Rb @ m.jnt_pos[j]
                    s = np.concatenate([axis, np.cross(anchor, axis)])
                else:        # slide
                    pb = pb + axis * q
                    s = np.concatenate([np.zeros(3), axis])
                Sb = s.reshape(6, 1)
                dofs[b] = [da]
                vj = s * qvel[da]
                cb = _crm(v[par]) @ vj                   # the axis is fixed in the parent: dS/dt = v_parent x S
                vb = v[par] + vj
            S[b] = Sb
        R[b], pos[b], v[b], Sdot_qd[b] = Rb, pb, vb, cb
        Ri = Rb @ _quat2mat(m.body_iquat[b])
        Ic = Ri @ np.diag(m.body_inertia[b]) @ Ri.T
        c = pb + Rb @ m.body_ipos[b]
        mass = float(m.body_mass[b])
        cx = _skew(c)
        Ib = np.zeros((6, 6))
        Ib[:3, :3] = Ic - mass * cx @ cx; Ib[:3, 3:] = mass * cx; Ib[3:, :3] = -mass * cx; Ib[3:, 3:] = mass * np.eye(3)
        I[b] = Ib
    # bias forces p = v x* (I v); gravity enters as a base acceleration of -g
    IA = [Ib.copy() for Ib in I]
    pA = [np.zeros(6)] * nb
    for b in range(1, nb):
        pA[b] = -_crm(v[b]).T @ (I[b] @ v[b])
    U, Dinv, u = [None] * nb, [None] * nb, [None] * nb
    for b in range(nb - 1, 0, -1):
        par = int(m.body_parentid[b])
        if dofs[b]:
            Sb = S[b]
            U[b] = IA[b] @ Sb
            Dm = Sb.T @ U[b] + np.diag(m.dof_armature[dofs[b]])
            Dinv[b] = np.linalg.inv(Dm)
            u[b] = tau[dofs[b]] - Sb.T @ pA[b]
            Ia = IA[b] - U[b] @ Dinv[b] @ U[b].T
            pa = pA[b] + Ia @ Sdot_qd[b] + U[b] @ Dinv[b] @ u[b]
        else:
            Ia, pa = IA[b], pA[b]
        if par > 0:
            IA[par] = IA[par] + Ia
            pA[par] = pA[par] + pa
    a = [np.zeros(6)] * nb
    a[0] = np.concatenate([np.zeros(3), -np.asarray(m.gravity, dtype=float)])
    qacc = np.zeros(m.nv)
    for b in range(1, nb):
        par = int(m.body_parentid[b])
        ap = a[par] + Sdot_qd[b]
        if dofs[b]:
            qdd = Dinv[b] @ (u[b] - U[b].T @ ap)
            qacc[dofs[b]] = qdd
            a[b] = ap + S[b] @ qdd
        else:
            a[b] = ap
    return qacc
