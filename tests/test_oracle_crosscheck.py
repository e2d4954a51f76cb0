"""Cross-checks of the CPU oracle's physics against independently formulated solvers (oracle/crosscheck.py): the primal Newton
optimum vs projected Gauss-Seidel on the dual of the same regularised problem, its KKT conditions, and qacc_smooth vs an
articulated-body algorithm.  These catch algebra / active-set / line-search mistakes in the Newton code and in the
CRBA + RNE + factorisation route; they do NOT pin the contact model's parameters against MuJoCo (nothing in this image can:
scripts/pin_vs_mujoco.py is the one-command pin for a box that has the wheel)."""
import numpy as np
import pytest

from learninghumanoidwalking_amd.envs.h1 import H1Spec
from learninghumanoidwalking_amd.envs.jvrc_walk import JvrcWalkSpec
from oracle import crosscheck as cc
from oracle.physics import OracleSim


def _poses(spec, rs, n_dof):
    """standing, fallen, tangled legs, joint limits violated on both sides, airborne -- as tests/test_jvrc_gpu.py"""
    m = spec.model()
    lo, hi = m.jnt_range[1:, 0], m.jnt_range[1:, 1]
    nom = spec.nominal_pose
    out = []
    q = nom.copy(); out.append(("standing", q, np.zeros(m.nv)))
    q = nom.copy(); q[2] -= 0.004; out.append(("pressed into the floor", q, rs.normal(size=m.nv) * 0.2))
    for k in range(3):
        q = nom.copy(); q[2] = rs.uniform(0.12, 0.4); quat = rs.normal(size=4); q[3:7] = quat / np.linalg.norm(quat)
        q[7:] = rs.uniform(lo, hi); out.append((f"fallen {k}", q, rs.normal(size=m.nv) * 0.3))
    q = nom.copy(); q[2] = 1.3; q[8], q[8 + n_dof // 2] = 0.3, -0.3; out.append(("legs crossed in the air", q, rs.normal(size=m.nv) * 0.3))
    q = nom.copy(); q[2] = 1.2
    q[7:] = np.where(rs.uniform(size=n_dof) < 0.5, lo - rs.uniform(0.02, 0.2, n_dof), hi + rs.uniform(0.02, 0.2, n_dof))
    out.append(("limits violated", q, rs.normal(size=m.nv) * 0.3))
    return out


def _sim(spec, tight=True, floss=None, damping=None):
    m = spec.model().copy()
    if tight:                      # run the Newton solver to machine precision so that the optimum itself is compared
        m.tolerance = 1e-15
        m.iterations = 200
    if floss is not None:
        m.arrays["dof_frictionloss"][6:] = floss
    if damping is not None:
        m.arrays["dof_damping"][6:] = damping
    return m, OracleSim(m)


@pytest.mark.parametrize("spec_cls,n_dof,floss", [(JvrcWalkSpec, 12, None), (H1Spec, 10, 1.3)])
def test_newton_optimum_equals_dual_pgs_optimum_and_satisfies_kkt(spec_cls, n_dof, floss):
    spec = spec_cls()
    rs = np.random.default_rng(7)
    m, s = _sim(spec, floss=floss)
    kinds = set()
    for name, q, v in _poses(spec, rs, n_dof):
        s.reset_data()
        s.qpos[:] = q; s.qvel[:] = v
        s.ctrl[:] = rs.normal(size=m.nu) * 5.0
        s.forward(True)
        if s.nefc == 0:
            continue
        p = cc.constraint_problem(s)
        kinds |= set(int(k) for k in p["kind"])
        scale = 1.0 + np.abs(p["qacc"]).max()
        stat, law = cc.kkt_residuals(p)
        assert stat < 1e-9 * (1.0 + np.abs(p["qfrc_smooth"]).max()), (name, stat)
        assert law < 1e-10 * (1.0 + np.abs(p["force"]).max()), (name, law)
        f, qacc, sweeps = cc.dual_pgs(p)
        assert sweeps < 200000, (name, "PGS did not converge")
        np.testing.assert_allclose(qacc, p["qacc"], rtol=0, atol=1e-8 * scale, err_msg=f"{name}: qacc, PGS ({sweeps} sweeps) vs Newton")
        np.testing.assert_allclose(f, p["force"], rtol=0, atol=1e-8 * (1.0 + np.abs(p["force"]).max()), err_msg=f"{name}: efc_force")
        # the dual optimum satisfies the primal KKT conditions too
        stat2, law2 = cc.kkt_residuals(p, qacc, f)
        assert stat2 < 1e-8 * (1.0 + np.abs(p["qfrc_smooth"]).max()) and law2 < 1e-7 * (1.0 + np.abs(f).max()), (name, stat2, law2)
    assert cc.EFC_CONTACT in kinds
    if floss is None:
        assert cc.EFC_LIMIT in kinds        # (the H1 stand-in has no limited joints, like the reference's jointlimited=false)
    else:
        assert cc.EFC_FRICTION in kinds


def test_default_tolerance_newton_is_within_solver_tolerance_of_the_optimum():
    """With MuJoCo's default tolerance (1e-8, scaled) the oracle stops close to the optimum the dual solver finds."""
    spec = JvrcWalkSpec()
    rs = np.random.default_rng(3)
    m, s = _sim(spec, tight=False)
    for name, q, v in _poses(spec, rs, 12):
        s.reset_data()
        s.qpos[:] = q; s.qvel[:] = v
        s.forward(False)
        if s.nefc == 0:
            continue
        p = cc.constraint_problem(s)
        _, qacc, _ = cc.dual_pgs(p)
        assert np.abs(qacc - p["qacc"]).max() < 1e-4 * (1.0 + np.abs(p["qacc"]).max()), name


@pytest.mark.parametrize("spec_cls,n_dof", [(JvrcWalkSpec, 12), (H1Spec, 10)])
def test_qacc_smooth_equals_articulated_body_algorithm(spec_cls, n_dof):
    spec = spec_cls()
    rs = np.random.default_rng(11)
    m, s = _sim(spec, tight=False, damping=0.7)
    for name, q, v in _poses(spec, rs, n_dof):
        s.reset_data()
        s.qpos[:] = q; s.qvel[:] = v * 3.0
        s.ctrl[:] = rs.normal(size=m.nu) * 5.0
        s.forward(True)
        tau = np.array(s.qfrc_smooth) + np.array(s.qfrc_bias)       # passive + actuator (+ applied): everything but the bias force
        qacc = cc.aba_qacc(m, np.array(s.qpos), np.array(s.qvel), tau)
        ref = np.array(s.qacc_smooth)
        np.testing.assert_allclose(qacc, ref, rtol=0, atol=1e-9 * (1.0 + np.abs(ref).max()), err_msg=name)


def _rot(q):
    from learninghumanoidwalking_amd import mjcf
    return mjcf.quat2mat(np.asarray(q, float))


def _integrate_pos(m, q, v, eps):
    """mj_integratePos for a free root + hinges: positions advance by eps * v, the root quaternion by the body-frame rotation"""
    q2 = np.array(q, float)
    q2[0:3] += eps * v[0:3]
    w = v[3:6] * eps
    ang = np.linalg.norm(w)
    if ang > 0:
        ax = w / ang
        dq = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])
        a, b = q2[3:7], dq
        q2[3:7] = [a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                   a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]]
        q2[3:7] /= np.linalg.norm(q2[3:7])
    q2[7:] += eps * v[6:]
    return q2


@pytest.mark.parametrize("spec_cls,n_dof", [(JvrcWalkSpec, 12), (H1Spec, 10)])
def test_contact_jacobian_equals_finite_differences_of_the_geometry(spec_cls, n_dof):
    """Rows of efc_J for a contact are Jn +- mu Jt1, Jn +- mu Jt2 (pyramid edges).  Jn v must equal the rate at which the two
    bodies' material points at the contact separate along the normal when the configuration moves with velocity v --
    computed here from the KINEMATICS alone (central differences of xpos / xmat), independent of the Jacobian code."""
    spec = spec_cls()
    rs = np.random.default_rng(5)
    m, s = _sim(spec, tight=False)
    checked = 0
    for name, q, v in _poses(spec, rs, n_dof):
        s.reset_data()
        s.qpos[:] = q; s.qvel[:] = 0
        s.forward(False)
        if s.ncon == 0:
            continue
        J = np.array(s.efc("efc_J")).reshape(s.nefc, m.nv)
        cons = [s.contact(i) for i in range(s.ncon)]
        xpos0, xmat0 = np.array(s.xpos), np.array(s.xmat).reshape(-1, 3, 3)
        vel = rs.normal(size=m.nv)
        eps = 1e-6
        frames = []
        for sign in (+1, -1):
            s.reset_data()
            s.qpos[:] = _integrate_pos(m, q, vel, sign * eps); s.qvel[:] = 0
            s.forward(False)
            frames.append((np.array(s.xpos), np.array(s.xmat).reshape(-1, 3, 3)))
        for c in cons:
            if c["efc_address"] < 0:
                continue
            a = c["efc_address"]
            b1, b2 = int(m.arrays["geom_bodyid"][c["geom1"]]), int(m.arrays["geom_bodyid"][c["geom2"]])
            n, t1, t2 = c["frame"][0], c["frame"][1], c["frame"][2]
            rate = []
            for b in (b1, b2):                       # velocity of the body-fixed point that sits at the contact position
                loc = xmat0[b].T @ (c["pos"] - xpos0[b])
                pp = frames[0][0][b] + frames[0][1][b] @ loc
                pm = frames[1][0][b] + frames[1][1][b] @ loc
                rate.append((pp - pm) / (2 * eps))
            rel = rate[1] - rate[0]                  # body 2 relative to body 1; the normal points from geom1 to geom2
            jn = 0.5 * (J[a] + J[a + 1]) @ vel
            jt1 = (J[a] - J[a + 1]) @ vel / (2 * c["mu"])
            jt2 = (J[a + 2] - J[a + 3]) @ vel / (2 * c["mu"])
            np.testing.assert_allclose([jn, jt1, jt2], [n @ rel, t1 @ rel, t2 @ rel], rtol=0, atol=2e-6, err_msg=f"{name}: contact at {c['pos']}")
            checked += 1
    assert checked >= 10


def test_free_flight_conserves_linear_and_angular_momentum():
    """No gravity, no contact with the world, no damping / frictionloss, internal actuator torques (and leg-leg contacts) only:
    the total linear momentum and the angular momentum about the origin, computed from body kinematics (xipos, ximat, the
    com-based cvel), are constants of the motion while the limbs thrash.  Semi-implicit Euler conserves them to O(h): the
    drift over a fixed time must be small and halve with the time step -- a check of CRBA + RNE + integration that uses none
    of them."""
    spec = JvrcWalkSpec()
    phase = np.random.default_rng(9).uniform(0, 6.28, 12)

    def drift(h):
        m = spec.model().copy()
        m.gravity = np.zeros(3)
        m.arrays["dof_damping"][:] = 0
        m.arrays["dof_frictionloss"][:] = 0
        m.arrays["jnt_limited"][:] = 0
        m.timestep = h
        s = OracleSim(m)
        s.qpos[:] = spec.nominal_pose
        s.qpos[2] = 5.0
        s.qvel[:] = np.random.default_rng(3).normal(size=m.nv) * 0.5

        def momentum():
            s.forward(True)
            com = np.array(s.subtree_com)[1]
            P, Lang = np.zeros(3), np.zeros(3)
            for b in range(1, m.nbody):
                mass = m.arrays["body_mass"][b]
                if mass == 0 or m.arrays["body_rootid"][b] != 1:
                    continue
                w, vc = np.array(s.cvel)[b][:3], np.array(s.cvel)[b][3:]
                r = np.array(s.xipos)[b]
                vb = vc + np.cross(w, r - com)                      # com-based spatial velocity -> velocity of the body's own com
                R = np.array(s.ximat)[b].reshape(3, 3)
                P += mass * vb
                Lang += R @ np.diag(m.arrays["body_inertia"][b]) @ R.T @ w + mass * np.cross(r, vb)
            return P, Lang

        P0, L0 = momentum()
        for k in range(int(round(0.1 / h))):
            s.ctrl[:] = 0.15 * np.sin(25.0 * k * h + phase)          # x gear 100: +-15 N m
            s.step()
            for i in range(s.ncon):      # leg-leg contacts are internal forces (momentum-neutral); nothing may touch the world
                c = s.contact(i)
                assert m.arrays["body_rootid"][m.arrays["geom_bodyid"][c["geom1"]]] == 1 and m.arrays["body_rootid"][m.arrays["geom_bodyid"][c["geom2"]]] == 1
        P1, L1 = momentum()
        return np.abs(P1 - P0).max() / (1 + np.abs(P0).max()), np.abs(L1 - L0).max() / (1 + np.abs(L0).max())

    (p1, l1), (p2, l2), (p4, l4) = drift(0.001), drift(0.0005), drift(0.00025)
    assert p1 < 5e-3 and l1 < 5e-3, (p1, l1)
    assert 1.6 < p1 / p2 < 2.4 and 1.6 < p2 / p4 < 2.4, (p1, p2, p4)       # first-order in h, as the integrator is
    assert 1.6 < l1 / l2 < 2.4 and 1.6 < l2 / l4 < 2.4, (l1, l2, l4)
