"""Primitive narrow phases of the CPU oracle on configurations with closed-form answers, against the documented contact
conventions of MuJoCo (mjContact): dist = signed distance between the surfaces (negative = penetration), pos = midpoint
between them, frame[0] = normal pointing from geom1 to geom2, geoms ordered by type (plane < sphere < capsule < box)."""
import numpy as np

from learninghumanoidwalking_amd import mjcf
from oracle.physics import OracleSim


def _two(g1, g2, p2="0 0 1"):
    xml = f"""<mujoco><worldbody>
      <body name="a" pos="0 0 0"><freejoint/><geom name="a" {g1} mass="1"/></body>
      <body name="b" pos="{p2}"><freejoint/><geom name="b" {g2} mass="1"/></body>
    </worldbody></mujoco>"""
    m = mjcf.compile_string(xml)
    return m, OracleSim(m)


def _plane(g):
    xml = f"""<mujoco><worldbody><geom name="floor" type="plane" size="0 0 1"/>
      <body name="b" pos="0 0 1"><freejoint/><geom name="b" {g} mass="1"/></body></worldbody></mujoco>"""
    m = mjcf.compile_string(xml)
    return m, OracleSim(m)


def _set(s, pos, quat=(1, 0, 0, 0), adr=0):
    s.qpos[adr:adr + 3] = pos
    s.qpos[adr + 3:adr + 7] = np.asarray(quat, float) / np.linalg.norm(quat)


def _cons(s):
    s.qvel[:] = 0
    s.forward(False)
    return [s.contact(i) for i in range(s.ncon)]


Y90 = [np.cos(np.pi / 4), 0, np.sin(np.pi / 4), 0]      # local z -> world x


def test_plane_sphere_capsule_box():
    m, s = _plane('type="sphere" size="0.1"')
    _set(s, [0.3, -0.2, 0.1 - 0.004])
    (c,) = _cons(s)
    assert abs(c["dist"] + 0.004) < 1e-15 and c["geom1"] == m.geom_id("floor")
    np.testing.assert_allclose(c["frame"][0], [0, 0, 1], atol=1e-15)             # from the plane up into the sphere
    np.testing.assert_allclose(c["pos"], [0.3, -0.2, -0.002], atol=1e-15)        # midway between z = 0 and z = -0.004
    m, s = _plane('type="capsule" size="0.05 0.2"')
    _set(s, [0, 0, 0.05 - 0.002], Y90)                                           # lying along x: one contact per end
    cs = _cons(s)
    assert len(cs) == 2 and all(abs(c["dist"] + 0.002) < 1e-12 for c in cs)
    np.testing.assert_allclose(sorted(c["pos"][0] for c in cs), [-0.2, 0.2], atol=1e-12)
    _set(s, [0, 0, 0.2 + 0.05 - 0.003])                                          # upright: the lower cap only
    (c,) = _cons(s)
    assert abs(c["dist"] + 0.003) < 1e-12
    m, s = _plane('type="box" size="0.2 0.1 0.05"')
    _set(s, [0, 0, 0.05 - 0.001])                                                # flat: four corners
    cs = _cons(s)
    assert len(cs) == 4 and all(abs(c["dist"] + 0.001) < 1e-12 for c in cs)
    assert sorted((round(c["pos"][0], 6), round(c["pos"][1], 6)) for c in cs) == [(-0.2, -0.1), (-0.2, 0.1), (0.2, -0.1), (0.2, 0.1)]
    ang = 0.2
    _set(s, [0, 0, 0.2 * np.sin(ang) + 0.05 * np.cos(ang) - 0.001], [np.cos(ang / 2), 0, np.sin(ang / 2), 0])   # tilted about y: one edge
    cs = _cons(s)
    assert len(cs) == 2 and all(abs(c["dist"] + 0.001) < 1e-12 for c in cs)


def test_sphere_sphere_and_sphere_capsule():
    m, s = _two('type="sphere" size="0.1"', 'type="sphere" size="0.15"')
    _set(s, [0, 0, 0]); _set(s, [0.2, 0, 0.1], adr=7)
    (c,) = _cons(s)
    dvec = np.array([0.2, 0, 0.1]); n = dvec / np.linalg.norm(dvec)
    assert abs(c["dist"] - (np.linalg.norm(dvec) - 0.25)) < 1e-15
    np.testing.assert_allclose(c["frame"][0], n, atol=1e-15)                      # from geom1 (a) to geom2 (b)
    np.testing.assert_allclose(c["pos"], n * (0.1 + 0.5 * c["dist"]), atol=1e-15)
    m, s = _two('type="capsule" size="0.05 0.3"', 'type="sphere" size="0.1"')     # the sphere is geom1 (lower type id)
    _set(s, [0, 0, 0], Y90); _set(s, [0.1, 0.14, 0], adr=7)
    (c,) = _cons(s)
    assert c["geom1"] == m.geom_id("b") and abs(c["dist"] - (0.14 - 0.15)) < 1e-12
    np.testing.assert_allclose(c["frame"][0], [0, -1, 0], atol=1e-12)             # from the sphere towards the capsule axis
    _set(s, [0.3 + 0.1, 0.0, 0.09], adr=7)                                        # beyond the end: against the end cap
    (c,) = _cons(s)
    dv = np.array([0.3, 0, 0]) - np.array([0.4, 0, 0.09])
    assert abs(c["dist"] - (np.linalg.norm(dv) - 0.15)) < 1e-12


def test_capsule_capsule_crossed_and_parallel():
    m, s = _two('type="capsule" size="0.05 0.3"', 'type="capsule" size="0.04 0.3"')
    _set(s, [0, 0, 0], Y90)                                                       # a along x
    _set(s, [0.1, 0, 0.08], [np.cos(np.pi / 4), np.sin(np.pi / 4), 0, 0], adr=7)  # b along y (rotated about x), 0.08 above
    (c,) = _cons(s)
    assert abs(c["dist"] - (0.08 - 0.09)) < 1e-12
    np.testing.assert_allclose(c["frame"][0], [0, 0, 1], atol=1e-12)
    np.testing.assert_allclose(c["pos"], [0.1, 0, 0.05 + 0.5 * c["dist"]], atol=1e-12)
    _set(s, [0.05, 0, 0.085], Y90, adr=7)                                         # parallel, overlapping: two contacts
    cs = _cons(s)
    assert len(cs) == 2 and all(abs(c["dist"] - (0.085 - 0.09)) < 1e-12 for c in cs)
    xs = sorted(c["pos"][0] for c in cs)
    assert xs[0] >= -0.3 - 1e-9 and xs[1] <= 0.35 + 1e-9 and xs[1] - xs[0] > 0.4
    _set(s, [0, 0, 0.2], Y90, adr=7)                                              # apart: nothing
    assert _cons(s) == []


def test_plane_cylinder_and_sphere_cylinder():
    """round 6: the two cylinder pairs MuJoCo resolves analytically (mjc_PlaneCylinder, mjc_SphereCylinder; the restatement is from recall --
    what is held here is the documented contact convention on configurations with closed-form answers)."""
    r, hl = 0.1, 0.25
    m, s = _plane(f'type="cylinder" size="{r} {hl}"')
    # upright on the plane, 2 mm deep: the near cap is flat on it -> the deepest rim point plus the two triangle points, all at the same depth;
    # the far cap is out of range
    _set(s, [0.3, -0.2, hl - 0.002])
    cs = _cons(s)
    assert len(cs) == 3 and all(abs(c["dist"] + 0.002) < 1e-12 and c["geom1"] == m.geom_id("floor") for c in cs)
    for c in cs:
        np.testing.assert_allclose(c["frame"][0], [0, 0, 1], atol=1e-15)
        assert abs(np.hypot(c["pos"][0] - 0.3, c["pos"][1] + 0.2) - r) < 1e-12 and abs(c["pos"][2] + 0.001) < 1e-12      # on the rim, midway in depth
    ang = [np.arctan2(c["pos"][1] + 0.2, c["pos"][0] - 0.3) for c in cs]
    d = sorted((np.diff(sorted(ang)) % (2 * np.pi)).tolist())
    np.testing.assert_allclose(d, [2 * np.pi / 3, 2 * np.pi / 3], atol=1e-9)                                            # an equilateral triangle
    # lying on its side along x, 3 mm deep: one contact under each cap's lowest rim point
    _set(s, [0, 0, r - 0.003], Y90)
    cs = _cons(s)
    assert len(cs) == 2 and all(abs(c["dist"] + 0.003) < 1e-12 for c in cs)
    np.testing.assert_allclose(sorted(c["pos"][0] for c in cs), [-hl, hl], atol=1e-12)
    for c in cs:
        assert abs(c["pos"][1]) < 1e-12 and abs(c["pos"][2] + 0.0015) < 1e-12
    # tilted by 0.3 rad about y: the lowest point of the lower rim only
    t = 0.3
    zc = hl * np.cos(t) + r * np.sin(t) - 0.001
    _set(s, [0, 0, zc], [np.cos(t / 2), 0, np.sin(t / 2), 0])
    (c,) = _cons(s)
    assert abs(c["dist"] + 0.001) < 1e-12
    assert abs(abs(c["pos"][0]) - abs(-hl * np.sin(t) + r * np.cos(t))) < 1e-12
    _set(s, [0, 0, zc + 0.0011], [np.cos(t / 2), 0, np.sin(t / 2), 0])
    assert _cons(s) == []
    # sphere against the cylinder: beside the lateral surface, over a cap, off the rim -- the sphere is geom1 (lower type id)
    m, s = _two(f'type="cylinder" size="{r} {hl}"', 'type="sphere" size="0.05"')
    _set(s, [0, 0, 0]); _set(s, [r + 0.05 - 0.004, 0, 0.1], adr=7)                      # side
    (c,) = _cons(s)
    assert c["geom1"] == m.geom_id("b") and abs(c["dist"] + 0.004) < 1e-12
    np.testing.assert_allclose(c["frame"][0], [-1, 0, 0], atol=1e-12)                   # from the sphere towards the axis
    np.testing.assert_allclose(c["pos"], [r - 0.002, 0, 0.1], atol=1e-12)
    _set(s, [0.03, -0.02, hl + 0.05 - 0.003], adr=7)                                    # over the upper cap
    (c,) = _cons(s)
    assert abs(c["dist"] + 0.003) < 1e-12
    np.testing.assert_allclose(c["frame"][0], [0, 0, -1], atol=1e-12)
    np.testing.assert_allclose(c["pos"], [0.03, -0.02, hl - 0.0015], atol=1e-12)
    dx, dz = 0.03, 0.03                                                                 # off the rim: nearest point = the rim circle
    _set(s, [r + dx, 0, -(hl + dz)], adr=7)
    (c,) = _cons(s)
    gap = np.hypot(dx, dz) - 0.05
    assert abs(c["dist"] - gap) < 1e-12
    np.testing.assert_allclose(c["frame"][0], [-dx / np.hypot(dx, dz), 0, dz / np.hypot(dx, dz)], atol=1e-12)
    # the pairs MuJoCo sends through its general convex collider are refused with a message that names the geoms
    import pytest
    with pytest.raises(mjcf.MjcfError, match="cylinder"):
        _two(f'type="cylinder" size="{r} {hl}"', f'type="capsule" size="0.05 0.2"')
    with pytest.raises(mjcf.MjcfError, match="ellipsoid"):
        _two('type="ellipsoid" size="0.1 0.2 0.3"', 'type="sphere" size="0.05"')
    # inertia of a cylinder from its geom (user_objects.cc: m = rho pi r^2 2 l; I_zz = m r^2 / 2, I_xx = m (3 r^2 + (2 l)^2) / 12)
    m2 = mjcf.compile_string(f'<mujoco><compiler inertiafromgeom="true"/><worldbody><body name="c"><freejoint/><geom type="cylinder" size="{r} {hl}" density="1000"/></body></worldbody></mujoco>')
    mass = 1000 * np.pi * r * r * 2 * hl
    b = m2.body_id("c")
    assert abs(m2.arrays["body_mass"][b] - mass) < 1e-9
    np.testing.assert_allclose(sorted(m2.arrays["body_inertia"][b]), sorted([mass * r * r / 2, mass * (3 * r * r + 4 * hl * hl) / 12, mass * (3 * r * r + 4 * hl * hl) / 12]), rtol=1e-12)


def test_plane_ellipsoid():
    """round 6: plane-ellipsoid (mjc_PlaneConvex on a smooth geom, from recall): one contact at the ellipsoid's support point towards the plane.
    Closed forms: axis-aligned, the support point is the end of the semi-axis along the normal; rotated, the height of the lowest point of an
    ellipsoid is sqrt(sum_i (s_i R_zi)^2) below its centre."""
    sz = np.array([0.1, 0.2, 0.3])
    m, s = _plane('type="ellipsoid" size="0.1 0.2 0.3"')
    _set(s, [0.4, -0.1, sz[2] - 0.002])
    (c,) = _cons(s)
    assert c["geom1"] == m.geom_id("floor") and abs(c["dist"] + 0.002) < 1e-12
    np.testing.assert_allclose(c["frame"][0], [0, 0, 1], atol=1e-15)
    np.testing.assert_allclose(c["pos"], [0.4, -0.1, -0.001], atol=1e-12)
    _set(s, [0, 0, sz[2] + 1e-3])
    assert _cons(s) == []
    rs = np.random.default_rng(0)
    for _ in range(20):
        q = rs.normal(size=4); q /= np.linalg.norm(q)
        R = mjcf.quat2mat(q)
        h = float(np.sqrt(((sz * R[2, :]) ** 2).sum()))            # centre height at which the ellipsoid touches z = 0
        _set(s, [0.1, 0.2, h - 0.003], q)
        (c,) = _cons(s)
        assert abs(c["dist"] + 0.003) < 1e-12
        # the support point: the surface point whose outward normal is -z, i.e. x_local = -S^2 R^T z / |S R^T z|
        sup = np.array([0.1, 0.2, h - 0.003]) + R @ (-(sz ** 2) * R[2, :] / h)
        np.testing.assert_allclose(c["pos"], sup + np.array([0, 0, 0.0015]), atol=1e-12)
    # inertia of an ellipsoid from its geom: m = rho 4/3 pi a b c, I = m / 5 (b^2 + c^2, a^2 + c^2, a^2 + b^2)
    m2 = mjcf.compile_string('<mujoco><compiler inertiafromgeom="true"/><worldbody><body name="e"><freejoint/><geom type="ellipsoid" size="0.1 0.2 0.3" density="800"/></body></worldbody></mujoco>')
    mass = 800 * 4 / 3 * np.pi * 0.1 * 0.2 * 0.3
    b = m2.body_id("e")
    assert abs(m2.arrays["body_mass"][b] - mass) < 1e-9
    np.testing.assert_allclose(sorted(m2.arrays["body_inertia"][b]), sorted(mass / 5 * np.array([0.13, 0.10, 0.05])), rtol=1e-12)
