"""A stand-in JVRC edit with cylinder and ellipsoid geoms, shared by the emulator test and its GPU twin: the shanks become cylinders that
collide with the floor (plane-cylinder, MuJoCo engine_collision_primitive.c mjc_PlaneCylinder) and with a ball added at each
ankle (sphere-cylinder, mjc_SphereCylinder), the thighs ellipsoids that collide with the floor (plane-ellipsoid, mjc_PlaneConvex);
the contype / conaffinity masks keep every other pair among the supported ones."""
import numpy as np


def cylinder_spec(tmp_path):
    from learninghumanoidwalking_amd.envs.jvrc_walk import JVRC_STANDIN_XML, JvrcWalkSpec
    xml = open(JVRC_STANDIN_XML).read()
    for side in "RL":
        shank = (f'<geom name="{side}_KNEE_S-geom" type="capsule" size="0.045" fromto="0.01 0 -0.06 0.035 0 -0.28" '
                 'contype="2" conaffinity="3"/>')
        ankle = ('<inertial pos="0 0 0" mass="1.0" diaginertia="0.0015 0.0015 0.0015"/>\n'
                 f'                <body name="{side}_ANKLE_P_S"')
        thigh = f'<geom name="{side}_HIP_Y_S-geom" type="capsule" size="0.05" fromto="0 0 -0.09 0 0 -0.30" contype="2" conaffinity="3"/>'
        assert shank in xml and ankle in xml and thigh in xml
        xml = xml.replace(thigh, f'<geom name="{side}_HIP_Y_S-geom" type="ellipsoid" size="0.06 0.05 0.15" pos="0 0 -0.195" contype="0" conaffinity="1"/>')
        xml = xml.replace(shank, shank.replace('type="capsule"', 'type="cylinder"')
                          .replace('contype="2" conaffinity="3"', 'contype="4" conaffinity="1"'))
        xml = xml.replace(ankle, ankle.replace(
            '\n', f'\n                <geom name="{side}_ankle_ball" type="sphere" size="0.05" contype="2" conaffinity="7"/>\n', 1))
    path = tmp_path / "jvrc_cylinder.xml"
    path.write_text(xml)
    return JvrcWalkSpec(xml_path=str(path))


def contact_kinds(m, sim):
    return {(int(m.geom_type[sim.contact(k)["geom1"]]), int(m.geom_type[sim.contact(k)["geom2"]])) for k in range(sim.ncon)}


def cylinder_poses(spec, probe, n, seed=5, max_trials=20000, max_con=9):
    """`n` random fallen poses (qpos rows) found with the oracle env `probe`: a third each with a sphere-cylinder, a plane-cylinder and a
    plane-ellipsoid contact; at most `max_con` contacts each (the walking layout holds 16 per env, and the fall adds some)."""
    m = spec.model()
    rs = np.random.default_rng(seed)
    lo, hi = m.jnt_range[1:, 0], m.jnt_range[1:, 1]
    want = [(2, 5)] * (n // 3) + [(0, 5)] * (n // 3) + [(0, 4)] * (n - 2 * (n // 3))
    out = []
    for _ in range(max_trials):
        q = np.array(spec.nominal_pose, float)
        q[2] = rs.uniform(0.05, 0.9)
        quat = rs.normal(size=4)
        q[3:7] = quat / np.linalg.norm(quat)
        q[7:] = rs.uniform(lo, hi)
        probe.set_state(q, np.zeros(m.nv))
        if probe.sim.ncon <= max_con and want[len(out)] in contact_kinds(m, probe.sim):
            out.append(q)
            if len(out) == n:
                return np.array(out)
    raise AssertionError("no pose with the wanted cylinder contacts found")
