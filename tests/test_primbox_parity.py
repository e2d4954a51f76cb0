"""Sphere-box and capsule-box narrow phases of the HIP stepper against the CPU oracle.  The stand-in JVRC masks those pairs
(contype / conaffinity); this test unmasks them in a copy of the model -- the foot boxes also collide with the other leg's
collision primitives, as in the reference's gen_xml export where every collision geom has contype = conaffinity = 1
(envs/jvrc/gen_xml.py:98-122), and the left shin's capsule becomes a sphere so that sphere-box occurs -- and steps poses
(found offline with the oracle) in which a foot is pushed into the other leg.  Runs on the SIMT emulator in the CPU suite and
on the GPU in the `-m gpu` suite."""
import numpy as np
import pytest

from learninghumanoidwalking_amd.envs.jvrc_walk import JVRC_STANDIN_XML, JvrcWalkSpec

# joint angles (root held at z = 1.3, upright): three with a foot box against the shin sphere, three against a capsule
POSES = [
    [-0.3731, -0.2763, -0.4324, 2.2088, -0.2468, -0.9156, 0.1164, -0.1075, 0.0349, 1.6857, 0.2412, -1.1984],
    [-1.7754, -0.6924, -0.3899, 2.12, -0.1647, 0.7249, -0.8339, -0.2887, 0.101, 0.4117, -0.3301, -1.2423],
    [-0.2782, -0.7053, -0.4767, 2.129, 0.1075, 0.0397, 0.0517, -0.2387, 0.1999, 1.2379, -0.3872, -0.9754],
    [-0.3733, 0.0884, -0.3417, 2.1752, -0.55, -1.1568, -1.1245, 0.127, 0.4925, 2.1527, 0.4532, 0.5058],
    [-0.353, -0.191, 0.3282, 0.9963, 0.446, -0.2255, -0.7649, -0.3376, 0.2823, 1.4554, -0.2289, -0.1259],
    [0.2766, -0.0379, -0.2005, 2.264, 0.1498, 0.3834, -1.2521, -0.3198, 0.5014, 2.3812, 0.0107, 0.1214],
]


def _spec(tmp_path):
    xml = open(JVRC_STANDIN_XML).read()
    shin = '<geom name="L_KNEE_S-geom" type="capsule" size="0.045" fromto="0.01 0 -0.06 0.035 0 -0.28" contype="2" conaffinity="3"/>'
    assert shin in xml and xml.count('contype="0" conaffinity="1"') == 2
    xml = xml.replace('contype="0" conaffinity="1"', 'contype="0" conaffinity="3"')
    xml = xml.replace(shin, '<geom name="L_KNEE_S-geom" type="sphere" size="0.075" pos="0.02 0 -0.2" contype="2" conaffinity="3"/>')
    path = tmp_path / "jvrc_primbox.xml"
    path.write_text(xml)
    return JvrcWalkSpec(xml_path=str(path))


def _run(spec, env, step, get_state, set_state):
    from oracle.env_jvrc_walk import OracleJvrcWalkEnv
    N = len(POSES)
    m = spec.model()
    orc = [OracleJvrcWalkEnv(spec, seed=3, env_id=i) for i in range(N)]
    for o in orc:
        o.reset()
    q = np.tile(spec.nominal_pose, (N, 1))
    q[:, 2] = 1.3
    q[:, 7:] = POSES
    v = np.zeros((N, 18))
    set_state(q, v)
    for i, o in enumerate(orc):
        o.set_state(q[i], v[i])
    rs = np.random.default_rng(2)
    act = (rs.normal(size=(3, N, 12)) * 0.2).astype(np.float32)
    kinds = set()
    for t in range(3):
        step(act[t])
        for i, o in enumerate(orc):
            o.step(act[t, i])
            for k in range(o.sim.ncon):
                c = o.sim.contact(k)
                kinds.add((int(m.geom_type[c["geom1"]]), int(m.geom_type[c["geom2"]])))
        gq, gv = get_state()
        oq = np.array([o.sim.qpos.copy() for o in orc]); ov = np.array([o.sim.qvel.copy() for o in orc])
        np.testing.assert_allclose(gq, oq, rtol=0, atol=1e-11, err_msg=f"qpos t={t}")
        np.testing.assert_allclose(gv, ov, rtol=0, atol=1e-8, err_msg=f"qvel t={t}")
        set_state(oq, ov)
        for o in orc:
            o.set_state(o.sim.qpos.copy(), o.sim.qvel.copy())
    assert (2, 6) in kinds and (3, 6) in kinds, kinds
    assert env.pop_fault_stats() == (0, 0)


def test_sphere_box_and_capsule_box_on_the_emulator(tmp_path):
    from tests import emu
    spec = _spec(tmp_path)
    env = emu.make_emulated(spec, len(POSES), seed=3)
    env.reset()
    _run(spec, env, lambda a: env.step(a), env.get_state, env.set_state)


@pytest.mark.gpu
def test_sphere_box_and_capsule_box_on_the_gpu(tmp_path):
    import torch
    spec = _spec(tmp_path)
    env = spec.make_batched(len(POSES), seed=3, device=0)
    env.reset()
    _run(spec, env, lambda a: env.step(torch.from_numpy(a).cuda()), env.get_state, env.set_state)


@pytest.mark.gpu
def test_create_refuses_pairs_without_a_narrow_phase():
    """A model that reaches lhw_env_create without passing mjcf.build_pairs (pack_from_mjmodel) is checked there."""
    from learninghumanoidwalking_amd import _lib
    spec = JvrcWalkSpec()
    m = spec.model()
    g = int(m.pair_geom2[0])
    old = int(m.arrays["geom_type"][g])
    m.arrays["geom_type"][g] = 5          # mjGEOM_CYLINDER
    try:
        with pytest.raises(_lib.LhwError) as err:
            spec.make_batched(2, seed=0, device=0)
        assert "narrow phase" in str(err.value)
    finally:
        m.arrays["geom_type"][g] = old
