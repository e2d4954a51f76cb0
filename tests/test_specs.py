"""Host-side task specs and the oracle's task layer against fixtures generated from the reference (CPU)."""
import os
import subprocess
import sys

import numpy as np

from learninghumanoidwalking_amd.envs.jvrc_walk import JvrcWalkSpec, phase_clock_lut

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_clock_lut_is_bit_exact_with_reference_pchip():
    g = np.load(os.path.join(G, "rewards.npz"))
    spec = JvrcWalkSpec()
    assert spec.period == 88 and spec.frame_skip == 25
    np.testing.assert_array_equal(spec.clock_lut(), g["jvrc_lut"])
    np.testing.assert_array_equal(phase_clock_lut(0.4, 0.1, 0.1, 40, 40), g["h1_lut"])
    lut = spec.clock_lut()
    dbl = [p for p in range(88) if lut[0, p] == 1 and lut[2, p] == 1]
    assert dbl == list(range(32, 43)) + list(range(76, 87))      # SURVEY.md 8c known answer


def test_mirror_tables_equal_reference_matrices():
    g = np.load(os.path.join(G, "misc.npz"))
    (osrc, osign), (asrc, asign) = JvrcWalkSpec().mirror_tables()
    x = np.random.default_rng(0).normal(size=(4, 37))
    ref = x @ g["mir_obs"]
    ref[:, [29, 30]] *= -1            # clock entries: sin(arcsin(c) + pi) == -c
    np.testing.assert_allclose(x[:, osrc] * osign, ref, atol=0)
    a = np.random.default_rng(1).normal(size=(4, 12))
    np.testing.assert_allclose(a[:, asrc] * asign, a @ g["mir_act"], atol=0)


def test_mirror_tables_of_h1_walk_and_jvrc_step_equal_reference_matrices():
    from learninghumanoidwalking_amd.envs import H1WalkSpec, JvrcStepSpec
    g = np.load(os.path.join(G, "misc.npz"))
    for spec, D, A, mo, ma, clock in ((H1WalkSpec(), 43, 10, g["mir_obs_h1walk"], g["mir_act_h1walk"], [35, 36]),
                                      (JvrcStepSpec(), 39, 12, g["mir_obs_step"], g["mir_act"], [29, 30])):
        (osrc, osign), (asrc, asign) = spec.mirror_tables()
        x = np.random.default_rng(0).normal(size=(4, D))
        ref = x @ mo
        ref[:, clock] *= -1
        np.testing.assert_allclose(x[:, osrc] * osign, ref, atol=0)
        a = np.random.default_rng(1).normal(size=(4, A))
        np.testing.assert_allclose(a[:, asrc] * asign, a @ ma, atol=0)
        assert spec.mirror_inds()[2] == clock and spec.obs_mean.shape == (D,) and spec.obs_std.shape == (D,)


def test_h1_walk_clock_matches_reference_spline_table():
    from learninghumanoidwalking_amd.envs import H1WalkSpec
    g = np.load(os.path.join(G, "rewards.npz"))
    s = H1WalkSpec()
    assert s.period == 40
    np.testing.assert_allclose(s.clock_lut(), g["h1_lut"], rtol=0, atol=1e-15)


def test_oracle_reward_terms_equal_reference_functions():
    from oracle import env_jvrc_walk as e
    g = np.load(os.path.join(G, "rewards.npz"))
    i = g["inputs"]
    qvel, qacc, tq, ptq, a, pa = i[:18], i[18:36], i[36:48], i[48:60], i[60:72], i[72:84]
    mine = np.array([
        e.r_fwd_vel(np.array([0.3, -0.1]), np.array([0.2, 0.0])), e.r_yaw_vel(0.37, 0.1), e.r_action(a, pa), e.r_torque(tq, ptq),
        e.r_height(0.77, 0.8, 0.2, 0.005), e.r_height(0.795, 0.8, 0.0, 0.0), e.r_root_accel(qvel, qacc),
        e.r_clock(120.0, 500.0, -1.0, 0.5, 62.0 * 9.8 * 0.5),
        e.r_clock(np.linalg.norm([0.1, 0, 0.05]), np.linalg.norm([0.3, 0.1, 0]), 1.0, -0.25, 0.2)])
    np.testing.assert_allclose(mine, g["terms"], rtol=0, atol=1e-15)


def test_oracle_env_surface_like_reference_tests():
    """Shape / finiteness / reward-sum checks of reference tests/test_environments.py:39-188 on the oracle env."""
    from oracle.env_jvrc_walk import make_oracle_jvrc_walk
    env = make_oracle_jvrc_walk(seed=3)
    obs = env.reset()
    assert obs.shape == (37,) and np.isfinite(obs).all()
    rs = np.random.default_rng(0)
    for _ in range(20):
        obs, r, done, info = env.step(rs.normal(size=12) * 0.2)
        assert obs.shape == (37,) and np.isfinite(obs).all() and isinstance(done, bool)
        assert abs(r - sum(info.values())) < 1e-6 and len(info) == 10
        if done:
            env.reset()


def test_device_rng_header_matches_python_restatement(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include <stdio.h>\n#include "%s/learninghumanoidwalking_amd/csrc/lhw_rng.h"\nint main(){'
                   'for(unsigned e=0;e<3;e++)for(unsigned c=0;c<3;c++)for(unsigned s=0;s<3;s++)'
                   'printf("%%llu %%.17g %%d\\n",(unsigned long long)lhw_rng_bits(12345ull,e,2,c*1000+7,s),'
                   'lhw_rng_uniform(12345ull,e,1,c,s,-0.5,0.5),lhw_rng_randint(99ull,e,2,c,s,88));return 0;}' % ROOT)
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-O1", "-o", str(exe), str(src), "-lm"])
    out = subprocess.check_output([str(exe)]).decode().split("\n")
    from oracle import rng
    k = 0
    for e in range(3):
        for c in range(3):
            for s in range(3):
                b, u, r = out[k].split()
                assert int(b) == rng.bits(12345, e, 2, c * 1000 + 7, s)
                assert float(u) == rng.uniform(12345, e, 1, c, s, -0.5, 0.5)
                assert int(r) == rng.randint(99, e, 2, c, s, 88)
                k += 1


def test_jvrc_yaml_keys_of_base_humanoid_env_follow_the_reference_key_by_key(tmp_path):
    """BaseHumanoidEnv's generic hooks on a JVRC env, as the reference's code treats them (envs/common/base_humanoid_env.py:76-92,
    247-338; envs/jvrc/jvrc_base.py:133-138; envs/common/domain_randomization.py:44): observation_noise is never applied by the JVRC
    robot state -> accepted and ignored; dynamics_randomization fails there (no body named 'pelvis') -> refused; init_noise and perturbation
    run there and here."""
    import pytest
    import yaml
    from learninghumanoidwalking_amd.envs.jvrc_walk import JVRC_BASE_YAML
    base = yaml.safe_load(open(JVRC_BASE_YAML))

    def spec_with(**extra):
        p = tmp_path / ("cfg_" + "_".join(extra) + ".yaml")
        p.write_text(yaml.safe_dump(dict(base, **extra)))
        return JvrcWalkSpec(yaml_path=str(p))

    s = spec_with(observation_noise=dict(enabled=True, type="uniform", multiplier=1.0, scales=dict(root_orient=0.05)))
    assert s.obs_dim == 37
    with pytest.raises(NotImplementedError):
        spec_with(dynamics_randomization=dict(enable=True, interval=0.5))
    with pytest.raises(NotImplementedError):
        spec_with(perturbation=dict(enable=True, interval=5.0, bodies=["PELVIS_S", "R_KNEE_S", "L_KNEE_S"]))      # more than two bodies
    assert spec_with(init_noise=3).init_noise_deg == 3.0
    sp = spec_with(perturbation=dict(enable=True, interval=5.0, bodies=["PELVIS_S"], force_magnitude=10, torque_magnitude=2))
    assert sp.perturb_interval == 200 and sp.perturbation_config()["force"] == 10.0
    assert spec_with(perturbation=dict(enable=False, interval=5.0, bodies=["PELVIS_S"])).perturbation_config() is None
    spec_with(dynamics_randomization=dict(enable=False, interval=0.5), init_noise=0)      # configured but off: fine
