""""YAML configs and footstep plans drop in unchanged" (BASELINE.json north_star): every Spec is built from the REFERENCE's own
envs/*/configs/*.yaml and utils/footstep_plans.txt, and the resulting env is reset and stepped -- on the host-side SIMT
emulator, so this runs in the CPU suite.  Skipped where /root/reference is absent (the GPU box)."""
import os

import numpy as np
import pytest

from tests import emu

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "envs")), reason="reference checkout not present")


def _cases():
    from learninghumanoidwalking_amd.envs.h1 import H1Spec
    from learninghumanoidwalking_amd.envs.h1_walk import H1WalkSpec
    from learninghumanoidwalking_amd.envs.jvrc_step import JvrcStepSpec
    from learninghumanoidwalking_amd.envs.jvrc_walk import JvrcWalkSpec
    jv = os.path.join(REF, "envs", "jvrc", "configs", "base.yaml")
    return [("jvrc_walk", lambda: JvrcWalkSpec(yaml_path=jv)),
            ("jvrc_step", lambda: JvrcStepSpec(yaml_path=jv, plans_path=os.path.join(REF, "utils", "footstep_plans.txt"))),
            ("h1", lambda: H1Spec(yaml_path=os.path.join(REF, "envs", "h1", "configs", "base.yaml"))),
            ("h1_walk", lambda: H1WalkSpec(yaml_path=os.path.join(REF, "envs", "h1", "configs", "walk.yaml")))]


@pytest.mark.parametrize("name", ["jvrc_walk", "jvrc_step", "h1", "h1_walk"])
def test_reference_yaml_builds_and_steps(name):
    make = dict(_cases())[name]
    spec = make()
    default = type(spec)()
    # the shipped (compactly restated) configs carry the same values as the reference files
    assert spec.sim_dt == default.sim_dt and spec.control_dt == default.control_dt and spec.frame_skip == default.frame_skip
    np.testing.assert_array_equal(spec.kp, default.kp)
    np.testing.assert_array_equal(spec.kd, default.kd)
    env = emu.make_emulated(spec, 2, seed=1, max_traj_len=20)
    obs = env.reset().copy()
    assert obs.shape == (2, spec.obs_dim) and np.isfinite(obs).all()
    rs = np.random.default_rng(0)
    for _ in range(3):
        obs, rew, done, _ = env.step((rs.normal(size=(2, spec.act_dim)) * 0.1).astype(np.float32))
        assert np.isfinite(obs).all() and np.isfinite(rew).all()
    if name == "jvrc_step":
        assert len(spec.plans) > 100                       # the reference's plan file, not the shipped synthetic one
    assert env.pop_fault_stats() == (0, 0)
    env.close()
