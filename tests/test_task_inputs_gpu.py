"""GPU twin of tests/test_task_inputs.py: the task inputs exported by the HIP kernel through the C ABI
(lhw_env_enable_task_inputs / lhw_env_get_task_inputs) reproduce the kernel's fused walking reward terms when fed to
tasks/rewards.py (the reference module when present, else its pinned restatement)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_fused_walking_reward_equals_reference_rewards_on_exported_task_inputs_gpu():
    from learninghumanoidwalking_amd.envs.jvrc_walk import JvrcWalkSpec
    from tests.test_task_inputs import check_env
    spec = JvrcWalkSpec()
    env = spec.make_batched(37, seed=5, device=0)
    worst, is_ref = check_env(env, spec, 8, np.random.default_rng(2))
    print(f"fused reward terms vs rewards.py on the exported inputs (GPU): max |diff| {worst:.2e} (reference module: {is_ref})")


def test_task_inputs_need_enabling():
    from learninghumanoidwalking_amd import _lib
    from learninghumanoidwalking_amd.envs.jvrc_walk import JvrcWalkSpec
    env = JvrcWalkSpec().make_batched(2, seed=1, device=0)
    env.reset()
    with pytest.raises(_lib.LhwError):
        env.get_task_inputs()
