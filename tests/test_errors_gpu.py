"""Error behaviour of the C ABI on the GPU box: unsupported models and malformed arguments fail loudly with a message,
and edge-size batches (a single env, a batch that is not a multiple of the wave size) work."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_unsupported_models_are_rejected_with_a_message():
    from learninghumanoidwalking_amd import _lib, mjcf
    from learninghumanoidwalking_amd.batched_env import TASK_JVRC_WALK, BatchedEnv
    from learninghumanoidwalking_amd.envs import JvrcStepSpec, JvrcWalkSpec
    walk, step = JvrcWalkSpec(), JvrcStepSpec()
    kw = dict(frame_skip=walk.frame_skip, kp=walk.kp, kd=walk.kd, action_smoothing=0.5, nominal_qpos=walk.nominal_pose,
              action_offset=walk.action_offset(), task_params=[0.8], task_iparams=walk.body_ids(), clock_lut=walk.clock_lut())
    # the stepping model (foot box vs terrain box pairs) under the walking task: box-box is not compiled into those kernels
    with pytest.raises(_lib.LhwError, match="box-box"):
        BatchedEnv(step.model(), TASK_JVRC_WALK, 4, **kw)
    # wrong robot layout for the task
    from learninghumanoidwalking_amd.envs import H1Spec
    with pytest.raises(_lib.LhwError, match="12 actuated"):
        BatchedEnv(H1Spec().model(), TASK_JVRC_WALK, 4, **kw)
    # missing gait clock
    kw2 = dict(kw)
    kw2["clock_lut"] = None
    with pytest.raises(_lib.LhwError, match="clock"):
        BatchedEnv(walk.model(), TASK_JVRC_WALK, 4, **kw2)
    # a model beyond the compiled-in limits (too many actuated joints)
    chain = "".join(f"<body pos='0 0 -0.1'><joint name='j{i}' type='hinge' axis='0 1 0'/><geom type='capsule' size='.02 .04'/>" for i in range(20))
    xml = (f"<mujoco><worldbody><body name='r' pos='0 0 3'><freejoint/><geom type='sphere' size='.1'/>{chain}{'</body>' * 20}</body></worldbody>"
           "<actuator>" + "".join(f"<motor joint='j{i}'/>" for i in range(20)) + "</actuator></mujoco>")
    big = mjcf.compile_string(xml)
    with pytest.raises(_lib.LhwError, match="exceeds compiled limits"):
        BatchedEnv(big, TASK_JVRC_WALK, 4, **kw)


def test_too_long_footstep_plan_is_refused(tmp_path):
    from learninghumanoidwalking_amd.envs import JvrcStepSpec
    f = tmp_path / "plans.txt"
    f.write_text("---\n" + "\n".join(f"{0.1 * i},0.1,0.0" for i in range(25)) + "\n---\n")
    with pytest.raises(ValueError, match="at most 20"):
        JvrcStepSpec(plans_path=str(f))


def test_wrong_argument_shapes_fail_before_the_kernel():
    from learninghumanoidwalking_amd.envs import JvrcWalkSpec
    env = JvrcWalkSpec().make_batched(3, seed=0, device=0)
    env.reset()
    with pytest.raises(AssertionError):
        env.step(torch.zeros(3, 11, device="cuda"))
    with pytest.raises(AssertionError):
        env.step(torch.zeros(3, 12, device="cuda", dtype=torch.float64))
    with pytest.raises(AssertionError):
        env.reset(torch.ones(2, dtype=torch.uint8, device="cuda"))
    env.close()


@pytest.mark.parametrize("n", [1, 3, 65])
def test_edge_batch_sizes_match_the_oracle(n):
    """One env, fewer envs than lanes, and a count that is not a multiple of anything."""
    from learninghumanoidwalking_amd.envs import JvrcWalkSpec
    from oracle.env_jvrc_walk import OracleJvrcWalkEnv
    spec = JvrcWalkSpec()
    env = spec.make_batched(n, seed=4, device=0)
    obs = env.reset().cpu().numpy()
    ids = sorted({0, n - 1})
    orc = {i: OracleJvrcWalkEnv(spec, seed=4, env_id=i) for i in ids}
    for i in ids:
        np.testing.assert_allclose(obs[i], orc[i].reset(), rtol=1e-6, atol=1e-6)
    act = (np.random.default_rng(n).normal(size=(2, n, 12)) * 0.2).astype(np.float32)
    for t in range(2):
        o, r, d, _ = env.step(torch.from_numpy(act[t]).cuda())
        for i in ids:
            ro = orc[i].step(act[t, i])
            np.testing.assert_allclose(o.cpu().numpy()[i], ro[0], rtol=1e-5, atol=2e-6)
            assert abs(float(r[i]) - ro[1]) < 2e-6
    # masked reset of a single env leaves the others untouched
    q0, _ = env.get_state()
    mask = torch.zeros(n, dtype=torch.uint8, device="cuda")
    mask[n - 1] = 1
    env.reset(mask)
    q1, _ = env.get_state()
    if n > 1:
        np.testing.assert_array_equal(q0[: n - 1], q1[: n - 1])
    assert not np.array_equal(q0[n - 1], q1[n - 1])
    env.close()
