"""Single-env surface and the run_experiment.py entry point on the GPU (mirrors reference tests/test_environments.py
and tests/test_training.py:206-235)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["cartpole", "jvrc_walk", "jvrc_step", "h1", "h1_walk"])
def test_single_env_surface(name):
    from learninghumanoidwalking_amd.envs import single_env
    env = single_env(name, seed=1)
    obs = env.reset()
    assert isinstance(obs, np.ndarray) and obs.shape == env.observation_space.shape and np.isfinite(obs).all()
    rs = np.random.default_rng(0)
    for _ in range(30):
        a = rs.uniform(-1, 1, env.action_space.shape[0]) * 0.3
        obs, r, done, info = env.step(a)
        # tests/test_environments.py:250-266 step signature, :174-188 reward = sum of components
        assert isinstance(obs, np.ndarray) and isinstance(r, float) and isinstance(done, bool) and isinstance(info, dict)
        assert obs.shape == env.observation_space.shape and np.isfinite(obs).all() and np.isfinite(r)
        assert abs(r - sum(info.values())) < 1e-6
        if done:
            env.reset()
    if name == "jvrc_walk":   # tests/test_environments.py:194-226 mirror / clock index validity
        assert len(env.robot.mirrored_obs) == 37 and len(env.robot.mirrored_acts) == 12 and env.robot.clock_inds == [29, 30]
        assert env.obs_mean.shape == (37,) and env.obs_std.shape == (37,)
    if name == "h1_walk":
        assert len(env.robot.mirrored_obs) == 43 and len(env.robot.mirrored_acts) == 10 and env.robot.clock_inds == [35, 36]
    if name == "jvrc_step":
        assert len(env.robot.mirrored_obs) == 39 and env.robot.clock_inds == [29, 30] and env.obs_mean.shape == (39,)
        assert set(info) == {"foot_frc_score", "foot_vel_score", "orient_cost", "height_error", "step_reward", "upper_body_reward"}
    with pytest.raises(AssertionError):
        env.step(np.zeros(env.action_space.shape[0] + 1))
    # tests/test_environments.py:232-246 required attributes; read-only RobotInterface subset (robot_interface.py:60-185)
    for attr in ("observation_space", "action_space", "robot", "task", "interface", "model", "data"):
        assert hasattr(env, attr), attr
    q, v = env.get_state()
    np.testing.assert_array_equal(env.interface.get_qpos(), q)
    np.testing.assert_array_equal(env.data.qvel, v)
    assert env.interface.nq() == q.size and env.interface.nv() == v.size
    if name != "cartpole":
        nu = env.action_space.shape[0]
        pos, vel, tq = (env.interface.get_act_joint_positions(), env.interface.get_act_joint_velocities(),
                        env.interface.get_act_joint_torques())
        assert pos.shape == vel.shape == tq.shape == (nu,) and np.isfinite(tq).all()
        # the getters return the fields of the last forward pass (one sim step behind the integrated state)
        np.testing.assert_allclose(pos, q[7:], atol=0.05)
        if name.startswith("jvrc") and not done:   # motor positions of get_obs (the H1 observations carry noise)
            np.testing.assert_allclose(obs[5:5 + nu], pos, atol=1e-5)
        assert len(env.interface.get_gear_ratios()) == nu and env.interface.get_robot_mass() > 10
    env.close()


def test_ppo_evaluate_is_deterministic_and_tracks_the_best_checkpoint(tmp_path):
    """PPO.evaluate (reference rl/algos/ppo.py:408-426): five deterministic batches on the persistent envs, mean episode
    return / length, save_if_best."""
    import torch
    from types import SimpleNamespace
    from learninghumanoidwalking_amd.envs import ENVIRONMENTS
    from learninghumanoidwalking_amd.ppo import PPO
    args = SimpleNamespace(gamma=0.99, lam=0.95, lr=3e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=512, epochs=2,
                           max_traj_len=40, num_procs=64, num_envs=64, max_grad_norm=0.5, mirror_coeff=0.4, eval_freq=100, recurrent=False,
                           imitate=None, imitate_coeff=0.3, learn_std=False, std_dev=0.2, no_mirror=False, infer_fp16=False, continued=None,
                           logdir=str(tmp_path), device_index=0)
    algo = PPO(ENVIRONMENTS["jvrc_walk"], args, seed=3)
    for attr in ("policy", "critic", "old_policy", "actor_optimizer", "critic_optimizer", "lr", "eps"):   # reference tests/test_algorithms.py
        assert hasattr(algo, attr), attr
    assert algo.actor_optimizer.param_groups[0]["lr"] == 3e-4 and "state" in algo.critic_optimizer.state_dict()
    batch = algo.sample_parallel_with_workers(deterministic=True)
    # deterministic: the stored actions are the policy means
    mu, _, _, _ = algo.kernels.forward(batch.states[:64], deterministic=True, want_value=False)
    assert torch.equal(mu, batch.actions[:64])
    ti = batch.traj_idx.cpu().numpy()
    assert ti[0] == 0 and ti[-1] == 64 * 40 and (np.diff(ti) > 0).all() and (np.diff(ti) <= 40).all()
    em = batch.env_major()            # traj_idx slices the env-major view into whole trajectories (one env, consecutive steps)
    dn = em.dones.reshape(-1).cpu().numpy()
    for a, b in zip(ti[:-1], ti[1:]):
        assert (dn[a:b - 1] == 0).all() and a // 40 == (b - 1) // 40
    assert torch.equal(em.states.view(64, 40, -1)[5, 7], batch.states.view(40, 64, -1)[7, 5])
    r0, l0 = algo.evaluate(0)
    files = set(os.listdir(tmp_path))
    assert {"actor_0.pt", "critic_0.pt", "actor.pt", "critic.pt"} <= files and algo.best_metric == r0 and 0 < l0 <= 40
    stamp = os.path.getmtime(os.path.join(tmp_path, "actor.pt"))
    algo.best_metric = r0 + 1e9                      # a worse evaluation must not overwrite the best checkpoint
    r1, _ = algo.evaluate(1)
    assert "actor_1.pt" in os.listdir(tmp_path) and os.path.getmtime(os.path.join(tmp_path, "actor.pt")) == stamp
    assert np.isfinite(r1)


def test_run_experiment_train_cartpole(tmp_path):
    cmd = [sys.executable, os.path.join(ROOT, "run_experiment.py"), "train", "--env", "cartpole", "--logdir", str(tmp_path),
           "--n-itr", "2", "--num-envs", "64", "--max-traj-len", "50", "--minibatch-size", "256", "--eval-freq", "100",
           "--learn-std", "--entropy-coeff", "0.01", "--std-dev", "0.15", "--seed", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "Sampling took" in out.stdout and "Optimizer took" in out.stdout and "fps=" in out.stdout   # scripts/benchmark_training.py:75-79
    run = [d for d in os.listdir(tmp_path) if d.endswith("_cartpole")]
    assert len(run) == 1
    files = os.listdir(os.path.join(tmp_path, run[0]))
    assert "actor_0.pt" in files and "critic_0.pt" in files and "experiment.pkl" in files
    # --continued from the checkpoint just written (reference tests/test_evaluation.py: train -> save -> load -> continue)
    actor = os.path.join(tmp_path, run[0], "actor_0.pt")
    cmd2 = [c for c in cmd if c not in ("--learn-std",)] + ["--continued", actor, "--n-itr", "1"]
    out2 = subprocess.run(cmd2, capture_output=True, text=True, timeout=600)
    assert out2.returncode == 0, out2.stdout[-2000:] + out2.stderr[-2000:]
    assert "Loaded (pre-trained) actor from" in out2.stdout


def test_ppo_learns_cartpole_a_little():
    """Sanity of the whole loop: mean episode return rises over 15 iterations of cartpole swing-up."""
    from types import SimpleNamespace
    from learninghumanoidwalking_amd.envs import CartpoleSpec
    from learninghumanoidwalking_amd.ppo import PPO
    args = SimpleNamespace(gamma=0.99, lam=0.95, lr=3e-4, eps=1e-5, entropy_coeff=0.01, clip=0.2, minibatch_size=4096, epochs=3,
                           max_traj_len=200, num_procs=512, num_envs=512, max_grad_norm=0.5, mirror_coeff=0.4, eval_freq=10**9,
                           recurrent=False, imitate=None, learn_std=True, std_dev=0.3, no_mirror=True, continued=None,
                           logdir="/tmp/lhw_test_learn", device_index=0)
    algo = PPO(CartpoleSpec, args, seed=1)
    b = algo.sample_parallel_with_workers()
    algo.obs_rms.update(b.states.cpu().numpy())
    algo.kernels.set_obs_norm(algo.obs_rms.mean, algo.obs_rms.std)
    rets = []
    for itr in range(15):
        algo.iterate(itr)
        rs, ls, cnt = algo._ep_stats
        rets.append(rs / max(cnt, 1))
        assert np.isfinite(list(algo.last_losses.values())).all()
    assert np.mean(rets[-3:]) > np.mean(rets[:3]) * 1.05, rets


def test_bench_two_ranks_sharing_one_gpu():
    """N>1 path of bench.py end to end (env sharding, gradient / advantage all-reduces, max-over-ranks timing) with two
    ranks on the one GPU of the test box over gloo; on the 8-GPU node the same code runs one rank per GPU over RCCL."""
    import json
    env = dict(os.environ, LHW_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    port = 29600 + (os.getpid() % 1000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--num-envs", "128", "--traj-len", "8", "--minibatch-size", "256", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert abs(d["value"] - 2 * 128 * 8 / (d["ms_per_step"] / 1e3)) / d["value"] < 1e-6   # whole-job aggregate


@pytest.mark.parametrize("world", [2, 4])
def test_bench_gpus_n_as_a_plain_command_launches_its_own_ranks(world):
    """`python bench.py --gpus N` without a launcher: bench.py re-executes itself as N ranks under torch.distributed.run
    (fan-out is the entry point's job, like the reference's PPO starting its Ray workers, rl/algos/ppo.py:215-297) and rank 0
    prints the one JSON line with the data-parallel block (per-rank env counts, gradient all-reduce time per optimiser step).
    Four ranks as well as two: what is uneven between ranks (build lock, rendezvous, the MAX over ranks) only shows above two."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["LHW_SHARE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "1", "--num-envs", "128", "--traj-len", "8",
           "--minibatch-size", "256", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["value"] > 0
    assert abs(d["value"] - world * 128 * 8 / (d["ms_per_step"] / 1e3)) / d["value"] < 1e-6   # whole-job aggregate over all ranks
    dp = d["data_parallel"]
    assert dp["n_gpus"] == world and dp["envs_per_rank"] == [128] * world
    assert dp["allreduce_calls_per_iter"] == d["optimizer_steps_per_iter"] and dp["allreduce_ms_per_step"] > 0


def test_run_experiment_gpus_2_launches_its_own_ranks(tmp_path):
    """`python run_experiment.py train --gpus 2 ...` as a plain command: two ranks (sharing the test box's GPU over gloo), envs sharded
    by global env id, gradients all-reduced, rank 0 logs and checkpoints -- the reference's entry point starts its own workers too
    (run_experiment.py:132-133 ray.init, rl/algos/ppo.py:184-193)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["LHW_SHARE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "run_experiment.py"), "train", "--env", "jvrc_walk", "--gpus", "2", "--num-envs", "32", "--max-traj-len", "8",
           "--minibatch-size", "128", "--n-itr", "2", "--eval-freq", "100", "--logdir", str(tmp_path), "--seed", "3"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert out.stdout.count("********** Iteration 1 ************") == 1          # rank 0 only
    assert "Sampling took" in out.stdout and "for 512 steps." in out.stdout      # 2 ranks x 32 envs x 8 control steps
    runs = [d for d in os.listdir(tmp_path) if d.endswith("_jvrc_walk")]
    assert len(runs) == 1 and os.path.exists(os.path.join(tmp_path, runs[0], "actor_0.pt"))


def test_ppo_imitate_wiring(tmp_path):
    """--imitate: expert checkpoint + env projector are picked up, the imitation loss is reported, and an env without a
    projector raises the reference's error (reference tests/test_imitation.py)."""
    import torch
    from types import SimpleNamespace
    from learninghumanoidwalking_amd.checkpoint import save_reference_checkpoint
    from learninghumanoidwalking_amd.envs import CartpoleSpec, JvrcWalkSpec
    from learninghumanoidwalking_amd.imitation import ImitationQuery
    from learninghumanoidwalking_amd.ppo import PPO
    from learninghumanoidwalking_amd.ppo_kernels import PpoKernels, reference_init

    # expert: a 37 -> 12 actor written in the reference's checkpoint format
    ek = PpoKernels(37, 12, hidden=256, max_rows=64)
    ek.set_tensors(reference_init(37, 12, 256, 0.2, generator_seed=5))
    ek.set_obs_norm(np.zeros(37), np.ones(37))
    expert_path = os.path.join(tmp_path, "actor.pt")
    save_reference_checkpoint(ek.get_tensors(), torch.zeros(37), torch.ones(37), False, expert_path, os.path.join(tmp_path, "critic.pt"))
    assert os.path.exists(expert_path)

    class Projector:
        def __call__(self, obs_batch):
            mask = obs_batch[:, 33] > 0.5           # standing-mode samples only
            return ImitationQuery(expert_obs=obs_batch[mask], sample_mask=mask, action_indices=torch.arange(12))

    class Spec(JvrcWalkSpec):
        def imitation_projector(self):
            return Projector()

    def args(**kw):
        d = dict(gamma=0.99, lam=0.95, lr=3e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=256, epochs=1, max_traj_len=8,
                 num_procs=64, num_envs=64, max_grad_norm=0.5, mirror_coeff=0.4, eval_freq=10**9, recurrent=False, imitate=expert_path,
                 imitate_coeff=0.3, learn_std=False, std_dev=0.223, no_mirror=False, continued=None, logdir=str(tmp_path / "log"),
                 device_index=0)
        d.update(kw)
        return SimpleNamespace(**d)

    algo = PPO(Spec, args(), seed=1)
    assert algo.imitation_projector is not None and algo.base_policy is not None
    algo.iterate(0)
    assert algo.last_losses["imitation"] > 0
    with pytest.raises(ValueError, match="imitation_projector"):
        PPO(CartpoleSpec, args(), seed=1)


def test_run_experiment_train_recurrent(tmp_path):
    cmd = [sys.executable, os.path.join(ROOT, "run_experiment.py"), "train", "--env", "cartpole", "--logdir", str(tmp_path),
           "--n-itr", "2", "--num-envs", "64", "--max-traj-len", "20", "--minibatch-size", "32", "--eval-freq", "100",
           "--recurrent", "--seed", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "Sampling took" in out.stdout and "Optimizer took" in out.stdout
    run = [d for d in os.listdir(tmp_path) if d.endswith("_cartpole")]
    raw = open(os.path.join(tmp_path, run[0], "actor_0.pt"), "rb").read()
    assert b"Gaussian_LSTM_Actor" in raw
