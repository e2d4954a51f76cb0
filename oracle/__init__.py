"""CPU oracle -- TEST INFRASTRUCTURE ONLY (see oracle/mjc_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""


def make_oracle_env(name, seed=0, env_id=0, max_traj_len=400):
    """(env, obs_dim, act_dim) for the CPU baseline / parity harnesses."""
    if name == "cartpole":
        from learninghumanoidwalking_amd.envs import CartpoleSpec
        from .env_cartpole import OracleCartpoleEnv
        spec = CartpoleSpec()
        return OracleCartpoleEnv(spec.model(), seed=seed, env_id=env_id, kp=spec.kp, kd=spec.kd, frame_skip=spec.frame_skip,
                                 max_traj_len=max_traj_len), 5, 1
    if name == "jvrc_walk":
        from .env_jvrc_walk import make_oracle_jvrc_walk
        return make_oracle_jvrc_walk(seed=seed, env_id=env_id, max_traj_len=max_traj_len), 37, 12
    if name == "jvrc_step":
        from .env_jvrc_step import make_oracle_jvrc_step
        return make_oracle_jvrc_step(seed=seed, env_id=env_id, max_traj_len=max_traj_len), 39, 12
    if name == "h1_walk":
        from .env_h1_walk import make_oracle_h1_walk
        return make_oracle_h1_walk(seed=seed, env_id=env_id, max_traj_len=max_traj_len), 43, 10
    if name == "h1":
        from .env_h1 import make_oracle_h1
        return make_oracle_h1(seed=seed, env_id=env_id, max_traj_len=max_traj_len), 35, 10
    raise KeyError(name)
