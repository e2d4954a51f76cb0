"""obs_history_len > 1 (reference envs/common/base_humanoid_env.py:177-197, 274): the product keeps the history above the
kernels (batched_env.history_update); tests/golden/refenv_history.npz holds the reference's OWN JvrcWalkEnv executed with
obs_history_len = 3 (tests/golden/gen_refenv.py history): feeding history_update the base observations of that run (the first
37 entries of every logged observation), its done flags and its post-reset observations must reproduce the reference's full
observations exactly -- shift order (newest first), zero fill after a reset, the terminal observation."""
import os

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refenv_history.npz"))
P = "jvrc_walk_h3_"


def test_history_update_reproduces_the_executed_reference_history():
    from learninghumanoidwalking_amd.batched_env import history_update
    base, H = int(G[P + "base_obs_len"]), int(G[P + "history_len"])
    assert (base, H) == (37, 3)
    obs, done, reset_obs, reset_at = G[P + "obs"], G[P + "done"], G[P + "reset_obs"], list(G[P + "reset_at"])
    assert obs.shape[1] == base * H and done.sum() >= 1
    full = torch.zeros(1, base * H, dtype=torch.float64)
    full[0, :base] = torch.from_numpy(reset_obs[0][:base])          # BatchedEnv.reset: zero fill, then the first observation
    np.testing.assert_array_equal(full[0].numpy(), reset_obs[0])
    nres = 1
    for t in range(obs.shape[0]):
        term_base = torch.from_numpy(obs[t][:base]).unsqueeze(0)    # the state the step reached
        d = torch.tensor([done[t]], dtype=torch.uint8)
        # what the kernel returns as `obs` for an env whose episode ended: the first observation after the auto-reset
        nxt = reset_obs[nres][:base] if done[t] else obs[t][:base]
        new_full, term = history_update(full, torch.from_numpy(nxt).unsqueeze(0), term_base, d, base)
        np.testing.assert_array_equal(term[0].numpy(), obs[t], err_msg=f"(terminal) observation of step {t}")
        if done[t]:
            assert reset_at[nres] == t
            np.testing.assert_array_equal(new_full[0].numpy(), reset_obs[nres], err_msg=f"observation after the reset at step {t}")
            nres += 1
        else:
            np.testing.assert_array_equal(new_full[0].numpy(), obs[t])
        full = new_full
    assert nres == len(reset_at)


def test_spec_dimensions_and_normalisation_follow_the_history_length(tmp_path):
    from learninghumanoidwalking_amd.envs.jvrc_walk import JVRC_BASE_YAML, JvrcWalkSpec
    from learninghumanoidwalking_amd.envs.jvrc_step import JvrcStepSpec
    src = open(JVRC_BASE_YAML).read()
    assert "obs_history_len: 1" in src
    y = tmp_path / "h3.yaml"
    y.write_text(src.replace("obs_history_len: 1", "obs_history_len: 3"))
    s = JvrcWalkSpec(yaml_path=str(y))
    assert (s.base_obs_dim, s.history_len, s.obs_dim) == (37, 3, 111)
    np.testing.assert_allclose(s.obs_mean, G[P + "obs_mean"], rtol=0, atol=1e-12)     # jvrc_walk.py:62-63 (np.tile), executed
    np.testing.assert_allclose(s.obs_std, G[P + "obs_std"], rtol=0, atol=1e-12)
    with pytest.raises(NotImplementedError):
        s.mirror_tables()
    s1 = JvrcWalkSpec()
    assert (s1.base_obs_dim, s1.history_len, s1.obs_dim, len(s1.obs_mean)) == (37, 1, 37, 37) and s1.mirror_tables() is not None
    st = JvrcStepSpec(yaml_path=str(y))
    assert (st.base_obs_dim, st.obs_dim, len(st.obs_mean), len(st.obs_std)) == (39, 117, 117, 117)
