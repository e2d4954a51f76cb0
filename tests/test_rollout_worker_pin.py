"""The rollout semantics the batched path implements (Rollout.collect + lhw_gae, checked on the GPU against
oracle/ppo_oracle.py::gae_batch) against the reference's RolloutWorker.sample + PPOBuffer, EXECUTED: tests/golden/
rollout_worker.npz holds three consecutive sample() calls of the reference worker (rl/workers/rollout_worker.py:97-199) on a
scripted env with the reference's own actor / critic -- episodes ending by termination, by truncation at max_traj_len, in the
middle of a buffer (episode carried into the next call) and exactly at a buffer end.  The oracle's flag / bootstrap rules
(terminated -> 0, truncated -> V(next state), buffer end -> V(current state)) must reproduce the reference's returns."""
import os

import numpy as np
import torch

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rollout_worker.npz"))


def _nets():
    from oracle import ppo_oracle as po
    g = lambda k: torch.tensor(G[k])
    actor = [g("w_actor_layers.0.weight"), g("w_actor_layers.0.bias"), g("w_actor_layers.1.weight"), g("w_actor_layers.1.bias"),
             g("w_means.weight"), g("w_means.bias")]
    critic = [g("w_c_critic_layers.0.weight"), g("w_c_critic_layers.0.bias"), g("w_c_critic_layers.1.weight"), g("w_c_critic_layers.1.bias"),
              g("w_c_network_out.weight"), g("w_c_network_out.bias")]
    return (lambda x: po.mlp(x, *actor)), (lambda x: po.mlp(x, *critic))


def test_gae_batch_bootstrap_rules_equal_executed_rollout_worker():
    from oracle import ppo_oracle as po
    D, A, MAXLEN, STEPS = (int(x) for x in G["cfg"])
    table, rew_table, term_at = G["table"], G["rew_table"], set(int(x) for x in G["term_at"])
    mu, V = _nets()
    k = 0               # global env step
    traj_len = 0
    carried = None
    ends_seen = 0
    for call in range(3):
        pre = f"c{call}_"
        states, actions, rewards = G[pre + "states"], G[pre + "actions"], G[pre + "rewards"][:, 0]
        values, returns, dones = G[pre + "values"][:, 0], G[pre + "returns"][:, 0], G[pre + "dones"][:, 0]
        assert states.shape[0] == STEPS
        if carried is not None:      # an episode that was running when the previous buffer filled continues, no reset
            np.testing.assert_allclose(states[0], carried, rtol=0, atol=1e-6)
        flags = np.zeros(STEPS, np.uint8)
        vterm = np.zeros(STEPS, np.float32)
        ep_lens, ep_rews, cur_len, cur_rew = [], [], traj_len, 0.0
        with torch.no_grad():
            # deterministic actions and values as the worker computed them from the stored states
            np.testing.assert_allclose(mu(torch.tensor(states, dtype=torch.float32)).numpy(), actions, rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(V(torch.tensor(states, dtype=torch.float32)).numpy()[:, 0], values, rtol=1e-5, atol=1e-6)
            for t in range(STEPS):
                nxt = table[(k * 7 + 1) % 400] + 0.01 * float(np.sum(actions[t]))
                assert abs(rewards[t] - np.float32(rew_table[k])) < 1e-6
                traj_len += 1
                terminated, truncated = k in term_at, traj_len >= MAXLEN
                flags[t] = int(terminated) | (2 if truncated else 0)
                assert bool(dones[t]) == (terminated or truncated)
                if terminated or truncated:
                    vterm[t] = float(V(torch.tensor(nxt, dtype=torch.float32)))
                    ep_lens.append(traj_len)
                    traj_len = 0
                k += 1
            ended = bool(flags[-1])
            vfinal = 0.0 if ended else float(V(torch.tensor(nxt, dtype=torch.float32)))
        ret = po.gae_batch(rewards[:, None], values[:, None], flags[:, None], vterm[:, None], np.array([vfinal], np.float32), 0.99, 0.95)
        np.testing.assert_allclose(ret[:, 0], returns, rtol=0, atol=2e-6, err_msg=f"returns of sample() call {call}")
        # the reference reports the lengths of the episodes completed in the call (an episode spanning calls counts whole)
        got = list(G[pre + "ep_lens"])
        assert len(got) == len(ep_lens) and got[1:] == ep_lens[1:] and got[0] >= ep_lens[0] - 0
        assert bool(G[pre + "carried"][0]) == (not ended)
        carried = (nxt if not ended else None)
        ends_seen += len(ep_lens)
        # traj_idx = the boundaries the batched path records too (reset points + buffer end)
        bounds = [0] + [t + 1 for t in range(STEPS) if flags[t]]
        if bounds[-1] != STEPS:
            bounds.append(STEPS)
        assert list(G[pre + "traj_idx"]) == bounds
    assert ends_seen >= 8 and any(G[f"c{c}_carried"][0] for c in range(3)) and not all(G[f"c{c}_carried"][0] for c in range(3))
