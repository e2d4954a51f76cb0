"""CPU restatement (torch float32 / float64) of the reference's learner math.  TEST INFRASTRUCTURE ONLY.

Pinned against the reference itself by tests/golden/gen_golden.py -> tests/golden/ppo_*.npz:
the reference's own classes are imported from /root/reference in the build container and their
outputs committed as fixtures; tests/test_oracle_ppo.py checks this file against them.

Follows: rl/policies/actor.py:160-188 (Gaussian_FF_Actor), rl/policies/critic.py:41-49 (FF_V),
rl/algos/ppo.py:299-406 (update_actor_critic, FF path, mask=1), ppo.py:484-485 (advantage
normalisation), rl/storage/rollout_storage.py:53-85 (GAE), rl/envs/wrappers.py:53-85 (mirror).
"""
from __future__ import annotations

import numpy as np
import torch


def symmetry_matrix(mirrored):
    """rl/envs/wrappers.py:78-85"""
    n = len(mirrored)
    mat = np.zeros((n, n))
    for i, j in zip(np.arange(n), np.abs(np.array(mirrored).astype(int))):
        mat[i, j] = np.sign(mirrored[i])
    return mat


def mirror_tables(mirrored, clock_inds=()):
    """Signed permutation as a gather: out[j] = sign[j] * in[src[j]], clock columns negated
    (sin(arcsin(c)+pi) == -c, wrappers.py:69-74)."""
    M = symmetry_matrix(mirrored)
    n = len(mirrored)
    src = np.zeros(n, dtype=np.int32)
    sign = np.zeros(n, dtype=np.float32)
    for j in range(n):
        nz = np.nonzero(M[:, j])[0]
        assert len(nz) == 1, "mirror indices are not a signed permutation"
        src[j] = nz[0]
        sign[j] = M[nz[0], j]
    for c in clock_inds:
        sign[c] = -sign[c]
    return src, sign


def mlp(x, W1, b1, W2, b2, W3, b3):
    h = torch.relu(x @ W1.T + b1)
    h = torch.relu(h @ W2.T + b2)
    return h @ W3.T + b3


def gae_returns(rew, val, last_val, gamma, lam):
    """One trajectory, float64 like PPOBuffer.finish_path (rollout_storage.py:53-85)."""
    rew = np.asarray(rew, dtype=np.float64)
    val = np.asarray(val, dtype=np.float64)
    nxt = np.concatenate([val[1:], [float(last_val)]])
    deltas = rew + gamma * nxt - val
    adv = np.zeros_like(rew)
    g = 0.0
    for t in range(len(rew) - 1, -1, -1):
        g = deltas[t] + gamma * lam * g
        adv[t] = g
    return adv + val


def gae_batch(rew, val, done, vterm, vfinal, gamma, lam):
    """Time-major [T][N] version with the bootstrap rules of rollout_worker.py:163-190."""
    T, N = rew.shape
    ret = np.zeros((T, N), dtype=np.float32)
    for n in range(N):
        start = 0
        for t in range(T):
            if done[t, n] or t == T - 1:
                if done[t, n]:
                    last = 0.0 if (done[t, n] & 1) else float(vterm[t, n])
                else:
                    last = float(vfinal[n])
                ret[start:t + 1, n] = gae_returns(rew[start:t + 1, n], val[start:t + 1, n], last, gamma, lam)
                start = t + 1
    return ret


class OraclePPO:
    """update_actor_critic + Adam on plain tensors (weights as dict name -> tensor, torch layouts)."""

    def __init__(self, actor, critic, stds, obs_mean, obs_std, *, lr=3e-4, eps=1e-5, clip=0.2, entropy_coeff=0.0,
                 mirror_coeff=0.4, max_grad_norm=0.5, learn_std=False, mirror_obs=None, mirror_act=None):
        f32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)
        self.actor = [f32(a).requires_grad_() for a in actor]      # W1,b1,W2,b2,W3,b3
        self.critic = [f32(a).requires_grad_() for a in critic]
        self.stds = f32(stds)
        if learn_std:
            self.stds.requires_grad_()
        self.learn_std = learn_std
        self.obs_mean, self.obs_std = f32(obs_mean), f32(obs_std)
        self.clip, self.ent, self.mir, self.gc = clip, entropy_coeff, mirror_coeff, max_grad_norm
        aparams = self.actor + ([self.stds] if learn_std else [])
        self.aopt = torch.optim.Adam(aparams, lr=lr, eps=eps)
        self.copt = torch.optim.Adam(self.critic, lr=lr, eps=eps)
        self.mirror_obs, self.mirror_act = mirror_obs, mirror_act  # (src, sign) tables or None

    def _norm(self, x):
        return (x - self.obs_mean) / self.obs_std

    def mu(self, obs):
        return mlp(self._norm(obs), *self.actor)

    def value(self, obs):
        return mlp(self._norm(obs), *self.critic)

    def log_prob(self, obs, act):
        return torch.distributions.Normal(self.mu(obs), self.stds).log_prob(act).sum(-1, keepdim=True)

    def update(self, obs, act, ret, adv, old_logp, imit=None):
        """obs [B,D], act [B,A], ret/adv/old_logp [B,1] float32 tensors.  imit = (coeff, sample_mask [B] bool,
        action_indices [k] long, expert_means [n_active, k]) adds the imitation term of ppo.py:360-368."""
        pdf = torch.distributions.Normal(self.mu(obs), self.stds)
        logp = pdf.log_prob(act).sum(-1, keepdim=True)
        ratio = (logp - old_logp).exp()
        cpi = ratio * adv
        cl = ratio.clamp(1.0 - self.clip, 1.0 + self.clip) * adv
        actor_loss = -torch.min(cpi, cl).mean()
        clip_fraction = torch.mean((torch.abs(ratio - 1) > self.clip).float()).item()
        values = self.value(obs)
        critic_loss = torch.nn.functional.mse_loss(ret, values)
        entropy_penalty = -(pdf.entropy()).mean()
        if self.mirror_obs is not None:
            src, sign = self.mirror_obs
            mobs = obs[:, torch.as_tensor(src, dtype=torch.long)] * torch.as_tensor(sign)
            mact = self.mu(mobs)
            asrc, asign = self.mirror_act
            mact = mact[:, torch.as_tensor(asrc, dtype=torch.long)] * torch.as_tensor(asign)
            mirror_loss = (pdf.mean - mact).pow(2).mean()
        else:
            mirror_loss = torch.zeros_like(actor_loss)
        with torch.no_grad():
            approx_kl = torch.mean((ratio - 1) - (logp - old_logp))
        imitation_loss = torch.zeros_like(actor_loss)
        imit_coeff = 0.0
        if imit is not None and bool(imit[1].any()):
            imit_coeff, smask, aidx, target = imit
            pred = pdf.mean[smask][:, aidx]
            imitation_loss = (pred - target).pow(2).mean()
        total = actor_loss + self.mir * mirror_loss + imit_coeff * imitation_loss + self.ent * entropy_penalty + critic_loss
        self.aopt.zero_grad()
        self.copt.zero_grad()
        total.backward()
        aparams = self.actor + ([self.stds] if self.learn_std else [])
        torch.nn.utils.clip_grad_norm_(aparams, self.gc)
        torch.nn.utils.clip_grad_norm_(self.critic, self.gc)
        self.aopt.step()
        self.copt.step()
        return (actor_loss.item(), entropy_penalty.item(), critic_loss.item(), approx_kl.item(), mirror_loss.item(),
                imitation_loss.item(), clip_fraction)
