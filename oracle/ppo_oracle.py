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


# ----------------------------------------------------------------------------- recurrent (LSTM) branch
def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.LSTMCell arithmetic (gate order i, f, g, o)."""
    gates = x @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
    i, f, g, o = gates.chunk(4, dim=-1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    return torch.sigmoid(o) * torch.tanh(c2), c2


def lstm_net(x, reset, p):
    """x [T, B, D] normalised inputs, reset [T, B] bool (episode starts at t), p = [w_ih1, w_hh1, b_ih1, b_hh1, w_ih2, w_hh2,
    b_ih2, b_hh2, w_out, b_out] -> read-out [T, B, O].  Gaussian_LSTM_Actor._get_dist_params / LSTM_V.forward
    (reference rl/policies/actor.py:233-262, critic.py:84-112) on env columns with in-column episode starts."""
    T, B, _ = x.shape
    H = p[1].shape[1]
    h1 = c1 = h2 = c2 = torch.zeros(B, H)
    ys = []
    for t in range(T):
        keep = (~reset[t]).float().unsqueeze(-1)
        h1, c1, h2, c2 = h1 * keep, c1 * keep, h2 * keep, c2 * keep
        h1, c1 = lstm_cell(x[t], h1, c1, *p[0:4])
        h2, c2 = lstm_cell(h1, h2, c2, *p[4:8])
        ys.append(h2 @ p[8].t() + p[9])
    return torch.stack(ys)


class OracleRecurrentPPO:
    """PPO.update_actor_critic, recurrent branch (reference rl/algos/ppo.py:299-406 with mask, :512-533), stated on env
    columns: every (t, column) entry is a valid sample, episode starts inside a column reset the state."""

    def __init__(self, actor, critic, stds, obs_mean, obs_std, *, lr=3e-4, eps=1e-5, clip=0.2, mirror_coeff=0.4, max_grad_norm=0.5,
                 mirror_obs=None, mirror_act=None):
        f32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)
        self.actor = [f32(a).requires_grad_() for a in actor]
        self.critic = [f32(a).requires_grad_() for a in critic]
        self.stds = f32(stds)
        self.obs_mean, self.obs_std = f32(obs_mean), f32(obs_std)
        self.clip, self.mir, self.gc = clip, mirror_coeff, max_grad_norm
        self.aopt = torch.optim.Adam(self.actor, lr=lr, eps=eps)
        self.copt = torch.optim.Adam(self.critic, lr=lr, eps=eps)
        self.mirror_obs, self.mirror_act = mirror_obs, mirror_act

    def mu(self, obs, reset):
        return lstm_net((obs - self.obs_mean) / self.obs_std, reset, self.actor)

    def value(self, obs, reset):
        return lstm_net((obs - self.obs_mean) / self.obs_std, reset, self.critic)

    def log_prob(self, obs, reset, act):
        return torch.distributions.Normal(self.mu(obs, reset), self.stds).log_prob(act).sum(-1, keepdim=True)

    def update(self, obs, reset, act, ret, adv, old_logp):
        """obs [T,B,D], reset [T,B] bool, act [T,B,A], ret/adv/old_logp [T,B,1]."""
        pdf = torch.distributions.Normal(self.mu(obs, reset), self.stds)
        logp = pdf.log_prob(act).sum(-1, keepdim=True)
        ratio = (logp - old_logp).exp()
        actor_loss = -torch.min(ratio * adv, ratio.clamp(1.0 - self.clip, 1.0 + self.clip) * adv).mean()
        critic_loss = (ret - self.value(obs, reset)).pow(2).mean()
        if self.mirror_obs is not None:
            src, sign = self.mirror_obs
            mobs = obs[..., torch.as_tensor(src, dtype=torch.long)] * torch.as_tensor(sign)
            mact = self.mu(mobs, reset)
            asrc, asign = self.mirror_act
            mact = mact[..., torch.as_tensor(asrc, dtype=torch.long)] * torch.as_tensor(asign)
            mirror_loss = (pdf.mean - mact).pow(2).mean()
        else:
            mirror_loss = torch.zeros_like(actor_loss)
        total = actor_loss + self.mir * mirror_loss + critic_loss
        self.aopt.zero_grad()
        self.copt.zero_grad()
        total.backward()
        torch.nn.utils.clip_grad_norm_(self.actor, self.gc)
        torch.nn.utils.clip_grad_norm_(self.critic, self.gc)
        self.aopt.step()
        self.copt.step()
        return actor_loss.item(), critic_loss.item(), mirror_loss.item()


def trajectories_to_columns(lengths, T):
    """Greedy packing of back-to-back trajectories into columns of exactly T steps: returns per column the list of
    (start offset in the concatenated arrays, length); raises if the lengths do not tile."""
    cols, cur, fill, off = [], [], 0, 0
    for n in lengths:
        if fill + n > T:
            raise ValueError("trajectory lengths do not tile columns of length %d" % T)
        cur.append((off, n))
        fill += n
        off += n
        if fill == T:
            cols.append(cur)
            cur, fill = [], 0
    if cur:
        raise ValueError("last column is not full")
    return cols
