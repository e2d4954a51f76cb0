"""Imitation-loss contract between PPO and the environments (mirrors reference rl/algos/imitation.py:15-42 and
rl/algos/ppo.py:111-122, 360-368).

An env description that supports ``--imitate`` exposes ``imitation_projector()`` returning a callable
``projector(obs_batch) -> ImitationQuery``: which samples of a policy-observation minibatch are shown to the expert
(``sample_mask``), what the expert sees for them (``expert_obs``) and which student action dimensions are compared
with the expert's output (``action_indices``).  The expert itself is a frozen feed-forward actor checkpoint; its
forward pass runs through the same HIP MLP kernels as the student's (``FrozenActor``)."""
from __future__ import annotations

from dataclasses import dataclass

import torch


@dataclass
class ImitationQuery:
    expert_obs: torch.Tensor       # (n_active, expert_obs_dim): already filtered to the contributing samples
    sample_mask: torch.Tensor      # (B,) bool: which rows of the minibatch fed the expert
    action_indices: torch.Tensor   # (k,) long: student action dims compared with the expert's output


class FrozenActor:
    """Expert policy loaded from a reference-format actor checkpoint; ``__call__(obs) -> mean actions`` on the GPU."""

    def __init__(self, path, device, max_rows):
        from .checkpoint import load_reference_actor
        from .ppo_kernels import PpoKernels
        t, om, osd, hidden = load_reference_actor(path)
        self.obs_dim, self.act_dim = int(t["a_w1"].shape[1]), int(t["a_w3"].shape[0])
        self.k = PpoKernels(self.obs_dim, self.act_dim, hidden=hidden, max_rows=max_rows, device=device)
        self.k.set_tensors(t)
        self.k.set_obs_norm(om.numpy(), osd.numpy())

    @torch.no_grad()
    def __call__(self, obs):
        obs = obs.to(self.k.device, torch.float32).contiguous()
        if obs.shape[1] != self.obs_dim:
            raise ValueError(f"expert expects {self.obs_dim}-dim observations, projector produced {obs.shape[1]}")
        mu, _, _, _ = self.k.forward(obs, deterministic=True, want_value=False)
        return mu[: obs.shape[0]].clone()


def dense_imitation_target(query: ImitationQuery, expert_means: torch.Tensor, batch: int, act_dim: int):
    """Scatter the expert means into the dense [B, A] target / mask the loss kernel consumes; returns
    (target f32, mask u8, n_selected)."""
    dev = expert_means.device
    rows = torch.nonzero(query.sample_mask.to(dev), as_tuple=False).reshape(-1)
    cols = query.action_indices.to(dev, torch.long)
    target = torch.zeros(batch, act_dim, dtype=torch.float32, device=dev)
    mask = torch.zeros(batch, act_dim, dtype=torch.uint8, device=dev)
    if rows.numel() and cols.numel():
        target[rows[:, None], cols[None, :]] = expert_means.to(torch.float32)
        mask[rows[:, None], cols[None, :]] = 1
    return target, mask, int(rows.numel() * cols.numel())
