"""PPO learner with on-device rollouts: the host side of the hot path.

Mirrors the reference's learner surface (reference rl/algos/ppo.py:27-641): ``PPO(env_fn, args,
seed)``, ``sample_parallel_with_workers()``, ``update_actor_critic(...)``, ``train(env_fn, n_itr)``
with the same stdout lines and checkpoint names -- but the actor-parallel Ray rollout
(rl/workers/rollout_worker.py) is replaced by one batched environment per GPU, and every
arithmetic step (policy/critic forward, env step, GAE, losses, backward, clip, Adam) runs in
liblhw.so.  Only gradients cross GPUs (``torch.distributed`` all-reduce, RCCL over xGMI).
"""
from __future__ import annotations

import datetime
import os
import sys
import time
from dataclasses import dataclass
from pathlib import Path

import numpy as np
import torch

from . import _lib, dist_utils
from .ppo_kernels import PpoKernels, reference_init


@dataclass
class BatchData:
    """Same fields as the reference's BatchData (rl/storage/rollout_storage.py:6-22); device tensors."""
    states: torch.Tensor
    actions: torch.Tensor
    rewards: torch.Tensor
    values: torch.Tensor
    returns: torch.Tensor
    dones: torch.Tensor
    traj_idx: torch.Tensor
    ep_lens: torch.Tensor
    ep_rewards: torch.Tensor
    n_envs: int = 0          # > 0: the sample tensors are TIME-major ([T][N] flattened) and traj_idx indexes env_major()

    def env_major(self) -> "BatchData":
        """The same batch with the samples in ENV-major order (sample (env n, step t) at n * T + t): one env's steps are contiguous,
        so `states[traj_idx[i]:traj_idx[i + 1]]` is trajectory i, as with the reference's concatenated worker buffers."""
        if not self.n_envs:
            return self
        N = self.n_envs
        tr = lambda x: x.reshape(-1, N, *x.shape[1:]).transpose(0, 1).reshape(x.shape)
        return BatchData(states=tr(self.states), actions=tr(self.actions), rewards=tr(self.rewards), values=tr(self.values),
                         returns=tr(self.returns), dones=tr(self.dones), traj_idx=self.traj_idx, ep_lens=self.ep_lens,
                         ep_rewards=self.ep_rewards, n_envs=0)


class _AdamView:
    """`actor_optimizer` / `critic_optimizer` of the reference (torch.optim.Adam over the two networks, ppo.py:128-129) as a
    read-only view: one fused Adam kernel updates the flat parameter vector, its moments live in PpoKernels.adam_m / adam_v."""

    def __init__(self, kernels, names, lr, eps):
        self._k, self._names = kernels, names
        self.defaults = dict(lr=lr, betas=(0.9, 0.999), eps=eps, weight_decay=0)
        self.param_groups = [dict(self.defaults, params=list(names))]

    def state_dict(self):
        k = self._k
        key = getattr(k, "_adam_view_specs", None)      # (recurrent kernels: _view takes the tensor's spec)
        view = (lambda flat, n: k._view(flat, key[n])) if key else k._view
        state = {n: dict(step=int(k.adam_step), exp_avg=view(k.adam_m, n).detach().cpu().clone(),
                         exp_avg_sq=view(k.adam_v, n).detach().cpu().clone()) for n in self._names}
        return dict(state=state, param_groups=[dict(self.defaults, params=list(self._names))])

    def zero_grad(self):      # gradients are zeroed by the apply kernel after every optimiser step
        pass


class RunningMeanStd:
    """Chan parallel mean/variance (reference rl/envs/normalize.py:4-61)."""

    def __init__(self, epsilon=1e-4, shape=()):
        self.mean = np.zeros(shape, dtype=np.float64)
        self.var = np.ones(shape, dtype=np.float64)
        self.count = epsilon

    def update_from_moments(self, batch_mean, batch_var, batch_count):
        delta = batch_mean - self.mean
        tot = self.count + batch_count
        self.mean = self.mean + delta * batch_count / tot
        M2 = self.var * self.count + batch_var * batch_count + np.square(delta) * self.count * batch_count / tot
        self.var = M2 / tot
        self.count = tot

    def update(self, x):
        x = np.asarray(x)
        self.update_from_moments(x.mean(axis=0), x.var(axis=0), x.shape[0])

    @property
    def std(self):
        return np.sqrt(self.var + 1e-8)


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


class Rollout:
    """Time-major rollout storage + collection loop for one GPU (N envs x T steps per iteration).

    Semantics of RolloutWorker.sample (reference rl/workers/rollout_worker.py:97-199) for every env:
    exactly T transitions per call, episodes carried over between calls, truncation at max_traj_len,
    bootstrap (not done) * V(s') at episode ends and V(s_T) where the buffer fills mid-episode.
    """

    def __init__(self, env, kernels: PpoKernels, T: int, seed: int = 0, task=None, max_traj_len: int | None = None):
        self.env, self.k, self.T, self.seed = env, kernels, int(T), int(seed)
        # task: a task_hook.VectorTask evaluated OUTSIDE the kernels after every control step; its reward / termination replace
        # the fused ones and the rollout, not the env, truncates at max_traj_len and resets (see _collect_hooked)
        self.task, self.max_traj_len = task, int(max_traj_len if max_traj_len is not None else T)
        if task is not None and getattr(env, "history_len", 1) > 1:
            # (the hooked paths write the reset observation / history rows themselves and know nothing of the env-side history deque)
            raise NotImplementedError("a plugged-in task with obs_history_len > 1 is not supported")
        self.reward_only = bool(task is not None and getattr(task, "reward_only", False))
        self._tin_all = None
        N, D, A, dev = env.n_envs, env.obs_dim, env.act_dim, env.device
        self.N = N
        self.obs = torch.zeros(T + 1, N, D, dtype=torch.float32, device=dev)
        self.act = torch.zeros(T, N, A, dtype=torch.float32, device=dev)
        self.mu = torch.zeros(N, A, dtype=torch.float32, device=dev)
        self.logp = torch.zeros(T, N, dtype=torch.float32, device=dev)
        self.val = torch.zeros(T, N, dtype=torch.float32, device=dev)
        self.rew = torch.zeros(T, N, dtype=torch.float32, device=dev)
        self.done = torch.zeros(T, N, dtype=torch.uint8, device=dev)
        self.vterm = torch.zeros(T, N, dtype=torch.float32, device=dev)
        self.vfinal = torch.zeros(N, dtype=torch.float32, device=dev)
        self.tob = torch.zeros(N, D, dtype=torch.float32, device=dev)
        self.tob_all = None      # [T][N][D] terminal observations of the feed-forward path (allocated on first use)
        # Number of independent env groups pipelined on separate streams (wave-per-env steppers; LHW_ROLLOUT_GROUPS overrides).
        # Two groups: one group's policy launch and the tail of its control-step kernel -- a launch ends with its slowest wave,
        # the chip draining meanwhile -- hide behind the other group's kernel.  Measured on the round-4 kernels
        # (profiles/r04_rollout_groups.txt): jvrc_walk @ 4096 +11-12 % env-steps/s over one group
        # (rollout 0.608 -> 0.538 s), h1 @ 4096 +11 %, jvrc_walk @ 2048 / 1024 +4 / +3 %; three or four groups lose badly
        # (jvrc_walk @ 4096: 1.07 s).  (An earlier round had measured one group ahead at 4096 envs, 2.07 vs 1.99 M, on a
        # slower control step and an unfused policy step.)  Small batches keep one group: their launches are latency-bound.
        auto = "1"
        if hasattr(env, "step_range") and env.task != 0:
            auto = "2" if N >= 1024 else "1"
        want = int(os.environ.get("LHW_ROLLOUT_GROUPS", auto))
        self.groups = max(1, min(want, N)) if env.task != 0 else 1
        self.streams = None
        self.counter = 0
        self.started = False
        self.env_base = getattr(env, "env_id_base", 0)

    def collect(self, deterministic=False):
        env, k, T = self.env, self.k, self.T
        if getattr(k, "recurrent", False):
            return self._collect_recurrent(deterministic)
        if not self.started:
            self.obs[0].copy_(env.reset())
            self.started = True
        else:
            self.obs[0].copy_(self.obs[T])
        # Only the actor sits on the step-to-step dependency chain.  The critic's weights do not change during a rollout
        # and it is feed-forward, so V(s_t), V(terminal obs) and V(s_T) are evaluated afterwards in large batches instead
        # of two extra 3-GEMM passes per control step (same values, ~2000 fewer kernel launches per iteration).
        if self.tob_all is None:
            self.tob_all = torch.zeros(T, self.N, self.obs.shape[2], dtype=torch.float32, device=self.obs.device)
        k.begin_rollout()      # theta is frozen for the whole rollout (policy steps and the batched critic passes behind them)
        try:
            if not getattr(self, "_rare_path_warm", False):
                self._warm_rare_path()
            self._collect_steps(deterministic)
        finally:
            k.end_rollout()

    def _warm_rare_path(self):
        """Runs the truncated-trajectory branch of _collect_steps once on eight dummy rows.  An untrained policy falls long before
        max_traj_len, so that branch is first taken some ten iterations into a run -- and the first call of its torch ops
        (index_select, index_copy_) loads their code objects: 0.1-0.2 s in the middle of that iteration (bench.py's iter_s showed
        it at iteration 10 of every run).  A process start-up cost, paid here before the first rollout instead."""
        self._rare_path_warm = True
        if self.obs.device.type != "cuda" or self.tob_all is None:
            return
        T = self.T
        need = torch.zeros(T * self.N, dtype=torch.bool, device=self.obs.device)
        need[:8] = True
        idx = torch.nonzero(need).reshape(-1)
        vals = _lib.empty(idx.numel(), dtype=torch.float32, device=self.obs.device)
        self._batched_values(self.tob_all.reshape(T * self.N, -1).index_select(0, idx), vals)
        self.vterm.reshape(-1).index_copy_(0, idx, vals)
        self.vterm.zero_()

    def _collect_hooked(self, deterministic):
        """Launch-per-step rollout with the TASK outside the kernels (task_hook.py): after each control step the env's exported
        task inputs go to `self.task.evaluate`, whose reward and termination are stored; truncation at max_traj_len, the episode
        statistics and the resets (`lhw_env_reset(mask)`: the reset code and random draws of the in-kernel auto-reset) are done
        here -- RobotBase.step + the episode handling of RolloutWorker.sample (robots/robot_base.py:88-96,
        rl/workers/rollout_worker.py:150-181) with an exchangeable task."""
        from .task_hook import device_task_inputs
        env, k, T, dev = self.env, self.k, self.T, self.obs.device
        ti = device_task_inputs(env)
        if not hasattr(self, "_traj_len"):
            self._traj_len = torch.zeros(self.N, dtype=torch.int32, device=dev)
            self._ep_ret = torch.zeros(self.N, dtype=torch.float64, device=dev)
            self._stats = torch.zeros(3, dtype=torch.float64, device=dev)      # sum of returns, sum of lengths, episodes
            self._scratch_rew = torch.zeros(self.N, dtype=torch.float32, device=dev)
            self._scratch_done = torch.zeros(self.N, dtype=torch.uint8, device=dev)
        for t in range(T):
            k.forward(self.obs[t], seed=self.seed, env_id_base=self.env_base, counter=self.counter, deterministic=deterministic,
                      want_value=False, want_mu=False, act=self.act[t], logp=self.logp[t])
            env.step(self.act[t], obs_out=self.obs[t + 1], term_obs_out=self.tob_all[t], rew_out=self._scratch_rew, done_out=self._scratch_done)
            rew, term = self.task.evaluate(ti)
            # a diverged env (the kernel has sanitised its state, the record is exported before that): the episode ends -- whatever
            # the plugged task's reward looks at
            bad = ~torch.isfinite(rew) | ~torch.isfinite(ti.qpos).all(1) | ~torch.isfinite(ti.qvel).all(1) | ~torch.isfinite(ti.qacc).all(1)
            rew = torch.where(bad, torch.zeros_like(rew), rew)
            term = term.bool() | bad
            self.rew[t].copy_(rew)
            self._traj_len += 1
            self._ep_ret += rew.double()
            trunc = self._traj_len >= self.max_traj_len
            self.done[t].copy_(term.to(torch.uint8) | (trunc.to(torch.uint8) << 1))
            ended = term | trunc
            # (no host round trip per control step: the statistics are masked sums and the reset launch takes the mask as it is --
            # an all-zero mask is a launch whose waves return at once)
            self._stats += torch.stack([(self._ep_ret * ended).sum(), (self._traj_len * ended).sum().double(), ended.sum().double()])
            self.task.reset(ended)
            env.reset(ended.to(torch.uint8), obs_out=self.obs[t + 1])      # rows of the other envs are left untouched
            self._traj_len.masked_fill_(ended, 0)
            self._ep_ret.masked_fill_(ended, 0)
            self.counter += 1
        self.last_mode = "hooked"

    # ---- reward-only plug-ins: the kernel keeps its own termination / truncation / resets, the task supplies the reward
    def _hook_state(self):
        dev = self.obs.device
        if not hasattr(self, "_traj_len"):
            self._traj_len = torch.zeros(self.N, dtype=torch.int32, device=dev)
            self._ep_ret = torch.zeros(self.N, dtype=torch.float64, device=dev)
            self._stats = torch.zeros(3, dtype=torch.float64, device=dev)      # sum of returns, sum of lengths, episodes

    def _episode_stats_from_buffers(self):
        """Finished-episode returns / lengths of this rollout from self.rew (the plugged task's rewards) and self.done (the kernel's
        flags), vectorised over [T, N]; episodes carried in from / out to the neighbouring rollouts through _ep_ret / _traj_len."""
        self._hook_state()
        T, N, dev = self.T, self.N, self.obs.device
        ended = self.done != 0
        cc = torch.cumsum(self.rew.double(), dim=0)
        tt = torch.arange(T, device=dev, dtype=torch.int64).unsqueeze(1).expand(T, N)
        last = torch.cummax(torch.where(ended, tt, torch.full_like(tt, -1)), dim=0).values          # last end at or before t
        prev = torch.cat([torch.full((1, N), -1, dtype=torch.int64, device=dev), last[:-1]], dim=0)  # last end strictly before t
        has_prev = prev >= 0
        base = torch.where(has_prev, cc.gather(0, prev.clamp(min=0)), torch.zeros_like(cc))
        ret = cc - base + torch.where(has_prev, torch.zeros_like(cc), self._ep_ret.unsqueeze(0).expand(T, N))
        length = torch.where(has_prev, tt - prev, tt + 1 + self._traj_len.unsqueeze(0).to(torch.int64))
        e = ended.double()
        self._stats += torch.stack([(ret * e).sum(), (length.double() * e).sum(), e.sum()])
        fin = last[-1]                                                                                # last end of the column (-1: none)
        none = fin < 0
        tail = cc[-1] - torch.where(none, torch.zeros_like(cc[-1]), cc.gather(0, fin.clamp(min=0).unsqueeze(0)).squeeze(0))
        self._ep_ret = torch.where(none, self._ep_ret + cc[-1], tail)
        self._traj_len = torch.where(none, self._traj_len.to(torch.int64) + T, (T - 1) - fin).to(torch.int32)

    def _evaluate_reward_batch(self, rec):
        """The plugged task's reward for [M, TASK_INPUT_DIM] records (non-finite -> 0: the kernel ended that episode itself)."""
        from .task_hook import TaskInputs
        env = self.env
        rew, _ = self.task.evaluate(TaskInputs(rec, env.nq, env.nv, env.act_dim))
        return torch.where(torch.isfinite(rew), rew, torch.zeros_like(rew)).float()

    def _collect_resident_hooked(self, deterministic) -> bool:
        """Reward-only task plug-ins at the resident rollout's speed: one lhw_env_rollout_task_inputs launch (fused termination,
        truncation and resets; the sim-facade record of every control step exported, [T][N][160] float64), then ONE evaluation of
        the task over the whole batch -- RobotBase.step's `task.calc_reward` (robots/robot_base.py:88-96) moved behind the rollout,
        which it may be because a reward does not feed back into the simulation."""
        env, k, T = self.env, self.k, self.T
        mode = os.environ.get("LHW_ROLLOUT_MODE", "auto")
        if mode not in ("auto", "resident") or not hasattr(env, "rollout") or not hasattr(k, "rollout_policy"):
            return False
        if not hasattr(env._L, "lhw_env_rollout_task_inputs") or getattr(env, "env_id_base", 0) != self.env_base:
            return False
        pol = k.rollout_policy(seed=self.seed, counter=self.counter, deterministic=deterministic)
        if pol is None:
            return False
        self._pol_keep = pol
        if self._tin_all is None:
            self._tin_all = _lib.empty(T, self.N, _lib.TASK_INPUT_DIM, dtype=torch.float64, device=self.obs.device)
        if not env.rollout(pol, T, self.obs, self.act, self.logp, self.tob_all, self.rew, self.done, task_inputs=self._tin_all):
            return False
        self.counter += T
        rec = self._tin_all.reshape(T * self.N, -1)
        rew = self.rew.reshape(-1)
        chunk = max(self.N, (1 << 19) // self.N * self.N)      # whole time slices, about half a million rows per evaluation
        for a in range(0, rec.shape[0], chunk):
            rew[a:a + chunk] = self._evaluate_reward_batch(rec[a:a + chunk])
        self._episode_stats_from_buffers()
        return True

    def _collect_reward_only_steps(self, deterministic):
        """The same plug-in on the launch-per-step pipeline (LHW_ROLLOUT_MODE=steps, or no resident kernel for this env / policy):
        the kernel's own flags and resets, the task's reward per control step, no host round trip."""
        from .task_hook import device_task_inputs
        env, k, T = self.env, self.k, self.T
        ti = device_task_inputs(env)
        if not hasattr(self, "_scratch_rew"):
            self._scratch_rew = torch.zeros(self.N, dtype=torch.float32, device=self.obs.device)
        for t in range(T):
            k.forward(self.obs[t], seed=self.seed, env_id_base=self.env_base, counter=self.counter, deterministic=deterministic,
                      want_value=False, want_mu=False, act=self.act[t], logp=self.logp[t])
            env.step(self.act[t], obs_out=self.obs[t + 1], term_obs_out=self.tob_all[t], rew_out=self._scratch_rew, done_out=self.done[t])
            self.rew[t].copy_(self._evaluate_reward_batch(ti.rec))
            self.counter += 1
        self._episode_stats_from_buffers()

    def pop_episode_stats(self):
        """(sum of finished-episode returns, sum of their lengths, their number) since the last call -- the env's own counters, or,
        with a task plugged in, the rollout's (returns in the plugged task's reward)."""
        if self.task is None:
            return self.env.pop_episode_stats()
        if not hasattr(self, "_stats"):
            return 0.0, 0.0, 0
        s = self._stats.cpu().numpy()
        self._stats.zero_()
        return float(s[0]), float(s[1]), int(s[2])

    def _collect_resident(self, deterministic) -> bool:
        """All T control steps in one launch per rollout group, the actor evaluated inside the stepper's wavefronts
        (BatchedEnv.rollout): bitwise the values of the launch-per-step loop below (float32 inference; with fp16 inference float32-rounding-close),
        without its per-control-step barrier across
        the envs of a group.  LHW_ROLLOUT_MODE = auto (default) | resident | steps (the launch-per-step pipeline)."""
        env, k, T = self.env, self.k, self.T
        mode = os.environ.get("LHW_ROLLOUT_MODE", "auto")
        if mode not in ("auto", "resident") or not hasattr(env, "rollout") or not hasattr(k, "rollout_policy"):
            return False
        # Measured, same box, interleaved (profiles/r05_rollout_modes.txt, final kernels): resident over launch-per-step with two groups --
        # jvrc_walk @ 4096 +25 % env-steps/s (rollout 0.524 -> 0.397 s), jvrc_step @ 4096 +26 %, h1_walk @ 8192 +9 %, h1 @ 8192 +2 %
        # (twice as many wavefronts as the chip holds and a narrow spread of wave times: there two whole-chip launches in flight
        # already backfill each other's tails).  So `auto` is resident wherever the library has the kernel.
        pol = k.rollout_policy(seed=self.seed, counter=self.counter, deterministic=deterministic)
        if pol is None:
            return False
        self._pol_keep = pol      # (passed by value at the launch; kept for the debugger's sake)
        # (the in-wave policy step keys its noise by the env's own global ids; the per-step calls below pass self.env_base + row)
        if getattr(env, "env_id_base", 0) != self.env_base:
            return False
        if not env.rollout(pol, T, self.obs, self.act, self.logp, self.tob_all, self.rew, self.done):
            if mode == "resident":
                raise _lib.LhwError(-4, "LHW_ROLLOUT_MODE=resident, but the library has no resident rollout kernel for this env / policy")
            return False
        self.counter += T
        return True

    def _collect_steps(self, deterministic):
        env, k, T = self.env, self.k, self.T
        G = self.groups
        self.last_mode = "steps"      # which path collected the last rollout (tests, bench line)
        if self.task is not None and self.reward_only:
            if self._collect_resident_hooked(deterministic):
                self.last_mode = "resident"
            else:
                self._collect_reward_only_steps(deterministic)
                self.last_mode = "hooked"
        elif self.task is not None:
            self._collect_hooked(deterministic)
        elif self._collect_resident(deterministic):
            self.last_mode = "resident"
        elif G <= 1:
            for t in range(T):
                k.forward(self.obs[t], seed=self.seed, env_id_base=self.env_base, counter=self.counter,
                          deterministic=deterministic, want_value=False, want_mu=False, act=self.act[t], logp=self.logp[t])
                env.step(self.act[t], obs_out=self.obs[t + 1], term_obs_out=self.tob_all[t], rew_out=self.rew[t], done_out=self.done[t])
                self.counter += 1
        else:
            # Environments are independent, so the batch is advanced as G groups on their own HIP streams: a group's policy
            # forward waits only for that group's control-step kernel.  Without the batch-wide barrier per control step the
            # tail of one group's kernel (waves that drew more contacts / Newton iterations / a reset) overlaps the other
            # group's work: 3.8 -> 2.9 ms per control step of 4096 envs with two groups.  Every value is unchanged (the RNG is
            # keyed by global env id and step counter, not by launch order).
            main = torch.cuda.current_stream(self.obs.device)
            if self.streams is None:
                self.streams = [torch.cuda.Stream(device=self.obs.device) for _ in range(G)]
            bounds = [(g * self.N // G, (g + 1) * self.N // G) for g in range(G)]
            for s in self.streams:
                s.wait_stream(main)
            for t in range(T):
                for s, (a, b) in zip(self.streams, bounds):
                    with torch.cuda.stream(s):
                        k.forward(self.obs[t, a:b], seed=self.seed, env_id_base=self.env_base + a, counter=self.counter,
                                  deterministic=deterministic, want_value=False, want_mu=False, ws_row=a, act=self.act[t, a:b],
                                  logp=self.logp[t, a:b])
                        env.step_range(a, b - a, self.act[t], self.obs[t + 1], self.tob_all[t], self.rew[t], self.done[t])
                self.counter += 1
            for s in self.streams:
                main.wait_stream(s)
        self._batched_values(self.obs[:T].reshape(T * self.N, -1), self.val.reshape(-1))
        # V(terminal observation) is only read where a trajectory was TRUNCATED (max_traj_len) without terminating: the bootstrap
        # of rl/workers/rollout_worker.py:163-190 (lhw_gae: `(f & 1) ? 0 : vterm`).  That is about one row per env and rollout --
        # 0.25 % of the T x N terminal observations at T = max_traj_len = 400 -- so only those rows go through the critic
        # (50 fewer 32768-row passes per iteration at 4096 envs); everywhere else vterm is 0 and unread.
        need = ((self.done & 2) != 0) & ((self.done & 1) == 0)
        idx = torch.nonzero(need.reshape(-1)).reshape(-1)
        self.vterm.zero_()
        if idx.numel():
            vals = _lib.empty(idx.numel(), dtype=torch.float32, device=self.obs.device)
            self._batched_values(self.tob_all.reshape(T * self.N, -1).index_select(0, idx), vals)
            self.vterm.reshape(-1).index_copy_(0, idx, vals)
        k.forward(self.obs[T], want_actor=False, value=self.vfinal)

    def _batched_values(self, obs_flat, out_flat):
        chunk = int(self.k.max_rows)
        for a in range(0, obs_flat.shape[0], chunk):
            b = min(a + chunk, obs_flat.shape[0])
            self.k.forward(obs_flat[a:b], want_actor=False, value=out_flat[a:b])


    def _collect_recurrent(self, deterministic):
        """LSTM policies.  As in the reference's worker (rollout_worker.py:130-190: `current_state` / hidden state are only
        initialised when they are None), episodes AND the LSTM hidden / cell state are carried from one batch to the next:
        the envs are reset once, before the first batch; afterwards a batch starts from the last observation of the previous
        one with the hidden state the previous batch left, zeroed only for envs whose episode ended on its last step.  The
        hidden state advances with every policy / critic call; terminal and final values are evaluated without advancing it.
        (The update, like the reference's, restarts every stored trajectory -- here every env column -- from a zero state.)"""
        env, k, T = self.env, self.k, self.T
        if not self.started:
            self.obs[0].copy_(env.reset())
            self._rec_reset = torch.ones(self.N, dtype=torch.uint8, device=self.obs.device)
            self.started = True
        else:
            self.obs[0].copy_(self.obs[T])
        reset = self._rec_reset
        for t in range(T):
            k.forward(self.obs[t], reset=reset, seed=self.seed, env_id_base=self.env_base, counter=self.counter,
                      deterministic=deterministic, commit=True, mu=self.mu, act=self.act[t], logp=self.logp[t], value=self.val[t])
            env.step(self.act[t], obs_out=self.obs[t + 1], term_obs_out=self.tob, rew_out=self.rew[t], done_out=self.done[t])
            k.forward(self.tob, commit=False, want_actor=False, value=self.vterm[t])
            reset = (self.done[t] != 0).to(torch.uint8)
            self.counter += 1
        self._rec_reset = reset
        k.forward(self.obs[T], commit=False, want_actor=False, value=self.vfinal)


class PPO:
    def __init__(self, env_fn, args, seed=None, task=None):
        """`task`: None -- the task fused into the env's kernels (the reference's own) -- or a callable `task(spec, device)`
        returning a task_hook.VectorTask: reward and termination then come from that object, evaluated outside the kernels on the
        exported task inputs after every control step (feed-forward policies, humanoid envs; see task_hook.py)."""
        self.seed = seed
        self.gamma, self.lam, self.lr, self.eps = args.gamma, args.lam, args.lr, args.eps
        self.ent_coeff, self.clip = args.entropy_coeff, args.clip
        self.minibatch_size, self.epochs = args.minibatch_size, args.epochs
        self.max_traj_len = args.max_traj_len
        # --num-procs keeps its meaning "number of parallel environments" (per GPU) unless --num-envs is given
        self.n_proc = int(getattr(args, "num_envs", None) or args.num_procs)
        self.grad_clip, self.mirror_coeff = args.max_grad_norm, args.mirror_coeff
        self.eval_freq = args.eval_freq
        self.recurrent = bool(getattr(args, "recurrent", False))
        if self.recurrent and getattr(args, "imitate", None):
            raise NotImplementedError("--imitate together with --recurrent is not supported")
        self.imitate_coeff = float(getattr(args, "imitate_coeff", 0.3))
        self.batch_size = self.n_proc * self.max_traj_len
        self.total_steps = 0
        self.iteration_count = 0
        self.save_path = Path(args.logdir)
        self.best_metric = -np.inf
        d = _dist()
        self.rank = d.get_rank() if d else 0
        self.world = d.get_world_size() if d else 1
        if self.rank == 0:
            self.save_path.mkdir(parents=True, exist_ok=True)
        dev_index = getattr(args, "device_index", None)
        if dev_index is None:
            dev_index = torch.cuda.current_device() if torch.cuda.is_available() else 0
        self.device = torch.device("cuda", dev_index)

        spec = env_fn()  # env description object (see envs/*): dims, obs norm, mirror tables, batched factory
        self.spec = spec
        obs_dim, act_dim = spec.obs_dim, spec.act_dim
        mirror = None if getattr(args, "no_mirror", False) else spec.mirror_tables()
        continued = getattr(args, "continued", None)
        self.obs_rms = None
        if self.recurrent:
            # LSTM actor / critic (ppo.py:84-86); a minibatch is `minibatch_size` whole env columns of the rollout
            from .rnn_kernels import RnnKernels, reference_init_lstm
            hidden = int(getattr(args, "lstm_hidden", 256))
            cols = min(self.n_proc, int(self.minibatch_size or self.n_proc))
            self.kernels = RnnKernels(
                obs_dim, act_dim, hidden=hidden, seq_len=self.max_traj_len, seq_cols=cols, rollout_rows=self.n_proc, device=self.device,
                learn_std=args.learn_std, lr=self.lr, eps=self.eps, clip=self.clip, entropy_coeff=self.ent_coeff,
                mirror_coeff=self.mirror_coeff, max_grad_norm=self.grad_clip,
                mirror_obs=mirror[0] if mirror else None, mirror_act=mirror[1] if mirror else None)
            self.kernels.recurrent = True
            if continued:
                from .checkpoint import load_recurrent_checkpoint
                cpath = Path(Path(continued).parent, "critic" + str(continued).split("actor")[1])
                t, om, osd, _ = load_recurrent_checkpoint(continued, cpath)
                t["stds"] = args.std_dev * torch.ones(act_dim)
                self.kernels.set_tensors(t)
                self.kernels.set_obs_norm(om.numpy(), osd.numpy())
                self._log("Loaded (pre-trained) actor from: " + str(continued))
                self._log("Loaded (pre-trained) critic from: " + str(cpath))
            else:
                self.kernels.set_tensors(reference_init_lstm(obs_dim, act_dim, hidden, args.std_dev, generator_seed=self._init_seed(seed)))
        else:
            self.kernels = PpoKernels(
                obs_dim, act_dim, hidden=256, max_rows=max(self.n_proc, int(self.minibatch_size or self.batch_size)),
                device=self.device, learn_std=args.learn_std, lr=self.lr, eps=self.eps, clip=self.clip,
                entropy_coeff=self.ent_coeff, mirror_coeff=self.mirror_coeff, max_grad_norm=self.grad_clip,
                mirror_obs=mirror[0] if mirror else None, mirror_act=mirror[1] if mirror else None)
        if getattr(args, "infer_fp16", False):
            if self.recurrent:
                raise NotImplementedError("--infer-fp16 is implemented for the feed-forward policies")
            self.kernels.set_inference_fp16(True)   # rollout inference on the fp16 MFMA; the update stays float32
        if getattr(args, "fp16", False):
            # BASELINE config 5: fp16 actor / critic -- inference AND every GEMM of the update with fp16 operands (float32
            # accumulation, master weights, loss and Adam)
            if self.recurrent:
                raise NotImplementedError("--fp16 is implemented for the feed-forward policies")
            self.kernels.set_inference_fp16(True)
            self.kernels.set_update_fp16(True)
        if self.recurrent:
            pass
        elif continued:
            # --continued actor_X.pt: load actor + sibling critic, re-initialise stds, keep the embedded obs
            # normalisation (reference rl/algos/ppo.py:69-82)
            from .checkpoint import load_reference_checkpoint
            cpath = Path(Path(continued).parent, "critic" + str(continued).split("actor")[1])
            t, om, osd = load_reference_checkpoint(continued, cpath)
            t["stds"] = args.std_dev * torch.ones(act_dim)
            self.kernels.set_tensors(t)
            self.kernels.set_obs_norm(om.numpy(), osd.numpy())
            self.obs_rms = None
            self._log("Loaded (pre-trained) actor from: " + str(continued))
            self._log("Loaded (pre-trained) critic from: " + str(cpath))
        else:
            # identical initial weights on every rank: the reference's init path under a fixed torch seed
            self.kernels.set_tensors(reference_init(obs_dim, act_dim, 256, args.std_dev, generator_seed=self._init_seed(seed)))
        if continued:
            pass
        elif spec.obs_mean is not None:
            if len(spec.obs_mean) != obs_dim or len(spec.obs_std) != obs_dim:
                raise ValueError(f"{type(spec).__name__}: obs_mean / obs_std have {len(spec.obs_mean)} / {len(spec.obs_std)} entries "
                                 f"for an observation of {obs_dim} (base {getattr(spec, 'base_obs_dim', obs_dim)} x history)")
            self.obs_rms = None
            self.kernels.set_obs_norm(spec.obs_mean, spec.obs_std)
            self._log("Using fixed observation normalization from environment.")
        else:
            self.obs_rms = RunningMeanStd(shape=(obs_dim,))
            self.kernels.set_obs_norm(self.obs_rms.mean, self.obs_rms.std)
            self._log("Using running observation normalization (will update during training).")
        env_seed = (seed if seed is not None else int(time.time())) & 0x7FFFFFFF
        self.env_seed = env_seed
        if task is not None and self.recurrent:
            raise NotImplementedError("a plugged-in task needs the feed-forward policies")
        self.task = task(spec, self.device) if task is not None else None
        # A task that decides terminations itself is consulted after every control step and the env never ends an episode by itself:
        # the rollout truncates and resets (Rollout._collect_hooked).  A reward-only task leaves all that to the kernel.
        own_done = self.task is not None and not getattr(self.task, "reward_only", False)
        self.env = spec.make_batched(self.n_proc, seed=env_seed, device=self.device, max_traj_len=0 if own_done else self.max_traj_len,
                                     env_id_base=dist_utils.shard_env_ids(self.n_proc, self.rank))
        self.env.env_id_base = dist_utils.shard_env_ids(self.n_proc, self.rank)
        self.rollout = Rollout(self.env, self.kernels, self.max_traj_len, seed=env_seed ^ 0x5DEECE66D, task=self.task, max_traj_len=self.max_traj_len)
        # --imitate: frozen expert + the env's projector (reference rl/algos/ppo.py:111-122)
        self.base_policy, self.imitation_projector = None, None
        if getattr(args, "imitate", None):
            factory = getattr(spec, "imitation_projector", None)
            projector = factory() if callable(factory) else None
            if projector is None:
                raise ValueError(f"--imitate was passed but env {type(spec).__name__} does not implement "
                                 "imitation_projector(); cannot construct expert query.")
            from .imitation import FrozenActor
            self.base_policy = FrozenActor(args.imitate, self.device, max_rows=self.kernels.max_rows)
            self.imitation_projector = projector
        self.policy = self.kernels  # attribute names the reference's tests look for
        self.critic = self.kernels
        # The reference deep-copies the actor into old_policy every iteration only to evaluate the behaviour log-probs; here
        # they are stored by the rollout (and recomputed in float32 for the fp16-inference mode), so old_policy is the policy.
        self.old_policy = self.kernels
        names = list(getattr(self.kernels, "TENSORS", []))
        if self.recurrent:       # RnnKernels addresses its tensors by (name, offset, shape) specs
            specs = self.kernels.tensor_specs()
            names = list(specs)
            self.kernels._adam_view_specs = specs
        self.actor_optimizer = _AdamView(self.kernels, [n for n in names if n.startswith("a_") or n == "stds"], self.lr, self.eps)
        self.critic_optimizer = _AdamView(self.kernels, [n for n in names if n.startswith("c_")], self.lr, self.eps)
        self.last_losses = {}

    # ------------------------------------------------------------------ sampling
    def sample_parallel_with_workers(self, deterministic=False) -> BatchData:
        self.env.set_iteration(self.iteration_count)
        ro = self.rollout
        ro.collect(deterministic=deterministic)
        ret, adv = self.kernels.gae(ro.rew, ro.val, ro.done, ro.vterm, ro.vfinal, self.gamma, self.lam)
        self._adv, self._ret = adv, ret
        T, N = ro.T, ro.N
        rs, ls, cnt = ro.pop_episode_stats()
        self._ep_stats = dist_utils.global_episode_stats(rs, ls, cnt, device=self.device if _dist() and _dist().get_backend() == 'nccl' else None)
        return BatchData(states=ro.obs[:T].reshape(T * N, -1), actions=ro.act.reshape(T * N, -1),
                         rewards=ro.rew.reshape(T * N, 1), values=ro.val.reshape(T * N, 1), returns=ret.reshape(T * N, 1),
                         dones=ro.done.reshape(T * N, 1), traj_idx=self._traj_idx(ro.done),
                         ep_lens=torch.tensor([ls / cnt] if cnt else []), ep_rewards=torch.tensor([rs / cnt] if cnt else []), n_envs=N)

    @staticmethod
    def _traj_idx(done):
        """Trajectory boundaries like PPOBuffer.traj_idx (rl/storage/rollout_storage.py:24-51: [0, end of 1st trajectory, ...,
        number of samples]) for the ENV-MAJOR order of the batch -- sample (env n, step t) at n * T + t, i.e.
        `states.view(T, N, -1).transpose(0, 1)`: one env's T steps are contiguous there, and a trajectory ends where that env's
        episode ended or the buffer filled.  (The tensors of BatchData themselves are time-major, [T][N] flattened.)"""
        T, N = done.shape
        ends = (done.t() != 0)                      # [N][T]
        ends[:, T - 1] = True
        idx = torch.nonzero(ends.reshape(-1)).reshape(-1) + 1
        return torch.cat([torch.zeros(1, dtype=idx.dtype, device=idx.device), idx])

    # ------------------------------------------------------------------ update
    def _normalize_advantages(self, adv_flat):
        """(adv - mean) / (unbiased std + eps) over the GLOBAL batch (ppo.py:484-485)."""
        pack = dist_utils.global_moments_pack(self.kernels.moments(adv_flat), adv_flat.numel())   # stays on the device
        self.kernels.standardize(adv_flat, pack, self.eps)

    def update_actor_critic(self, obs_batch, action_batch, return_batch, advantage_batch, mask=1, mirror_observation=None,
                            mirror_action=None, old_log_probs=None):
        """One optimiser step on an explicit minibatch (same 7-tuple as the reference, ppo.py:299-406).
        ``mask`` must be 1 (FF path).  The mirror functions are configured at construction; the
        arguments are accepted for signature compatibility."""
        if self.recurrent:
            raise NotImplementedError("explicit padded-trajectory minibatches are not exposed for the LSTM path; use optimize()")
        k = self.kernels
        obs = obs_batch.to(self.device, torch.float32).contiguous()
        B = obs.shape[0]
        act = action_batch.to(self.device, torch.float32).contiguous()
        if old_log_probs is None:  # the reference recomputes them with old_policy == policy before the first step
            sd = k.get_tensors()["stds"].to(self.device)
            mu, _, _, _ = k.forward(obs, deterministic=True, want_value=False)
            old_log_probs = torch.distributions.Normal(mu, sd).log_prob(act).sum(-1)
        xn, xm = k.normalize(obs)
        k.stats.zero_()
        idx = torch.arange(B, dtype=torch.int32, device=self.device)
        k.grad_minibatch(xn, xm, act, old_log_probs.reshape(-1).to(self.device, torch.float32).contiguous(),
                         advantage_batch.reshape(-1).to(self.device, torch.float32).contiguous(),
                         return_batch.reshape(-1).to(self.device, torch.float32).contiguous(), idx,
                         imitation=self._imitation_term(obs))
        self._allreduce_and_apply()
        s = k.stats.cpu().numpy()
        return (float(s[0]), self._entropy_penalty(), float(s[1]), float(s[3]), float(s[2]), float(s[5]), float(s[4]))

    def _imitation_term(self, obs_mb):
        """Imitation loss inputs of one minibatch of RAW observations (ppo.py:360-368): ask the env's projector what the
        expert should see, run the frozen expert, scatter its means into the dense target the loss kernel reads."""
        if self.imitation_projector is None:
            return None
        from .imitation import dense_imitation_target
        query = self.imitation_projector(obs_mb)
        if not bool(query.sample_mask.any()):
            return None
        target = self.base_policy(query.expert_obs)
        dense, mask, count = dense_imitation_target(query, target, obs_mb.shape[0], self.spec.act_dim)
        return (self.imitate_coeff, dense, mask, count) if count > 0 else None

    def _entropy_penalty(self):
        sd = self.kernels.get_tensors()["stds"].numpy().astype(np.float64)
        return float(-np.mean(0.5 + 0.5 * np.log(2 * np.pi) + np.log(sd)))

    def _allreduce_and_apply(self):
        scale = dist_utils.allreduce_grad_(self.kernels.grad)  # RCCL sum over xGMI: the only data-path collective
        self.kernels.apply(grad_scale=scale)

    def _optimize_recurrent(self, itr: int):
        """Recurrent branch of PPO.train (ppo.py:512-533): `epochs` passes over shuffled minibatches of whole trajectories
        -- here `minibatch_size` env columns of the time-major rollout (each column = back-to-back trajectories, the LSTM
        state is reset where one starts), last partial minibatch kept (drop_last=False) -- BPTT on each."""
        k, ro = self.kernels, self.rollout
        T, N = ro.T, ro.N
        adv = self._adv.reshape(-1)
        self._normalize_advantages(adv)
        ret = self._ret.reshape(-1)
        xn, xm = k.normalize(ro.obs[:T].reshape(T * N, -1))
        act, logp = ro.act.reshape(T * N, -1), ro.logp.reshape(-1)
        mb = min(int(self.minibatch_size or N), N, k.seq_cols)
        k.stats.zero_()
        n_updates = 0
        for epoch in range(self.epochs):
            g = torch.Generator(device=self.device)
            g.manual_seed((self.seed if self.seed is not None else 0) + itr * self.epochs + epoch + 1000003 * self.rank)
            perm = torch.randperm(N, generator=g, device=self.device, dtype=torch.int64).to(torch.int32)
            for start in range(0, N, mb):
                k.grad_columns(T, N, xn, xm, act, logp, adv, ret, ro.done, perm[start:start + mb].contiguous())
                self._allreduce_and_apply()
                n_updates += 1
        s = (k.stats / max(1, n_updates)).cpu().numpy()
        self.last_losses = dict(actor=float(s[0]), critic=float(s[1]), mirror=float(s[2]), kl=float(s[3]), clip_fraction=float(s[4]),
                                imitation=0.0, entropy=self._entropy_penalty(), n_updates=n_updates)
        return self.last_losses

    def optimize(self, itr: int):
        """The per-iteration update of PPO.train (ppo.py:484-566): advantage normalisation, then
        `epochs` passes of shuffled minibatches (drop_last).  Returns dict of mean losses."""
        if self.recurrent:
            return self._optimize_recurrent(itr)
        k, ro = self.kernels, self.rollout
        T, N = ro.T, ro.N
        n_samples = T * N
        adv = self._adv.reshape(-1)
        self._normalize_advantages(adv)
        ret = self._ret.reshape(-1)
        raw_obs = ro.obs[:T].reshape(n_samples, -1)
        xn, xm = k.normalize(raw_obs)
        act = ro.act.reshape(n_samples, -1)
        logp = ro.logp.reshape(-1)
        if getattr(k, "inference_fp16", False) and not getattr(k, "update_fp16", False):
            # (with --fp16 the update evaluates the policy with the same fp16-operand GEMMs as the rollout: nothing to do)
            # The rollout's log-probs came out of the fp16-operand forward; the update evaluates the policy in float32, so
            # ratio != 1 on the very first minibatch from precision noise alone (amplified by 1 / std^2).  The reference
            # evaluates old_policy and policy in the same precision (ppo.py:305-309): recompute the old log-probs once per
            # iteration with the float32 actor, before any weight changes.
            logp = self._float32_log_probs(raw_obs, act)
        mb = int(self.minibatch_size or n_samples)
        mb = min(mb, n_samples)
        k.stats.zero_()
        n_updates = 0
        # Single process, no imitation term: an optimiser step is ONE graph launch (lhw_ppo_step) on a stream of its own -- a hipGraph is
        # captured from a stream, and torch's default stream is the legacy one, which cannot be captured.  With data parallelism the
        # gradient all-reduce sits between the two halves: lhw_ppo_grad, all-reduce, lhw_ppo_apply, eagerly.
        fused = (self.imitation_projector is None and not dist_utils._active(dist_utils.dist()) and hasattr(k._L, "lhw_ppo_step")
                 and os.environ.get("LHW_PPO_GRAPH", "1") != "0")
        cur = torch.cuda.current_stream(self.device)
        if fused:
            if getattr(self, "_update_stream", None) is None:
                self._update_stream = torch.cuda.Stream(device=self.device)
            self._update_stream.wait_stream(cur)
        with torch.cuda.stream(self._update_stream if fused else cur):
            for epoch in range(self.epochs):
                perm = self._minibatch_perm(itr, epoch, n_samples)
                for start in range(0, n_samples - mb + 1, mb):
                    mb_idx = perm[start:start + mb]
                    if fused:
                        k.step_minibatch(xn, xm, act, logp, adv, ret, mb_idx)
                    else:
                        imit = self._imitation_term(raw_obs.index_select(0, mb_idx.long())) if self.imitation_projector is not None else None
                        k.grad_minibatch(xn, xm, act, logp, adv, ret, mb_idx, imitation=imit)
                        self._allreduce_and_apply()
                    n_updates += 1
        if fused:
            cur.wait_stream(self._update_stream)
        s = (k.stats / max(1, n_updates)).cpu().numpy()
        self.last_losses = dict(actor=float(s[0]), critic=float(s[1]), mirror=float(s[2]), kl=float(s[3]),
                                clip_fraction=float(s[4]), imitation=float(s[5]), entropy=self._entropy_penalty(), n_updates=n_updates)
        return self.last_losses

    def _minibatch_perm(self, itr, epoch, n_samples, rank=None):
        """Shuffle of this rank's samples for one epoch (per-rank shuffles: statistically, not bitwise, the reference's global
        randperm, ppo.py:487-490); a pure function of (seed, iteration, epoch, rank)."""
        g = torch.Generator(device=self.device)
        base = self.seed if self.seed is not None else 0
        g.manual_seed(base + itr * self.epochs + epoch + 1000003 * (self.rank if rank is None else rank))
        return torch.randperm(n_samples, generator=g, device=self.device, dtype=torch.int64).to(torch.int32)

    def _float32_log_probs(self, obs_flat, act_flat):
        k = self.kernels
        sd = k.get_tensors()["stds"].to(self.device)
        out = _lib.empty(obs_flat.shape[0], dtype=torch.float32, device=self.device)
        k.set_inference_fp16(False)
        try:
            chunk = int(k.max_rows)
            for a in range(0, obs_flat.shape[0], chunk):
                b = min(a + chunk, obs_flat.shape[0])
                mu, _, _, _ = k.forward(obs_flat[a:b], deterministic=True, want_value=False)
                out[a:b] = torch.distributions.Normal(mu, sd).log_prob(act_flat[a:b]).sum(-1)
        finally:
            k.set_inference_fp16(True)
        return out

    def iterate(self, itr: int):
        """One PPO iteration: rollout + GAE + update.  Returns (n_samples_local, sample_time, optimize_time)."""
        self.iteration_count = itr
        t0 = time.time()
        batch = self.sample_parallel_with_workers()
        torch.cuda.synchronize(self.device)
        t1 = time.time()
        self.optimize(itr)
        torch.cuda.synchronize(self.device)
        t2 = time.time()
        return batch, t1 - t0, t2 - t1

    # ------------------------------------------------------------------ checkpoints / eval
    def save(self, itr, metric=None):
        """actor_{itr}.pt / critic_{itr}.pt (+ actor.pt / critic.pt when the metric improves), like
        ModelCheckpointer.save_if_best (reference rl/utils/checkpointer.py:54-83).  The files are whole-module
        pickles naming the reference's classes, so `run_experiment.py eval` / `--continued` of the reference load them."""
        if self.rank != 0:
            return
        from .checkpoint import save_recurrent_checkpoint, save_reference_checkpoint
        if self.recurrent:
            save_reference_checkpoint = save_recurrent_checkpoint    # Gaussian_LSTM_Actor / LSTM_V pickles
        t = self.kernels.get_tensors()
        om, osd = self.kernels.obs_mean.cpu(), self.kernels.obs_std.cpu()
        save_reference_checkpoint(t, om, osd, self.kernels.learn_std, self.save_path / f"actor_{itr}.pt", self.save_path / f"critic_{itr}.pt")
        if metric is not None and metric > self.best_metric:
            self.best_metric = metric
            save_reference_checkpoint(t, om, osd, self.kernels.learn_std, self.save_path / "actor.pt", self.save_path / "critic.pt")

    def evaluate(self, itr, num_batches=5):
        """5 deterministic batches on the same persistent envs (ppo.py:408-426)."""
        rs = ls = cnt = 0
        for _ in range(num_batches):
            self.sample_parallel_with_workers(deterministic=True)
            r, l, c = self._ep_stats
            rs, ls, cnt = rs + r, ls + l, cnt + c
        mean_r = rs / cnt if cnt else 0.0
        mean_l = ls / cnt if cnt else 0.0
        self.save(itr, mean_r)
        return mean_r, mean_l

    # ------------------------------------------------------------------ training loop
    def _init_seed(self, seed):
        """Seed of the weight initialisation: the run's seed, or -- like the reference without --seed -- a fresh random one
        (rank 0's under torch.distributed, so that every rank starts from the same weights)."""
        if seed is not None:
            return int(seed)
        s = [int.from_bytes(os.urandom(4), "little")]
        d = _dist()
        if d:
            d.broadcast_object_list(s, src=0)
        return s[0]

    def _log(self, msg):
        """progress lines of PPO.train (reference rl/algos/ppo.py:459-566) -- rank 0 only under torch.distributed"""
        if self.rank == 0:
            print(msg)

    def train(self, env_fn, n_itr):
        train_start = time.time()
        k = self.kernels
        if self.obs_rms is not None:
            self._log("Warming up observation normalization...")
            for i in range(5):  # ppo.py:442-457
                batch = self.sample_parallel_with_workers()
                mean, var, n = dist_utils.global_batch_moments(batch.states)  # == update on the concatenated batch
                self.obs_rms.update_from_moments(mean.cpu().numpy(), var.cpu().numpy(), n)
                self._log(f"  Warmup batch {i + 1}: {int(n)} samples, obs_rms count: {self.obs_rms.count:.0f}")
            k.set_obs_norm(self.obs_rms.mean, self.obs_rms.std)
            self._log(f"Normalization initialized with {self.obs_rms.count:.0f} samples")
        for itr in range(n_itr):
            self._log(f"********** Iteration {itr} ************")
            batch, sample_time, optimize_time = self.iterate(itr)
            num_samples = batch.states.shape[0] * self.world
            self._log(f"Sampling took {sample_time:.2f}s for {num_samples} steps.")
            self._log(f"Optimizer took: {optimize_time:.2f}s")
            self.total_steps += num_samples
            L = self.last_losses
            rs, ls, cnt = self._ep_stats
            mean_eprew = rs / cnt if cnt else float("nan")
            mean_eplen = ls / cnt if cnt else float("nan")
            noise = float(np.mean(k.get_tensors()["stds"].numpy()))
            if self.rank == 0:
                w = sys.stdout.write
                w("-" * 37 + "\n")
                w(f"| {'Mean Eprew':>15} | {mean_eprew:>15.5g} |\n")
                w(f"| {'Mean Eplen':>15} | {mean_eplen:>15.5g} |\n")
                w(f"| {'Actor loss':>15} | {L['actor']:>15.3g} |\n")
                w(f"| {'Critic loss':>15} | {L['critic']:>15.3g} |\n")
                w(f"| {'Mirror loss':>15} | {L['mirror']:>15.3g} |\n")
                w(f"| {'Imitation loss':>15} | {L.get('imitation', 0.0):>15.3g} |\n")
                w(f"| {'Mean KL Div':>15} | {L['kl']:>15.3g} |\n")
                w(f"| {'Mean Entropy':>15} | {L['entropy']:>15.3g} |\n")
                w(f"| {'Clip Fraction':>15} | {L['clip_fraction']:>15.3g} |\n")
                w(f"| {'Mean noise std':>15} | {noise:>15.3g} |\n")
                w("-" * 37 + "\n")
                sys.stdout.flush()
            total_time = time.time() - train_start
            fps = self.total_steps / total_time
            iter_avg = total_time / (itr + 1)
            eta = round((n_itr - itr) * iter_avg)
            self._log(f"Total time elapsed: {total_time:.2f}s. Total steps: {self.total_steps} "
                      f"(fps={fps:.2f}. iter-avg={iter_avg:.2f}s. ETA={datetime.timedelta(seconds=eta)})")
            if itr == 0 or (itr + 1) % self.eval_freq == 0:
                t0 = time.time()
                mean_r, mean_l = self.evaluate(itr)
                self._log("====EVALUATE EPISODE====")
                self._log(f"(Episode length:{mean_l:.3f}. Reward:{mean_r:.3f}. Time taken:{time.time() - t0:.2f}s)")
