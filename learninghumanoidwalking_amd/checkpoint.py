"""Checkpoint interop with the reference.

The reference saves *whole modules* with ``torch.save(model, path)`` (reference rl/utils/checkpointer.py:38-52) and
loads them back in ``run_experiment.py eval`` (:268-276) and ``--continued`` (rl/algos/ppo.py:69-82), so a checkpoint
is a pickle that names the classes ``rl.policies.actor.Gaussian_FF_Actor`` and ``rl.policies.critic.FF_V`` and carries
their instance ``__dict__``.  This module writes / reads exactly that format from our flat parameter tensors, without
needing the reference on the path: when ``rl.policies`` is not importable, attribute-compatible stand-in classes are
registered under the same module paths for the duration of the (un)pickling.
"""
from __future__ import annotations

import contextlib
import importlib
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

_PATHS = {"actor": ("rl.policies.actor", "Gaussian_FF_Actor"), "critic": ("rl.policies.critic", "FF_V"),
          "lstm_actor": ("rl.policies.actor", "Gaussian_LSTM_Actor"), "lstm_critic": ("rl.policies.critic", "LSTM_V")}


class _StandInBase(nn.Module):
    """Forward passes equal to the reference's (actor.py:160-184, critic.py:41-49) so a stand-in object is usable too."""

    def _norm(self, state):
        return (state - self.obs_mean) / self.obs_std


def _make_standins():
    def actor_forward(self, state, deterministic=True):
        x = self._norm(state)
        for layer in self.actor_layers:
            x = self.nonlinearity(layer(x))
        mu = self.means(x)
        return mu if deterministic else torch.distributions.Normal(mu, self.stds).sample()

    def critic_forward(self, state):
        x = self._norm(state)
        for layer in self.critic_layers:
            x = self.nonlinearity(layer(x))
        return self.network_out(x)

    def init_hidden_state(self, batch_size=1, device=None):
        layers = self.actor_layers if hasattr(self, "actor_layers") else self.critic_layers
        self.hidden = [torch.zeros(batch_size, l.hidden_size) for l in layers]
        self.cells = [torch.zeros(batch_size, l.hidden_size) for l in layers]

    def lstm_forward(self, state, deterministic=True):
        # single time step or batch of single steps (actor.py:247-262, critic.py:96-112); sequences go step by step
        x = self._norm(state)
        flat = x.dim() == 1
        x = x.view(1, -1) if flat else x
        layers = self.actor_layers if hasattr(self, "actor_layers") else self.critic_layers
        for idx, layer in enumerate(layers):
            self.hidden[idx], self.cells[idx] = layer(x, (self.hidden[idx], self.cells[idx]))
            x = self.hidden[idx]
        y = self.network_out(x)
        return y.view(-1) if flat else y

    A = type("Gaussian_FF_Actor", (_StandInBase,), {"forward": actor_forward})
    C = type("FF_V", (_StandInBase,), {"forward": critic_forward})
    LA = type("Gaussian_LSTM_Actor", (_StandInBase,), {"forward": lstm_forward, "init_hidden_state": init_hidden_state})
    LC = type("LSTM_V", (_StandInBase,), {"forward": lstm_forward, "init_hidden_state": init_hidden_state})
    A.__module__, C.__module__ = _PATHS["actor"][0], _PATHS["critic"][0]
    LA.__module__, LC.__module__ = _PATHS["lstm_actor"][0], _PATHS["lstm_critic"][0]
    return {"actor": A, "critic": C, "lstm_actor": LA, "lstm_critic": LC}


@contextlib.contextmanager
def reference_classes():
    """Yield {"actor": cls, "critic": cls} that pickle as the reference's class paths."""
    try:
        real = {k: getattr(importlib.import_module(m), c) for k, (m, c) in _PATHS.items()}
        yield real
        return
    except Exception:
        pass
    saved = {n: sys.modules.get(n) for n in ("rl", "rl.policies", "rl.policies.actor", "rl.policies.critic")}
    cls = _make_standins()
    try:
        rl, pol = types.ModuleType("rl"), types.ModuleType("rl.policies")
        act, cri = types.ModuleType("rl.policies.actor"), types.ModuleType("rl.policies.critic")
        rl.policies, pol.actor, pol.critic = pol, act, cri
        act.Gaussian_FF_Actor, cri.FF_V = cls["actor"], cls["critic"]
        act.Gaussian_LSTM_Actor, cri.LSTM_V = cls["lstm_actor"], cls["lstm_critic"]
        sys.modules.update({"rl": rl, "rl.policies": pol, "rl.policies.actor": act, "rl.policies.critic": cri})
        yield cls
    finally:
        for n, m in saved.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m


def _linear(w, b):
    lin = nn.Linear(w.shape[1], w.shape[0])
    with torch.no_grad():
        lin.weight.copy_(torch.as_tensor(w, dtype=torch.float32))
        lin.bias.copy_(torch.as_tensor(b, dtype=torch.float32))
    return lin


def build_modules(tensors: dict, obs_mean, obs_std, learn_std: bool, classes: dict):
    """Reference-shaped module objects from ``PpoKernels.get_tensors()`` output."""
    om, os_ = torch.as_tensor(obs_mean, dtype=torch.float32).clone(), torch.as_tensor(obs_std, dtype=torch.float32).clone()
    actor = classes["actor"].__new__(classes["actor"])
    nn.Module.__init__(actor)
    actor.actor_layers = nn.ModuleList([_linear(tensors["a_w1"], tensors["a_b1"]), _linear(tensors["a_w2"], tensors["a_b2"])])
    actor.means = _linear(tensors["a_w3"], tensors["a_b3"])
    actor.learn_std = bool(learn_std)
    stds = torch.as_tensor(tensors["stds"], dtype=torch.float32).clone()
    actor.stds = nn.Parameter(stds) if learn_std else stds
    actor.action_dim, actor.state_dim = int(tensors["a_w3"].shape[0]), int(tensors["a_w1"].shape[1])
    actor.nonlinearity = F.relu
    actor.obs_std, actor.obs_mean = os_, om
    actor.bounded, actor.normc_init = False, True
    critic = classes["critic"].__new__(classes["critic"])
    nn.Module.__init__(critic)
    critic.critic_layers = nn.ModuleList([_linear(tensors["c_w1"], tensors["c_b1"]), _linear(tensors["c_w2"], tensors["c_b2"])])
    critic.network_out = _linear(tensors["c_w3"], tensors["c_b3"])
    critic.nonlinearity = F.relu
    critic.obs_std, critic.obs_mean = os_.clone(), om.clone()
    critic.normc_init = True
    return actor, critic


def save_reference_checkpoint(tensors, obs_mean, obs_std, learn_std, actor_path, critic_path):
    with reference_classes() as cls:
        actor, critic = build_modules(tensors, obs_mean, obs_std, learn_std, cls)
        torch.save(actor, actor_path)
        torch.save(critic, critic_path)


def load_reference_checkpoint(actor_path, critic_path):
    """-> (tensors dict in PpoKernels names, obs_mean, obs_std).  Reads files written by the reference or by us."""
    with reference_classes():
        actor = torch.load(actor_path, weights_only=False, map_location="cpu")
        critic = torch.load(critic_path, weights_only=False, map_location="cpu")
    a, c = actor.actor_layers, critic.critic_layers
    t = dict(a_w1=a[0].weight, a_b1=a[0].bias, a_w2=a[1].weight, a_b2=a[1].bias, a_w3=actor.means.weight, a_b3=actor.means.bias,
             c_w1=c[0].weight, c_b1=c[0].bias, c_w2=c[1].weight, c_b2=c[1].bias, c_w3=critic.network_out.weight,
             c_b3=critic.network_out.bias)
    t = {k: v.detach().float().clone() for k, v in t.items()}
    if hasattr(actor, "stds"):
        t["stds"] = torch.as_tensor(actor.stds).detach().float().clone()
    return t, torch.as_tensor(actor.obs_mean).float(), torch.as_tensor(actor.obs_std).float()


def load_reference_actor(actor_path):
    """Actor-only load (the expert of ``--imitate``): -> (tensors a_*, obs_mean, obs_std, hidden width)."""
    with reference_classes():
        actor = torch.load(actor_path, weights_only=False, map_location="cpu")
    a = actor.actor_layers
    if len(a) != 2 or a[0].weight.shape[0] != a[1].weight.shape[0]:
        raise ValueError("only two equal-width hidden layers are supported for the expert policy")
    t = dict(a_w1=a[0].weight, a_b1=a[0].bias, a_w2=a[1].weight, a_b2=a[1].bias, a_w3=actor.means.weight, a_b3=actor.means.bias)
    t = {k: v.detach().float().clone() for k, v in t.items()}
    return t, torch.as_tensor(actor.obs_mean).float(), torch.as_tensor(actor.obs_std).float(), int(a[0].weight.shape[0])


# ----------------------------------------------------------------------------- recurrent (LSTM) policies
def _lstm_cell(w_ih, w_hh, b_ih, b_hh):
    cell = nn.LSTMCell(w_ih.shape[1], w_hh.shape[1])
    with torch.no_grad():
        for dst, src in ((cell.weight_ih, w_ih), (cell.weight_hh, w_hh), (cell.bias_ih, b_ih), (cell.bias_hh, b_hh)):
            dst.copy_(torch.as_tensor(src, dtype=torch.float32))
    return cell


def save_recurrent_checkpoint(tensors, obs_mean, obs_std, learn_std, actor_path, critic_path):
    """Whole-module pickles naming rl.policies.actor.Gaussian_LSTM_Actor / rl.policies.critic.LSTM_V (reference
    rl/policies/actor.py:191-232, critic.py:52-66) from ``RnnKernels.get_tensors()``."""
    om, os_ = torch.as_tensor(obs_mean, dtype=torch.float32).clone(), torch.as_tensor(obs_std, dtype=torch.float32).clone()
    with reference_classes() as cls:
        actor = cls["lstm_actor"].__new__(cls["lstm_actor"])
        nn.Module.__init__(actor)
        t = tensors
        actor.actor_layers = nn.ModuleList([_lstm_cell(t["a_wih1"], t["a_whh1"], t["a_bih1"], t["a_bhh1"]),
                                            _lstm_cell(t["a_wih2"], t["a_whh2"], t["a_bih2"], t["a_bhh2"])])
        actor.network_out = _linear(t["a_wout"], t["a_bout"])
        actor.action_dim, actor.state_dim = int(t["a_wout"].shape[0]), int(t["a_wih1"].shape[1])
        actor.nonlinearity = torch.tanh
        actor.obs_std, actor.obs_mean = os_, om
        actor.learn_std = bool(learn_std)
        stds = torch.as_tensor(t["stds"], dtype=torch.float32).clone()
        actor.stds = nn.Parameter(stds) if learn_std else stds
        actor.bounded, actor.normc_init = False, True
        critic = cls["lstm_critic"].__new__(cls["lstm_critic"])
        nn.Module.__init__(critic)
        critic.critic_layers = nn.ModuleList([_lstm_cell(t["c_wih1"], t["c_whh1"], t["c_bih1"], t["c_bhh1"]),
                                              _lstm_cell(t["c_wih2"], t["c_whh2"], t["c_bih2"], t["c_bhh2"])])
        critic.network_out = _linear(t["c_wout"], t["c_bout"])
        critic.obs_std, critic.obs_mean = os_.clone(), om.clone()
        critic.normc_init = True
        for net in (actor, critic):
            layers = net.actor_layers if hasattr(net, "actor_layers") else net.critic_layers
            net.hidden = [torch.zeros(1, l.hidden_size) for l in layers]
            net.cells = [torch.zeros(1, l.hidden_size) for l in layers]
        torch.save(actor, actor_path)
        torch.save(critic, critic_path)


def load_recurrent_checkpoint(actor_path, critic_path):
    """-> (tensors in RnnKernels names, obs_mean, obs_std, hidden width)."""
    with reference_classes():
        actor = torch.load(actor_path, weights_only=False, map_location="cpu")
        critic = torch.load(critic_path, weights_only=False, map_location="cpu")
    t = {}
    for pre, layers, outl in (("a", actor.actor_layers, actor.network_out), ("c", critic.critic_layers, critic.network_out)):
        if len(layers) != 2 or layers[0].hidden_size != layers[1].hidden_size:
            raise ValueError("only two equal-width LSTM layers are supported")
        for k, cell in enumerate(layers, 1):
            t[f"{pre}_wih{k}"], t[f"{pre}_whh{k}"] = cell.weight_ih, cell.weight_hh
            t[f"{pre}_bih{k}"], t[f"{pre}_bhh{k}"] = cell.bias_ih, cell.bias_hh
        t[f"{pre}_wout"], t[f"{pre}_bout"] = outl.weight, outl.bias
    t = {k: v.detach().float().clone() for k, v in t.items()}
    t["stds"] = torch.as_tensor(actor.stds).detach().float().clone()
    return t, torch.as_tensor(actor.obs_mean).float(), torch.as_tensor(actor.obs_std).float(), int(actor.actor_layers[0].hidden_size)
