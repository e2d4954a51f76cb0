"""Checkpoint files are loadable by the REFERENCE's own classes and vice versa.  CPU."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from learninghumanoidwalking_amd import checkpoint as ck
from learninghumanoidwalking_amd.ppo_kernels import reference_init

REF = "/root/reference"
NAMES = ["w1", "b1", "w2", "b2", "w3", "b3"]


def _tensors(seed=3):
    t = reference_init(37, 12, 256, 0.223, generator_seed=seed)
    t["a_b1"] += 0.05
    t["c_b3"] += 0.2
    return t


def test_roundtrip_without_reference_on_path(tmp_path):
    t = _tensors()
    om, osd = torch.randn(37), torch.rand(37) + 0.5
    a, c = tmp_path / "actor_7.pt", tmp_path / "critic_7.pt"
    ck.save_reference_checkpoint(t, om, osd, False, a, c)
    assert "rl.policies.actor" not in sys.modules or os.path.isdir(REF)   # stand-in modules are removed again
    t2, om2, os2 = ck.load_reference_checkpoint(a, c)
    for k in t:
        assert torch.equal(t[k].float(), t2[k]), k
    assert torch.equal(om, om2) and torch.equal(osd, os2)
    raw = open(a, "rb").read()
    assert b"rl.policies.actor" in raw and b"Gaussian_FF_Actor" in raw    # pickled under the reference's class path


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "rl")), reason="reference checkout not present")
def test_reference_classes_load_our_files_and_compute_the_same(tmp_path):
    from oracle import ppo_oracle as po
    t = _tensors()
    om, osd = torch.randn(37) * 0.1, torch.rand(37) + 0.5
    a, c = tmp_path / "actor_0.pt", tmp_path / "critic_0.pt"
    ck.save_reference_checkpoint(t, om, osd, True, a, c)
    obs = torch.randn(5, 37)
    torch.save(obs, tmp_path / "obs.pt")
    code = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "from rl.policies.actor import Gaussian_FF_Actor; from rl.policies.critic import FF_V\n"
        "p = torch.load(%r, weights_only=False); v = torch.load(%r, weights_only=False)\n"
        "assert type(p) is Gaussian_FF_Actor and type(v) is FF_V, (type(p), type(v))\n"
        "obs = torch.load(%r)\n"
        "torch.save((p(obs, deterministic=True).detach(), v(obs).detach(), p.distribution(obs).stddev.detach()), %r)\n"
        % (REF, str(a), str(c), str(tmp_path / "obs.pt"), str(tmp_path / "out.pt")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    mu, val, sd = torch.load(tmp_path / "out.pt")
    xn = (obs - om) / osd
    np.testing.assert_allclose(mu.numpy(), po.mlp(xn, *[t[f"a_{n}"] for n in NAMES]).numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(val.numpy(), po.mlp(xn, *[t[f"c_{n}"] for n in NAMES]).numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(sd.numpy(), np.full((5, 12), 0.223, dtype=np.float32), rtol=1e-6)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "rl")), reason="reference checkout not present")
def test_we_load_files_written_by_the_reference(tmp_path):
    code = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "from rl.policies.actor import Gaussian_FF_Actor; from rl.policies.critic import FF_V\n"
        "torch.manual_seed(5); p = Gaussian_FF_Actor(37, 12, init_std=0.2, learn_std=False); v = FF_V(37)\n"
        "p.obs_mean = torch.zeros(37); p.obs_std = torch.ones(37); v.obs_mean = p.obs_mean; v.obs_std = p.obs_std\n"
        "torch.save(p, %r); torch.save(v, %r)\n" % (REF, str(tmp_path / "actor_3.pt"), str(tmp_path / "critic_3.pt")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    t, om, osd = ck.load_reference_checkpoint(tmp_path / "actor_3.pt", tmp_path / "critic_3.pt")
    ref = reference_init(37, 12, 256, 0.2, generator_seed=5)   # same torch seed -> same constructor draws
    for k in ref:
        assert torch.equal(ref[k].float(), t[k]), k


# ----------------------------------------------------------------------------- recurrent (LSTM) checkpoints
def _lstm_tensors(seed=4, H=32):
    from learninghumanoidwalking_amd.rnn_kernels import reference_init_lstm
    return reference_init_lstm(37, 12, H, 0.2, generator_seed=seed)


def test_recurrent_roundtrip_without_reference_on_path(tmp_path):
    t = _lstm_tensors()
    om, osd = torch.randn(37), torch.rand(37) + 0.5
    a, c = tmp_path / "actor_1.pt", tmp_path / "critic_1.pt"
    ck.save_recurrent_checkpoint(t, om, osd, False, a, c)
    t2, om2, os2, hidden = ck.load_recurrent_checkpoint(a, c)
    assert hidden == 32 and torch.equal(om, om2) and torch.equal(osd, os2)
    for k in t:
        assert torch.equal(t[k].float(), t2[k]), k
    raw = open(a, "rb").read()
    assert b"rl.policies.actor" in raw and b"Gaussian_LSTM_Actor" in raw


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "rl")), reason="reference checkout not present")
def test_reference_lstm_classes_load_our_files_and_we_load_theirs(tmp_path):
    """Gaussian_LSTM_Actor / LSTM_V of the reference unpickle our files and produce the oracle's sequence outputs; files
    written by the reference load back, and the same torch seed reproduces the reference's constructor draws."""
    from oracle import ppo_oracle as po
    t = _lstm_tensors()
    om, osd = torch.randn(37) * 0.1, torch.rand(37) + 0.5
    a, c = tmp_path / "actor_0.pt", tmp_path / "critic_0.pt"
    ck.save_recurrent_checkpoint(t, om, osd, False, a, c)
    obs = torch.randn(6, 3, 37)           # [T, B, D]: a batch of trajectories
    torch.save(obs, tmp_path / "obs.pt")
    code = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "from rl.policies.actor import Gaussian_LSTM_Actor; from rl.policies.critic import LSTM_V\n"
        "p = torch.load(%r, weights_only=False); v = torch.load(%r, weights_only=False)\n"
        "assert type(p) is Gaussian_LSTM_Actor and type(v) is LSTM_V, (type(p), type(v))\n"
        "obs = torch.load(%r)\n"
        "torch.save((p(obs, deterministic=True).detach(), v(obs).detach()), %r)\n"
        "torch.manual_seed(9); q = Gaussian_LSTM_Actor(37, 12, layers=(32, 32), init_std=0.2); w = LSTM_V(37, layers=(32, 32))\n"
        "q.obs_mean = torch.zeros(37); q.obs_std = torch.ones(37); w.obs_mean = q.obs_mean; w.obs_std = q.obs_std\n"
        "torch.save(q, %r); torch.save(w, %r)\n"
        % (REF, str(a), str(c), str(tmp_path / "obs.pt"), str(tmp_path / "out.pt"), str(tmp_path / "actor_9.pt"), str(tmp_path / "critic_9.pt")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    mu, val = torch.load(tmp_path / "out.pt")
    names = ["wih1", "whh1", "bih1", "bhh1", "wih2", "whh2", "bih2", "bhh2", "wout", "bout"]
    reset = torch.zeros(6, 3, dtype=torch.bool)
    reset[0] = True
    xn = (obs - om) / osd
    np.testing.assert_allclose(mu.numpy(), po.lstm_net(xn, reset, [t[f"a_{n}"] for n in names]).numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(val.numpy(), po.lstm_net(xn, reset, [t[f"c_{n}"] for n in names]).numpy(), rtol=1e-5, atol=1e-6)
    t9, _, _, hidden = ck.load_recurrent_checkpoint(tmp_path / "actor_9.pt", tmp_path / "critic_9.pt")
    ref = _lstm_tensors(seed=9)
    assert hidden == 32
    for k in ref:
        assert torch.equal(ref[k].float(), t9[k]), k
