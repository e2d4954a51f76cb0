"""ctypes view of the recurrent (LSTM) PPO kernels (include/lhw.h, lhw_rnn_*): Gaussian_LSTM_Actor / LSTM_V of the
reference (rl/policies/actor.py:191-286, critic.py:52-112) as one flat float32 parameter vector on the GPU."""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib
from .ppo_kernels import LhwPpoConfig, PpoKernels, _p, _setup

_SETUP_RNN = False


def _setup_rnn(L):
    global _SETUP_RNN
    if _SETUP_RNN:
        return
    vp, i32, i64, f32, u32, u64 = (ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_uint32, ctypes.c_uint64)
    L.lhw_rnn_create.argtypes = [ctypes.POINTER(LhwPpoConfig), i32, i32, i32, ctypes.POINTER(vp)]
    L.lhw_rnn_destroy.argtypes = [vp]
    L.lhw_rnn_param_count.argtypes = [vp]
    L.lhw_rnn_param_count.restype = i64
    L.lhw_rnn_layout.argtypes = [vp, ctypes.POINTER(i64)]
    L.lhw_rnn_forward.argtypes = [vp, vp, vp, i64, vp, vp, vp, u64, u32, u32, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp]
    L.lhw_rnn_grad.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp]
    L.lhw_rnn_apply.argtypes = [vp, vp, vp, vp, vp, i64, f32, vp]
    L.lhw_rnn_normalize.argtypes = [vp, vp, i64, vp, vp, vp, vp, vp]
    _SETUP_RNN = True


class RnnKernels:
    """Two stacked LSTM cells + linear read-out for the actor and for the critic.

    Tensor names (torch layouts): ``{a,c}_wih1 [4H, D]``, ``_whh1 [4H, H]``, ``_bih1``, ``_bhh1``, ``_wih2 [4H, H]``,
    ``_whh2``, ``_bih2``, ``_bhh2``, ``_wout [A | 1, H]``, ``_bout``, and ``stds``."""

    NET = ["wcat1", "bih1", "bhh1", "wcat2", "bih2", "bhh2", "wout", "bout"]

    def __init__(self, obs_dim, act_dim, *, hidden=256, seq_len=400, seq_cols=64, rollout_rows=4096, device=0, learn_std=False,
                 lr=3e-4, eps=1e-5, clip=0.2, entropy_coeff=0.0, mirror_coeff=0.4, max_grad_norm=0.5, mirror_obs=None,
                 mirror_act=None):
        if not torch.cuda.is_available():
            raise _lib.LhwError(-5, "no GPU visible: the PPO kernels have no CPU fallback")
        self.device = torch.device("cuda", device) if isinstance(device, int) else device
        L = _lib.lib()
        _setup(L)
        _setup_rnn(L)
        self._L = L
        cfg = LhwPpoConfig()
        cfg.device = self.device.index or 0
        cfg.obs_dim, cfg.act_dim, cfg.hidden = obs_dim, act_dim, hidden
        cfg.learn_std, cfg.max_rows = int(learn_std), int(rollout_rows)
        cfg.lr, cfg.eps, cfg.clip = lr, eps, clip
        cfg.entropy_coeff, cfg.mirror_coeff, cfg.max_grad_norm = entropy_coeff, mirror_coeff, max_grad_norm
        self._keep = []
        if mirror_obs is not None:
            (os_, og), (as_, ag) = mirror_obs, mirror_act
            arrs = [np.ascontiguousarray(os_, np.int32), np.ascontiguousarray(og, np.float32),
                    np.ascontiguousarray(as_, np.int32), np.ascontiguousarray(ag, np.float32)]
            self._keep = arrs
            cfg.mirror_obs_src, cfg.mirror_obs_sign, cfg.mirror_act_src, cfg.mirror_act_sign = [a.ctypes.data for a in arrs]
        self.use_mirror = mirror_obs is not None
        self._h = ctypes.c_void_p()
        _lib.check(L.lhw_rnn_create(ctypes.byref(cfg), int(seq_len), int(seq_cols), int(rollout_rows), ctypes.byref(self._h)))
        self.obs_dim, self.act_dim, self.hidden, self.learn_std = obs_dim, act_dim, hidden, learn_std
        self.seq_len, self.seq_cols, self.max_rows, self.eps = seq_len, seq_cols, rollout_rows, eps
        self.n_params = int(L.lhw_rnn_param_count(self._h))
        lay = (ctypes.c_int64 * 19)()
        _lib.check(L.lhw_rnn_layout(self._h, lay))
        self.offsets = list(lay)
        self.Dp, self.Op = int(lay[17]), int(lay[18])
        dev = self.device
        self.theta = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        self.grad = torch.zeros_like(self.theta)
        self.adam_m = torch.zeros_like(self.theta)
        self.adam_v = torch.zeros_like(self.theta)
        self.adam_step = 0
        self.stats = torch.zeros(16, dtype=torch.float32, device=dev)
        self.obs_mean = torch.zeros(obs_dim, dtype=torch.float32, device=dev)
        self.obs_std = torch.ones(obs_dim, dtype=torch.float32, device=dev)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.lhw_rnn_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- parameter views
    def _blocks(self, net):
        """name -> (offset, rows, ld, col0, cols) of every torch-layout tensor of one network inside theta."""
        H, D, Dp = self.hidden, self.obs_dim, self.Dp
        base = 0 if net == "a" else 9
        o = self.offsets[base:base + 8]
        out_rows, out_ld = (self.act_dim, self.Op) if net == "a" else (1, 4)
        K1 = Dp + H
        return {
            f"{net}_wih1": (o[0], 4 * H, K1, 0, D), f"{net}_whh1": (o[0], 4 * H, K1, Dp, H), f"{net}_bih1": (o[1], 4 * H), f"{net}_bhh1": (o[2], 4 * H),
            f"{net}_wih2": (o[3], 4 * H, 2 * H, 0, H), f"{net}_whh2": (o[3], 4 * H, 2 * H, H, H), f"{net}_bih2": (o[4], 4 * H), f"{net}_bhh2": (o[5], 4 * H),
            f"{net}_wout": (o[6], out_rows, H, 0, H), f"{net}_bout": (o[7], out_rows),
        }

    def _view(self, flat, spec):
        if len(spec) == 2:
            return flat[spec[0]:spec[0] + spec[1]]
        off, rows, ld, c0, cols = spec
        return flat[off:off + rows * ld].view(rows, ld)[:, c0:c0 + cols]

    def tensor_specs(self):
        d = {}
        d.update(self._blocks("a"))
        d["stds"] = (self.offsets[8], self.act_dim)
        d.update(self._blocks("c"))
        return d

    def get_tensors(self, flat=None):
        flat = self.theta if flat is None else flat
        return {n: self._view(flat, sp).detach().cpu().clone() for n, sp in self.tensor_specs().items()}

    def set_tensors(self, tensors: dict):
        specs = self.tensor_specs()
        for n, t in tensors.items():
            v = self._view(self.theta, specs[n])
            v.copy_(torch.as_tensor(t, dtype=torch.float32).reshape(v.shape))

    def set_obs_norm(self, mean, std):
        self.obs_mean.copy_(torch.as_tensor(np.asarray(mean), dtype=torch.float32))
        self.obs_std.copy_(torch.as_tensor(np.asarray(std), dtype=torch.float32))

    # ---- kernels
    def forward(self, obs, *, reset=None, seed=0, env_id_base=0, counter=0, deterministic=False, commit=True, want_actor=True,
                want_value=True, mu=None, act=None, logp=None, value=None):
        N, dev = obs.shape[0], self.device
        if want_actor:
            mu = _lib.empty(N, self.act_dim, dtype=torch.float32, device=dev) if mu is None else mu
            act = _lib.empty(N, self.act_dim, dtype=torch.float32, device=dev) if act is None else act
            logp = _lib.empty(N, dtype=torch.float32, device=dev) if logp is None else logp
        if want_value:
            value = _lib.empty(N, dtype=torch.float32, device=dev) if value is None else value
        _lib.check(self._L.lhw_rnn_forward(self._h, _p(self.theta), _p(obs), N, _p(self.obs_mean), _p(self.obs_std), _p(reset),
                                           int(seed) & (2**64 - 1), int(env_id_base), int(counter), int(deterministic), int(commit),
                                           _p(mu) if want_actor else None, _p(act) if want_actor else None,
                                           _p(logp) if want_actor else None, _p(value) if want_value else None, self._stream()))
        return mu, act, logp, value

    def normalize(self, obs, want_mirror=None):
        R = obs.shape[0]
        want_mirror = self.use_mirror if want_mirror is None else want_mirror
        xn = _lib.empty(R, self.Dp, dtype=torch.float32, device=self.device)
        xm = _lib.empty(R, self.Dp, dtype=torch.float32, device=self.device) if want_mirror else None
        _lib.check(self._L.lhw_rnn_normalize(self._h, _p(obs), R, _p(self.obs_mean), _p(self.obs_std), _p(xn), _p(xm), self._stream()))
        return xn, xm

    def gae(self, rew, val, done, vterm, vfinal, gamma, lam):
        T, N = rew.shape
        ret = _lib.empty(T, N, dtype=torch.float32, device=self.device)
        adv = _lib.empty(T, N, dtype=torch.float32, device=self.device)
        _lib.check(self._L.lhw_gae(T, N, _p(rew), _p(val), _p(done), _p(vterm), _p(vfinal), float(gamma), float(lam), _p(ret), _p(adv),
                                   self._stream()))
        return ret, adv

    def moments(self, x):
        if not hasattr(self, "_mom"):
            self._mom = torch.zeros(2, dtype=torch.float64, device=self.device)
        _lib.check(self._L.lhw_moments(_p(x), x.numel(), _p(self._mom), self._stream()))
        return self._mom

    standardize = PpoKernels.standardize

    def scale_shift(self, x, mean, inv):
        _lib.check(self._L.lhw_scale_shift(_p(x), x.numel(), float(mean), float(inv), self._stream()))

    def grad_columns(self, T, N, xn, xm, act, old_logp, adv, ret, done, cols):
        """BPTT over columns ``cols`` (int32 device tensor) of the time-major [T][N] rollout."""
        _lib.check(self._L.lhw_rnn_grad(self._h, _p(self.theta), _p(self.grad), int(T), int(N), _p(xn), _p(xm), _p(act), _p(old_logp),
                                        _p(adv), _p(ret), _p(done), _p(cols), int(cols.numel()), _p(self.stats), self._stream()))

    def apply(self, grad_scale=1.0):
        self.adam_step += 1
        _lib.check(self._L.lhw_rnn_apply(self._h, _p(self.theta), _p(self.grad), _p(self.adam_m), _p(self.adam_v), self.adam_step,
                                         float(grad_scale), self._stream()))


def reference_init_lstm(obs_dim, act_dim, hidden=256, init_std=0.2, generator_seed=None):
    """Initial weights with the RNG consumption of the reference's recurrent constructors (reference
    rl/policies/actor.py:191-232, critic.py:52-66, base.py:5-22): per network two nn.LSTMCell (default uniform init, not
    touched by normc_fn) and one nn.Linear read-out, then normc on the Linear (actor read-out x0.01); actor first."""
    import torch.nn as nn
    # the draws come from the CPU default generator, seeded here and restored afterwards; the CUDA generators are left alone
    # (torch.manual_seed would reseed them too)
    with torch.random.fork_rng(devices=[]):
        if generator_seed is not None:
            torch.default_generator.manual_seed(int(generator_seed))
        return _reference_init_lstm_draw(obs_dim, act_dim, hidden, init_std)


def _reference_init_lstm_draw(obs_dim, act_dim, hidden, init_std):
    import torch.nn as nn

    def net(out_dim, scale_out):
        cells = [nn.LSTMCell(obs_dim, hidden), nn.LSTMCell(hidden, hidden)]
        lin = nn.Linear(hidden, out_dim)
        lin.weight.data.normal_(0, 1)
        lin.weight.data *= 1 / torch.sqrt(lin.weight.data.pow(2).sum(1, keepdim=True))
        lin.bias.data.fill_(0)
        if scale_out is not None:
            lin.weight.data.mul_(scale_out)
        return cells, lin

    out = {}
    for pre, (cells, lin) in (("a", net(act_dim, 0.01)), ("c", net(1, None))):
        for k, cell in enumerate(cells, 1):
            out[f"{pre}_wih{k}"], out[f"{pre}_whh{k}"] = cell.weight_ih.data.clone(), cell.weight_hh.data.clone()
            out[f"{pre}_bih{k}"], out[f"{pre}_bhh{k}"] = cell.bias_ih.data.clone(), cell.bias_hh.data.clone()
        out[f"{pre}_wout"], out[f"{pre}_bout"] = lin.weight.data.clone(), lin.bias.data.clone()
    out["stds"] = init_std * torch.ones(act_dim)
    return out
