"""Host handle over the on-device PPO kernels (include/lhw.h, `lhw_ppo_*`, `lhw_gae`).

torch owns the flat parameter / gradient / Adam-state tensors ("PyTorch-ROCm for parameter
storage"); every arithmetic step of the learner runs in liblhw.so.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib

_SETUP = False


class LhwPpoConfig(ctypes.Structure):
    _fields_ = [
        ("device", ctypes.c_int32), ("obs_dim", ctypes.c_int32), ("act_dim", ctypes.c_int32), ("hidden", ctypes.c_int32),
        ("learn_std", ctypes.c_int32), ("max_rows", ctypes.c_int32), ("lr", ctypes.c_float), ("eps", ctypes.c_float),
        ("clip", ctypes.c_float), ("entropy_coeff", ctypes.c_float), ("mirror_coeff", ctypes.c_float),
        ("max_grad_norm", ctypes.c_float), ("mirror_obs_src", ctypes.c_void_p), ("mirror_obs_sign", ctypes.c_void_p),
        ("mirror_act_src", ctypes.c_void_p), ("mirror_act_sign", ctypes.c_void_p),
    ]


def _setup(L):
    global _SETUP
    if _SETUP:
        return
    vp, i32, i64, f32, u32, u64 = (ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_uint32,
                                   ctypes.c_uint64)
    L.lhw_ppo_create.argtypes = [ctypes.POINTER(LhwPpoConfig), ctypes.POINTER(vp)]
    L.lhw_ppo_destroy.argtypes = [vp]
    L.lhw_ppo_param_count.argtypes = [vp]
    L.lhw_ppo_param_count.restype = i64
    L.lhw_ppo_layout.argtypes = [vp, ctypes.POINTER(i64)]
    L.lhw_ppo_normalize.argtypes = [vp, vp, i64, vp, vp, vp, vp, vp]
    L.lhw_ppo_forward.argtypes = [vp, vp, vp, i64, vp, vp, u64, u32, u32, ctypes.c_int, vp, vp, vp, vp, vp]
    L.lhw_gae.argtypes = [i32, i32, vp, vp, vp, vp, vp, ctypes.c_double, ctypes.c_double, vp, vp, vp]
    L.lhw_moments.argtypes = [vp, i64, vp, vp]
    L.lhw_scale_shift.argtypes = [vp, i64, f32, f32, vp]
    L.lhw_standardize.argtypes = [vp, i64, vp, ctypes.c_double, vp]
    L.lhw_ppo_grad.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp]
    L.lhw_ppo_apply.argtypes = [vp, vp, vp, vp, vp, i64, f32, vp]
    L.lhw_ppo_set_inference_dtype.argtypes = [vp, ctypes.c_int]
    L.lhw_ppo_set_update_dtype.argtypes = [vp, ctypes.c_int]
    L.lhw_ppo_forward_at.argtypes = [vp, vp, vp, i64, vp, vp, u64, u32, u32, ctypes.c_int, i64, vp, vp, vp, vp, vp]
    L.lhw_ppo_begin_rollout.argtypes = [vp, vp, vp]
    L.lhw_ppo_end_rollout.argtypes = [vp]
    _SETUP = True


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class PpoKernels:
    """Flat-parameter actor/critic (D -> H -> H -> A | 1, ReLU) living on one GPU."""

    TENSORS = ["a_w1", "a_b1", "a_w2", "a_b2", "a_w3", "a_b3", "stds", "c_w1", "c_b1", "c_w2", "c_b2", "c_w3", "c_b3"]

    def __init__(self, obs_dim, act_dim, *, hidden=256, max_rows=4096, device=0, learn_std=False, lr=3e-4, eps=1e-5,
                 clip=0.2, entropy_coeff=0.0, mirror_coeff=0.4, max_grad_norm=0.5, mirror_obs=None, mirror_act=None):
        if not torch.cuda.is_available():
            raise _lib.LhwError(-5, "no GPU visible: the PPO kernels have no CPU fallback")
        self.device = torch.device("cuda", device) if isinstance(device, int) else device
        L = _lib.lib()
        _setup(L)
        self._L = L
        cfg = LhwPpoConfig()
        cfg.device = self.device.index or 0
        cfg.obs_dim, cfg.act_dim, cfg.hidden = obs_dim, act_dim, hidden
        cfg.learn_std, cfg.max_rows = int(learn_std), int(max_rows)
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
        _lib.check(L.lhw_ppo_create(ctypes.byref(cfg), ctypes.byref(self._h)))
        self.obs_dim, self.act_dim, self.hidden, self.max_rows, self.learn_std = obs_dim, act_dim, hidden, max_rows, learn_std
        self.eps = eps
        self.n_params = int(L.lhw_ppo_param_count(self._h))
        lay = (ctypes.c_int64 * 15)()
        _lib.check(L.lhw_ppo_layout(self._h, lay))
        self.offsets = list(lay)[:13]
        self.Dp, self.Op = int(lay[13]), int(lay[14])
        dev = self.device
        self.theta = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        self.grad = torch.zeros_like(self.theta)
        self.adam_m = torch.zeros_like(self.theta)
        self.adam_v = torch.zeros_like(self.theta)
        self.adam_step = 0
        self.stats = torch.zeros(16, dtype=torch.float32, device=dev)
        self._mom = torch.zeros(2, dtype=torch.float64, device=dev)
        self.obs_mean = torch.zeros(obs_dim, dtype=torch.float32, device=dev)
        self.obs_std = torch.ones(obs_dim, dtype=torch.float32, device=dev)
        H, D, A, Dp, Op = hidden, obs_dim, act_dim, self.Dp, self.Op
        self.shapes = dict(a_w1=(H, Dp, D), a_b1=(H,), a_w2=(H, H, H), a_b2=(H,), a_w3=(Op, H, H), a_b3=(Op,), stds=(A,),
                           c_w1=(H, Dp, D), c_b1=(H,), c_w2=(H, H, H), c_b2=(H,), c_w3=(4, H, H), c_b3=(4,))
        self.true_rows = dict(a_w3=A, a_b3=A, c_w3=1, c_b3=1)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.lhw_ppo_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- parameter views in torch layouts (for init / checkpoints / parity tests)
    def _view(self, flat, name):
        off = self.offsets[self.TENSORS.index(name)]
        shp = self.shapes[name]
        if len(shp) == 3:
            rows, ld, cols = shp
            v = flat[off:off + rows * ld].view(rows, ld)[:, :cols]
        else:
            v = flat[off:off + shp[0]]
        rows = self.true_rows.get(name)
        return v[:rows] if rows is not None else v

    def get_tensors(self, flat=None):
        flat = self.theta if flat is None else flat
        return {n: self._view(flat, n).detach().cpu().clone() for n in self.TENSORS}

    def set_tensors(self, tensors: dict):
        # theta changes: a rollout bracket opened on the old weights (its [in][out] copies, the resident rollout's actor view) is void
        self.end_rollout()
        for n, t in tensors.items():
            self._view(self.theta, n).copy_(torch.as_tensor(t, dtype=torch.float32).reshape(self._view(self.theta, n).shape))

    def set_inference_fp16(self, on=True):
        """Rollout inference (``forward``) with fp16 operands on the fp16 MFMA; the update stays float32."""
        _lib.check(self._L.lhw_ppo_set_inference_dtype(self._h, int(bool(on))))
        self.inference_fp16 = bool(on)

    def set_update_fp16(self, on=True):
        """Every GEMM of the update (``grad_minibatch``) with fp16 operands on the fp16 MFMA, float32 accumulation; master
        weights, loss and Adam stay float32 (BASELINE config 5, together with ``set_inference_fp16``)."""
        _lib.check(self._L.lhw_ppo_set_update_dtype(self._h, int(bool(on))))
        self.update_fp16 = bool(on)

    def set_obs_norm(self, mean, std):
        self.obs_mean.copy_(torch.as_tensor(np.asarray(mean), dtype=torch.float32))
        self.obs_std.copy_(torch.as_tensor(np.asarray(std), dtype=torch.float32))

    # ---- kernels
    def forward(self, obs, *, seed=0, env_id_base=0, counter=0, deterministic=False, want_actor=True, want_value=True,
                mu=None, act=None, logp=None, value=None, ws_row=0, want_mu=True):
        N = obs.shape[0]
        dev = self.device
        if want_actor:
            if want_mu:      # (the rollout needs only the sampled action and its log-density: one strided copy less per step)
                mu = _lib.empty(N, self.act_dim, dtype=torch.float32, device=dev) if mu is None else mu
            else:
                mu = None
            act = _lib.empty(N, self.act_dim, dtype=torch.float32, device=dev) if act is None else act
            logp = _lib.empty(N, dtype=torch.float32, device=dev) if logp is None else logp
        if want_value:
            value = _lib.empty(N, dtype=torch.float32, device=dev) if value is None else value
        _lib.check(self._L.lhw_ppo_forward_at(self._h, _p(self.theta), _p(obs), N, _p(self.obs_mean), _p(self.obs_std),
                                              int(seed) & (2**64 - 1), int(env_id_base), int(counter), int(deterministic), int(ws_row),
                                              _p(mu) if (want_actor and mu is not None) else None, _p(act) if want_actor else None,
                                              _p(logp) if want_actor else None, _p(value) if want_value else None,
                                              self._stream()))
        return mu, act, logp, value

    def begin_rollout(self):
        """theta is frozen until end_rollout(): the strip kernel's weight copies are made once (on the current stream) instead of
        in every policy step (lhw_ppo_begin_rollout)."""
        _lib.check(self._L.lhw_ppo_begin_rollout(self._h, _p(self.theta), self._stream()))

    def end_rollout(self):
        _lib.check(self._L.lhw_ppo_end_rollout(self._h))

    def rollout_policy(self, *, seed=0, counter=0, deterministic=False):
        """The frozen actor as the resident rollout (BatchedEnv.rollout -> lhw_env_rollout) evaluates it inside the stepper's
        wavefronts; valid inside a begin_rollout() bracket.  Returns None where the in-wave policy step does not apply (shapes outside
        the strip kernels): the caller keeps the launch-per-step pipeline.  With fp16 inference the view carries fp16_operands = 1: the
        in-wave policy step then rounds its operands to fp16 like the launch-per-step fp16 GEMMs do, but sums in another order -- that
        resident rollout is float32-rounding-close (~1e-6) to the launch-per-step one, not bitwise (the float32 policy step is bitwise)."""
        view = _lib.LhwRolloutPolicy()
        rc = self._L.lhw_ppo_rollout_policy(self._h, _p(self.theta), _p(self.obs_mean), _p(self.obs_std), int(seed) & (2**64 - 1),
                                            int(counter) & 0xFFFFFFFF, int(bool(deterministic)), ctypes.byref(view))
        if rc == -4:      # LHW_ERR_UNSUPPORTED
            return None
        _lib.check(rc)
        return view

    def normalize(self, obs, want_mirror=None):
        R = obs.shape[0]
        want_mirror = self.use_mirror if want_mirror is None else want_mirror
        xn = _lib.empty(R, self.Dp, dtype=torch.float32, device=self.device)
        xm = _lib.empty(R, self.Dp, dtype=torch.float32, device=self.device) if want_mirror else None
        _lib.check(self._L.lhw_ppo_normalize(self._h, _p(obs), R, _p(self.obs_mean), _p(self.obs_std), _p(xn), _p(xm),
                                             self._stream()))
        return xn, xm

    def gae(self, rew, val, done, vterm, vfinal, gamma, lam):
        T, N = rew.shape
        ret = _lib.empty(T, N, dtype=torch.float32, device=self.device)
        adv = _lib.empty(T, N, dtype=torch.float32, device=self.device)
        _lib.check(self._L.lhw_gae(T, N, _p(rew), _p(val), _p(done), _p(vterm), _p(vfinal), float(gamma), float(lam),
                                   _p(ret), _p(adv), self._stream()))
        return ret, adv

    def moments(self, x):
        _lib.check(self._L.lhw_moments(_p(x), x.numel(), _p(self._mom), self._stream()))
        return self._mom

    def standardize(self, x, stats3, eps):
        """x <- (x - mean) / (std + eps) from the device-resident {sum, sum of squares, count} (no host round trip)."""
        assert stats3.dtype == torch.float64 and stats3.numel() == 3 and stats3.is_cuda
        _lib.check(self._L.lhw_standardize(_p(x), x.numel(), _p(stats3), float(eps), self._stream()))

    def scale_shift(self, x, mean, inv):
        _lib.check(self._L.lhw_scale_shift(_p(x), x.numel(), float(mean), float(inv), self._stream()))

    def grad_minibatch(self, xn, xm, act, old_logp, adv, ret, idx, imitation=None):
        """``imitation`` = (coeff, target [B, A] f32, mask [B, A] u8, n_selected): the imitation term of this minibatch
        (rows in ``idx`` order), see lhw_ppo_set_imitation."""
        B = idx.numel()
        if imitation is not None:
            coeff, target, mask, count = imitation
            assert target.shape == (B, self.act_dim) and mask.shape == (B, self.act_dim) and mask.dtype == torch.uint8
            self._imit_keep = (target.contiguous(), mask.contiguous())      # must outlive the asynchronous kernel
            _lib.check(self._L.lhw_ppo_set_imitation(self._h, _p(self._imit_keep[0]), _p(self._imit_keep[1]), float(coeff), int(count)))
        _lib.check(self._L.lhw_ppo_grad(self._h, _p(self.theta), _p(self.grad), _p(xn), _p(xm), _p(act), _p(old_logp),
                                        _p(adv), _p(ret), _p(idx), B, _p(self.stats), self._stream()))

    def step_minibatch(self, xn, xm, act, old_logp, adv, ret, idx):
        """grad_minibatch + apply as ONE hipGraph launch (lhw_ppo_step: single process, no imitation term).  Must run on a torch stream
        other than the default one (a legacy default stream cannot be captured: the library then makes the two calls eagerly)."""
        self.adam_step += 1
        _lib.check(self._L.lhw_ppo_step(self._h, _p(self.theta), _p(self.grad), _p(self.adam_m), _p(self.adam_v), _p(xn), _p(xm), _p(act),
                                        _p(old_logp), _p(adv), _p(ret), _p(idx), idx.numel(), _p(self.stats), self.adam_step, 1.0, self._stream()))

    def apply(self, grad_scale=1.0):
        self.adam_step += 1
        _lib.check(self._L.lhw_ppo_apply(self._h, _p(self.theta), _p(self.grad), _p(self.adam_m), _p(self.adam_v),
                                         self.adam_step, float(grad_scale), self._stream()))


def reference_init(obs_dim, act_dim, hidden=256, init_std=0.223, generator_seed=None):
    """Initial weights with exactly the RNG consumption of the reference's constructors
    (reference rl/policies/actor.py:122-158, critic.py:15-39, base.py:5-22): three nn.Linear per
    net (default init draws), then normc (weight ~ N(0,1), rows scaled to unit L2 norm, bias 0),
    actor mean layer x0.01 -- so that the same torch seed gives the reference's initial weights."""
    import torch.nn as nn

    # the draws come from the CPU default generator, seeded here and restored afterwards; the CUDA generators are left alone
    # (torch.manual_seed would reseed them too)
    with torch.random.fork_rng(devices=[]):
        if generator_seed is not None:
            torch.default_generator.manual_seed(int(generator_seed))
        return _reference_init_draw(obs_dim, act_dim, hidden, init_std)


def _reference_init_draw(obs_dim, act_dim, hidden, init_std):
    import torch.nn as nn

    def net(out_dim, scale_out):
        layers = [nn.Linear(obs_dim, hidden), nn.Linear(hidden, hidden), nn.Linear(hidden, out_dim)]
        for l in layers:
            l.weight.data.normal_(0, 1)
            l.weight.data *= 1 / torch.sqrt(l.weight.data.pow(2).sum(1, keepdim=True))
            l.bias.data.fill_(0)
        if scale_out is not None:
            layers[2].weight.data.mul_(scale_out)
        return layers

    a = net(act_dim, 0.01)
    c = net(1, None)
    out = {}
    for pre, ls in (("a", a), ("c", c)):
        for k, l in enumerate(ls, 1):
            out[f"{pre}_w{k}"] = l.weight.data.clone()
            out[f"{pre}_b{k}"] = l.bias.data.clone()
    out["stds"] = init_std * torch.ones(act_dim)
    return out
