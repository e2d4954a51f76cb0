"""ctypes view of the float64 C oracle (oracle/mjc_oracle.c).  TEST INFRASTRUCTURE ONLY.

`OracleSim` plays the role of the reference's (mjModel, mjData) pair for the oracle env logic
in oracle/env_*.py: the attributes mirror the mjData fields the reference reads through
RobotInterface (reference envs/common/robot_interface.py:60-546).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libmjc_oracle.so")
    src = os.path.join(_HERE, "mjc_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "lhw_model_fields.h")
    if force or not os.path.exists(so) or (
        os.path.exists(src) and os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libmjc_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.orc_new.restype = ctypes.c_void_p
        L.orc_new.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.orc_free.argtypes = [ctypes.c_void_p]
        L.orc_reset_data.argtypes = [ctypes.c_void_p]
        L.orc_forward.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_step.argtypes = [ctypes.c_void_p]
        L.orc_field.restype = ctypes.POINTER(ctypes.c_double)
        L.orc_field.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        for f in ("orc_ncon", "orc_nefc", "orc_niter"):
            getattr(L, f).argtypes = [ctypes.c_void_p]
            getattr(L, f).restype = ctypes.c_int
        L.orc_time.argtypes = [ctypes.c_void_p]
        L.orc_time.restype = ctypes.c_double
        L.orc_efc_type.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_efc_type.restype = ctypes.c_int
        L.orc_dual_pgs.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                   ctypes.c_int, ctypes.c_double]
        L.orc_dual_pgs.restype = ctypes.c_int
        L.orc_contact.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.orc_contact_force.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.orc_object_velocity.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        _LIB = L
    return _LIB


class OracleSim:
    """One model + one data instance."""

    def __init__(self, model):
        self.model = model
        self._ib, self._db = model.pack()
        self._L = lib()
        self._d = self._L.orc_new(self._ib.ctypes.data, self._db.ctypes.data)
        if not self._d:
            raise RuntimeError("orc_new failed (bad model blob)")
        m = model
        self._shapes = dict(
            qpos=(m.nq,), qvel=(m.nv,), ctrl=(m.nu,), xfrc_applied=(m.nbody, 6), qacc_warmstart=(m.nv,),
            xpos=(m.nbody, 3), xquat=(m.nbody, 4), xmat=(m.nbody, 9), xipos=(m.nbody, 3), ximat=(m.nbody, 9), geom_xpos=(m.ngeom, 3),
            geom_xmat=(m.ngeom, 9), site_xpos=(m.nsite, 3), site_xmat=(m.nsite, 9), subtree_com=(m.nbody, 3),
            cvel=(m.nbody, 6), M=(m.nv, m.nv), qfrc_bias=(m.nv,), qfrc_passive=(m.nv,), qfrc_actuator=(m.nv,),
            qfrc_smooth=(m.nv,), qacc_smooth=(m.nv,), qfrc_constraint=(m.nv,), qacc=(m.nv,),
            actuator_length=(m.nu,), actuator_velocity=(m.nu,), actuator_force=(m.nu,),
        )
        self._views = {}

    def __del__(self):
        try:
            self._L.orc_free(self._d)
        except Exception:
            pass

    def repack(self):
        """Push edits made to ``self.model`` arrays (dyn-rand, timestep) into the C side in place."""
        ib, db = self.model.pack()
        assert ib.shape == self._ib.shape and db.shape == self._db.shape
        self._ib[:] = ib
        self._db[:] = db

    def __getattr__(self, name):
        shapes = self.__dict__.get("_shapes", {})
        if name in shapes:
            v = self._views.get(name)
            if v is None:
                n = int(np.prod(shapes[name]))
                if n == 0:
                    v = np.zeros(shapes[name])
                else:
                    p = self._L.orc_field(self._d, name.encode())
                    v = np.ctypeslib.as_array(p, shape=(n,)).reshape(shapes[name])
                self._views[name] = v
            return v
        raise AttributeError(name)

    def efc(self, name):
        n = self.nefc
        p = self._L.orc_field(self._d, name.encode())
        width = self.model.nv if name == "efc_J" else 1
        a = np.ctypeslib.as_array(p, shape=(n * width,)).copy()
        return a.reshape(n, width) if width > 1 else a

    def efc_types(self):
        """Row kinds in efc order: 0 frictionloss, 1 joint limit, 2 contact (pyramid edge)."""
        return np.array([self._L.orc_efc_type(self._d, r) for r in range(self.nefc)], dtype=np.int32)

    @property
    def ncon(self):
        return self._L.orc_ncon(self._d)

    @property
    def nefc(self):
        return self._L.orc_nefc(self._d)

    @property
    def niter(self):
        return self._L.orc_niter(self._d)

    @property
    def time(self):
        return self._L.orc_time(self._d)

    def reset_data(self):
        self._L.orc_reset_data(self._d)

    def forward(self, actuation=True):
        self._L.orc_forward(self._d, int(actuation))

    def step(self, n=1):
        for _ in range(n):
            self._L.orc_step(self._d)

    def contact(self, i):
        out = np.zeros(17)
        self._L.orc_contact(self._d, i, out.ctypes.data)
        return dict(dist=out[0], pos=out[1:4].copy(), frame=out[4:13].reshape(3, 3).copy(), geom1=int(out[13]),
                    geom2=int(out[14]), efc_address=int(out[15]), mu=out[16])

    def contact_force(self, i):
        out = np.zeros(6)
        self._L.orc_contact_force(self._d, i, out.ctypes.data)
        return out

    def object_velocity(self, body, local):
        out = np.zeros(6)
        self._L.orc_object_velocity(self._d, int(body), int(local), out.ctypes.data)
        return out
