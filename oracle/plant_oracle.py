"""
oracle/plant_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
ctypes binding of oracle/libplant_oracle.so (scalar restatement of the linear plant recurrence).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libplant_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "plant_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libplant_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


class PlantOracle:
    def __init__(self, num_envs, A, B, dt, x0, u0):
        build()
        self._L = C.CDLL(_LIB)
        self._L.plant_oracle_update_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int64] + [C.c_void_p] * 5
        self._L.plant_oracle_update_batch.restype = None
        self.n = int(num_envs)
        self.A = np.ascontiguousarray(A, np.float64).reshape(16)
        self.B = np.ascontiguousarray(B, np.float64).reshape(4)
        self.dt = float(dt)
        self.x = np.tile(np.asarray(x0, np.float64), (self.n, 1))
        self.u = np.full(self.n, float(u0))
        self.t_last = np.zeros(self.n)
        self.substeps = np.zeros(self.n, np.uint64)

    def update(self, now):
        now = np.ascontiguousarray(now, np.float64)
        self._L.plant_oracle_update_batch(self.A.ctypes.data, self.B.ctypes.data, self.dt, self.n, self.x.ctypes.data,
                                          self.u.ctypes.data, self.t_last.ctypes.data, now.ctypes.data,
                                          self.substeps.ctypes.data)

    def set_input(self, u, mask=None):
        u = np.asarray(u, np.float64)
        if mask is None:
            self.u[:] = u
        else:
            m = np.asarray(mask, bool)
            self.u[m] = u[m]
