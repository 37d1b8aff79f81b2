"""
oracle/ct_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes binding of oracle/libct_oracle.so (the scalar C restatement of
CounterTrafficEnv.step, see ct_oracle.c).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libct_oracle.so")

MAX_DEV = 32
MAX_RADIOS = MAX_DEV + 1
QUEUE_CAP = 100

FLAG_CARRY, FLAG_REFEXC, FLAG_TIE = 1, 2, 4


class Config(C.Structure):
    _fields_ = [
        ("num_devices", C.c_int32),
        ("pos", (C.c_double * 2) * MAX_RADIOS),
        ("mult", C.c_int32 * MAX_DEV),
        ("dest", C.c_int32 * MAX_DEV),
        ("slot", C.c_double),
        ("frequency", C.c_double),
        ("bandwidth", C.c_double),
        ("temperature_c", C.c_double),
        ("bit_rate", C.c_double),
        ("code_rate", C.c_double),
        ("max_ber", C.c_double),
        ("tx_power_dbm", C.c_double),
        ("counter_interval", C.c_double),
        ("counter_bound", C.c_int32),
        ("payload_value", C.c_int32),
        ("mac_header_bytes", C.c_int32),
        ("net_header_bytes", C.c_int32),
        ("duration_factor", C.c_int32),
        ("max_duration", C.c_int32),
        ("extra_att_db", (C.c_double * MAX_RADIOS) * MAX_RADIOS),
        ("start_time", C.c_double),
    ]


def build(force=False):
    """Compile the C restatement (building the checker is not using it)."""
    src = os.path.join(_HERE, "ct_oracle.c")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libct_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.cto_config_default.argtypes = [C.POINTER(Config), C.c_int]
        L.cto_config_default.restype = C.c_int
        L.cto_create.argtypes = [C.POINTER(Config), C.c_int64]
        L.cto_create.restype = C.c_void_p
        L.cto_destroy.argtypes = [C.c_void_p]
        L.cto_destroy.restype = None
        L.cto_set_position.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
        L.cto_set_position.restype = C.c_int
        L.cto_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.cto_reset.restype = None
        L.cto_step.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [C.c_int]
        L.cto_step.restype = C.c_int
        L.cto_get.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
        L.cto_get.restype = C.c_int
        for name, args in (("cto_attenuation", [C.c_void_p, C.c_int, C.c_int]),
                           ("cto_rx_power_mw", [C.c_void_p, C.c_int, C.c_int]),
                           ("cto_thermal_mw", [C.c_void_p]),
                           ("cto_ber", [C.c_void_p, C.c_double, C.c_double]),
                           ("cto_data_rate", [C.c_void_p])):
            fn = getattr(L, name)
            fn.argtypes, fn.restype = args, C.c_double
        _lib = L
    return _lib


def default_config(num_devices, positions=None, mult=None, dest=None, rrm_pos=None, extra_att=None, start_time=None):
    cfg = Config()
    if lib().cto_config_default(C.byref(cfg), num_devices) != 0:
        raise ValueError("num_devices out of range")
    D = num_devices
    if positions is not None:
        for i, (x, y) in enumerate(positions):
            cfg.pos[i][0], cfg.pos[i][1] = float(x), float(y)
    if rrm_pos is not None:
        cfg.pos[D][0], cfg.pos[D][1] = float(rrm_pos[0]), float(rrm_pos[1])
    if mult is not None:
        for i, m in enumerate(mult):
            cfg.mult[i] = int(m)
    if dest is not None:
        for i, m in enumerate(dest):
            cfg.dest[i] = int(m)
    if start_time is not None:
        cfg.start_time = float(start_time)
    if extra_att is not None:              # {(a, b): dB}, radio index D = the RRM; applied to both directions
        for (a, b), db in extra_att.items():
            cfg.extra_att_db[a][b] = cfg.extra_att_db[b][a] = float(db)
    return cfg


_FIELDS = {
    # name: (dtype, per-env shape as a function of (D, R))
    "now": (np.float64, lambda D, R: ()),
    "wake": (np.float64, lambda D, R: (D,)),
    "counter": (np.uint32, lambda D, R: (D,)),
    "qlen": (np.int32, lambda D, R: (D,)),
    "queue": (np.uint32, lambda D, R: (D, QUEUE_CAP)),
    "received": (np.int32, lambda D, R: (D,)),
    "latest_diff": (np.int32, lambda D, R: ()),
    "last_abs": (np.int32, lambda D, R: ()),
    "rx_power": (np.float64, lambda D, R: (R,)),
    "flags": (np.uint32, lambda D, R: ()),
    "n_tx": (np.uint64, lambda D, R: ()),
    "n_delivered": (np.uint64, lambda D, R: ()),
    "n_appended": (np.uint64, lambda D, R: ()),
    "n_popped": (np.uint64, lambda D, R: ()),
    "n_dropped": (np.uint64, lambda D, R: ()),
}


class CtOracle:
    """N independent CounterTraffic envs on the CPU (scalar C, f64)."""

    def __init__(self, num_envs, num_devices=2, config=None, nthreads=1):
        self.cfg = config if config is not None else default_config(num_devices)
        self.D = int(self.cfg.num_devices)
        self.R = self.D + 1
        self.n = int(num_envs)
        self.nthreads = int(nthreads)
        self._h = lib().cto_create(C.byref(self.cfg), self.n)
        if not self._h:
            raise MemoryError("cto_create failed")

    def close(self):
        if self._h:
            lib().cto_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_position(self, radio, x, y):
        """Position.set on one radio of every env of this handle (between steps)."""
        rc = lib().cto_set_position(self._h, int(radio), float(x), float(y))
        if rc:
            raise ValueError("cto_set_position: %s" % ("bad radio index" if rc == -1 else
                             "a keep-the-stale-attenuation rule applies to a pair whose model exists in some envs only: one handle per env"))

    def reset(self, mask=None):
        obs = np.empty(self.n, np.int32)
        m = None
        if mask is not None:
            m = np.ascontiguousarray(mask, np.uint8)
            assert m.shape == (self.n,)
        lib().cto_reset(self._h, m.ctypes.data if m is not None else None, obs.ctypes.data)
        return obs

    def step(self, device, duration):
        dev = np.ascontiguousarray(device, np.int32)
        dur = np.ascontiguousarray(duration, np.int32)
        assert dev.shape == (self.n,) and dur.shape == (self.n,)
        obs = np.empty(self.n, np.int32)
        rew = np.empty(self.n, np.float32)
        done = np.empty(self.n, np.uint8)
        rc = lib().cto_step(self._h, dev.ctypes.data, dur.ctypes.data, obs.ctypes.data,
                            rew.ctypes.data, done.ctypes.data, self.nthreads)
        if rc != 0:
            raise AssertionError("invalid action in %d env(s)" % rc)
        return obs, rew, done

    def get(self, field):
        dtype, shp = _FIELDS[field]
        out = np.empty((self.n,) + shp(self.D, self.R), dtype)
        rc = lib().cto_get(self._h, field.encode(), out.ctypes.data, out.nbytes)
        if rc != 0:
            raise KeyError("%s (rc=%d)" % (field, rc))
        return out

    # static tables
    def attenuation(self, a, b):
        return lib().cto_attenuation(self._h, a, b)

    def rx_power_mw(self, frm, to):
        return lib().cto_rx_power_mw(self._h, frm, to)

    def thermal_mw(self):
        return lib().cto_thermal_mw(self._h)

    def ber(self, signal_mw, noise_mw):
        return lib().cto_ber(self._h, signal_mw, noise_mw)

    def data_rate(self):
        return lib().cto_data_rate(self._h)
