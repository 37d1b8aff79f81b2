"""
Host-side mirror of the reference's benchmark fixture (tests/test_benchmark.py:52-91): N independent
replicas of a grid of uncoordinated PHY-only senders, advanced with ``runSimulation(seconds)`` like
``SimMan.runSimulation`` -- on the GPU, one wave per replica (gymwipe_amd/csrc/grid_phy.hip).
"""
import ctypes as C

import numpy as np

from . import _native as nat


class VecPhyGrid:
    SEND_INTERVAL = 1e-2                        # tests/test_benchmark.py:17

    def __init__(self, num_envs, num_devices, initial_delays, device="cuda:0", positions=None, mobile=False, seed=0,
                 tx_power_dbm=None, header_bytes=None, payload_bytes=None, send_interval=None, move_interval=None):
        """``initial_delays``: float64[N][n], the per-device random.uniform(0, SEND_INTERVAL) of the fixture."""
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("gymwipe_amd needs a HIP device; there is no CPU fallback")
        self._torch = torch
        self._L = nat.lib()
        self.num_envs, self.num_devices = int(num_envs), int(num_devices)
        self.device = torch.device(device)
        cfg = nat.GridConfig()
        nat.check(self._L.gw_grid_config_default(C.byref(cfg), self.num_envs, self.num_devices))
        cfg.hip_device = self.device.index or 0
        if positions is not None:
            for i, (x, y) in enumerate(positions):
                cfg.pos[i][0], cfg.pos[i][1] = float(x), float(y)
        cfg.mobile = 1 if mobile else 0           # mobile_device_grid: every device random-walks every 1 ms
        cfg.seed = int(seed)
        if tx_power_dbm is not None:
            cfg.tx_power_dbm = float(tx_power_dbm)
        if header_bytes is not None:
            cfg.header_bytes = int(header_bytes)
        if payload_bytes is not None:
            cfg.payload_bytes = int(payload_bytes)
        if send_interval is not None:
            cfg.send_interval = float(send_interval)
        if move_interval is not None:
            cfg.move_interval = float(move_interval)
        self.config = cfg
        d = np.ascontiguousarray(initial_delays, np.float64)
        assert d.shape == (self.num_envs, self.num_devices)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            nat.check(self._L.gw_grid_create(C.byref(cfg), d.ctypes.data, C.byref(self._h)))

    def runSimulation(self, seconds):           # simtools.py:77-88
        torch = self._torch
        with torch.cuda.device(self.device):
            nat.check(self._L.gw_grid_run(self._h, float(seconds), torch.cuda.current_stream(self.device).cuda_stream))

    def setPosition(self, device, x, y):
        """Position.set(x, y) of one device in every replica, now (devices/core.py:77-86)."""
        xs = np.ascontiguousarray(np.broadcast_to(np.asarray(x, np.float64), (self.num_envs,)))
        ys = np.ascontiguousarray(np.broadcast_to(np.asarray(y, np.float64), (self.num_envs,)))
        torch = self._torch
        with torch.cuda.device(self.device):
            nat.check(self._L.gw_grid_set_position(self._h, int(device), xs.ctypes.data, ys.ctypes.data,
                                                   torch.cuda.current_stream(self.device).cuda_stream))

    def get_state(self, field):
        N, n = self.num_envs, self.num_devices
        shapes = {"now": ((N,), np.float64), "events": ((N,), np.uint32), "n_tx": ((N,), np.uint32), "flags": ((N,), np.uint32),
                  "on_air": ((N,), np.uint32),
                  "rx_power": ((N, n), np.float64), "pos": ((N, n, 2), np.float64)}
        for k in ("n_sent", "hdr_ok", "hdr_fail", "pay_ok", "pay_fail"):
            shapes[k] = ((N, n), np.uint32)
        shape, dtype = shapes[field]
        out = np.empty(shape, dtype)
        nat.check(self._L.gw_grid_get_state(self._h, field.encode(), out.ctypes.data, out.nbytes))
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.gw_grid_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
