"""
GPU tier: what INTEGRATION.md ships to a reference maintainer -- `gymwipe_amd/lib/libgymwipe_amd.so` (code objects for ANY
XNACK setting) driven through PLAIN ctypes -- against the oracle.  The package's own loader prefers the xnack- build and the
CPython fast-call shim on this pool, so without these tests the portable library's kernels (other register allocation around
loads: XNACK replay constraints) and the plain `gw_step` ctypes path would never run on a GPU.

  * in process: the library is loaded a second time under its own handle and driven by the few ctypes lines INTEGRATION.md
    shows ("The binding a reference maintainer would add"): reference known answer, D = 4 with resets, the D = 16 multi-word
    records, a 64-step fused rollout -- outputs of every step and the final state, bit for bit;
  * in a fresh child process with GW_NO_XNACKOFF=1 GW_NO_PYFAST=1: the package itself on that library and on ctypes' gw_step.
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from util import action_stream

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PORTABLE = os.path.join(ROOT, "gymwipe_amd", "lib", "libgymwipe_amd.so")


class PlainBinding:
    """INTEGRATION.md's ctypes binding, written out: nothing of gymwipe_amd's Python is on the call path (the gw_config
    struct mirror is reused from gymwipe_amd._native, which is declarations only)."""

    def __init__(self, num_envs, num_devices):
        import torch
        from gymwipe_amd import _native as nat
        L = self.L = C.CDLL(PORTABLE, mode=C.RTLD_LOCAL)
        vp = C.c_void_p
        L.gw_last_error.restype = C.c_char_p
        L.gw_config_default.argtypes = [C.POINTER(nat.Config), C.c_int64, C.c_int32]
        L.gw_create.argtypes = [C.POINTER(nat.Config), C.POINTER(vp)]
        L.gw_destroy.argtypes = [vp]
        L.gw_step.argtypes = [vp] * 7
        L.gw_reset.argtypes = [vp] * 4
        L.gw_rollout.argtypes = [vp, C.c_int32] + [vp] * 6
        L.gw_get_state.argtypes = [vp, C.c_char_p, vp, C.c_size_t]
        cfg = nat.Config()
        assert L.gw_config_default(C.byref(cfg), num_envs, num_devices) == 0
        cfg.hip_device = 0
        self.h = vp()
        assert L.gw_create(C.byref(cfg), C.byref(self.h)) == 0, L.gw_last_error()
        self.N, self.D = num_envs, num_devices
        self.obs = torch.empty(num_envs, dtype=torch.int32, device="cuda:0")
        self.rew = torch.empty(num_envs, dtype=torch.float32, device="cuda:0")
        self.done = torch.empty(num_envs, dtype=torch.uint8, device="cuda:0")
        self.torch = torch

    def _stream(self):
        return self.torch.cuda.current_stream().cuda_stream

    def reset(self):
        assert self.L.gw_reset(self.h, None, self.obs.data_ptr(), self._stream()) == 0, self.L.gw_last_error()
        return self.obs.cpu().numpy()

    def step(self, dev, dur):
        t = self.torch
        a, b = t.from_numpy(dev).cuda(), t.from_numpy(dur).cuda()
        rc = self.L.gw_step(self.h, a.data_ptr(), b.data_ptr(), self.obs.data_ptr(), self.rew.data_ptr(), self.done.data_ptr(),
                            self._stream())
        assert rc == 0, self.L.gw_last_error()
        return self.obs.cpu().numpy(), self.rew.cpu().numpy(), self.done.cpu().numpy()

    def rollout(self, dev, dur):
        t = self.torch
        K = dev.shape[0]
        a, b = t.from_numpy(dev).cuda().contiguous(), t.from_numpy(dur).cuda().contiguous()
        o = t.empty((K, self.N), dtype=t.int32, device="cuda:0")
        r = t.empty((K, self.N), dtype=t.float32, device="cuda:0")
        d = t.empty((K, self.N), dtype=t.uint8, device="cuda:0")
        rc = self.L.gw_rollout(self.h, K, a.data_ptr(), b.data_ptr(), o.data_ptr(), r.data_ptr(), d.data_ptr(), self._stream())
        assert rc == 0, self.L.gw_last_error()
        return o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy()

    def get(self, field, dtype, shape):
        out = np.empty((self.N,) + shape, dtype)
        assert self.L.gw_get_state(self.h, field.encode(), out.ctypes.data, out.nbytes) == 0, self.L.gw_last_error()
        return out

    def close(self):
        self.L.gw_destroy(self.h)


def _state_equal(b, orc, where):
    D, R = b.D, b.D + 1
    for f, dt, shp in (("now", np.float64, ()), ("wake", np.float64, (D,)), ("counter", np.uint32, (D,)), ("qlen", np.int32, (D,)),
                       ("queue", np.uint32, (D, 100)), ("received", np.int32, (D,)), ("last_abs", np.int32, ()),
                       ("rx_power", np.float64, (R,)), ("flags", np.uint32, ()), ("n_tx", np.uint64, ()), ("n_popped", np.uint64, ())):
        x, y = b.get(f, dt, shp), orc.get(f)
        assert (x.view(np.uint8) == y.view(np.uint8)).all(), "state field %r differs %s" % (f, where)


def _loaded_libraries():
    with open("/proc/self/maps") as fh:
        return {line.split()[-1] for line in fh if "libgymwipe_amd" in line}


def test_portable_library_through_plain_ctypes_known_answer():
    """tests/envs/test_counter_traffic.py:25-34 of the reference on libgymwipe_amd.so: +2 / -2, then 0 / +2."""
    b = PlainBinding(1, 2)
    assert PORTABLE in _loaded_libraries()
    o, r, d = b.step(np.array([0], np.int32), np.array([3], np.int32))
    assert o[0] - 65536 == 2 and r[0] == -2.0 and d[0] == 0
    o, r, d = b.step(np.array([1], np.int32), np.array([12], np.int32))
    assert o[0] - 65536 == 0 and r[0] == 2.0
    assert b.get("now", np.float64, ())[0] == 0.017804000036000002
    b.close()


@pytest.mark.parametrize("D,N,K", [(4, 4096, 96), (16, 1024, 64)])
def test_portable_library_parity_with_resets(D, N, K):
    from oracle.ct_oracle import CtOracle
    b, orc = PlainBinding(N, D), CtOracle(N, D, nthreads=8)
    dev, dur = action_stream(100 + D, K, N, D)
    assert (b.reset() == orc.reset()).all()
    for k in range(K):
        if k and k % 32 == 0:
            assert (b.reset() == orc.reset()).all()
        o, r, d = b.step(dev[k], dur[k])
        oo, orr, od = orc.step(dev[k], dur[k])
        assert (o == oo).all() and (r == orr).all() and (d == od).all(), "outputs differ at step %d" % k
        if (k + 1) % 16 == 0:
            _state_equal(b, orc, "after step %d" % k)
    assert int(orc.get("flags").max()) & 3 == 0
    b.close()


@pytest.mark.parametrize("D", [4, 16])
def test_portable_library_fused_rollout(D, monkeypatch):
    from oracle.ct_oracle import CtOracle
    monkeypatch.setenv("GW_ROLLOUT_STRICT", "1")          # fail rather than fall back to step launches
    N, K = 2048, 64
    b, orc = PlainBinding(N, D), CtOracle(N, D, nthreads=8)
    dev, dur = action_stream(777 + D, K, N, D)
    assert (b.reset() == orc.reset()).all()
    o, r, d = b.rollout(dev, dur)
    for k in range(K):
        oo, orr, od = orc.step(dev[k], dur[k])
        assert (o[k] == oo).all() and (r[k] == orr).all() and (d[k] == od).all(), "outputs differ at step %d" % k
    _state_equal(b, orc, "after the rollout")
    b.close()


CHILD = r"""
import json, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch
import gymwipe_amd
from gymwipe_amd import _native as nat
from oracle.ct_oracle import CtOracle
from util import action_stream, assert_state_equal, STATE_FIELDS, STAT_FIELDS
assert nat.fast() is None, "GW_NO_PYFAST ignored"
assert os.path.basename(nat.lib()._name) == "libgymwipe_amd.so", nat.lib()._name
done = []
# the reference's known answer through the drop-in env
env = gymwipe_amd.make("CounterTraffic-v0")
o, r, _, info = env.step({"device": 0, "duration": 3}); assert o - env.COUNTER_BOUND == 2 and r == -2
o, r, _, info = env.step({"device": 1, "duration": 12}); assert o - env.COUNTER_BOUND == 0 and r == 2
assert info == {"Latest received values": "[2, 2]"} and env.get_state("now")[0] == 0.017804000036000002
done.append("known answer")
for D, N, K in ((4, 4096, 96), (16, 1024, 64)):
    env = gymwipe_amd.VecCounterTrafficEnv(N, num_devices=D, per_env_stats=True)
    orc = CtOracle(N, D, nthreads=8)
    dev, dur = action_stream(100 + D, K, N, D)
    assert (env.reset().cpu().numpy() == orc.reset()).all()
    for k in range(K):
        if k and k %% 32 == 0:
            assert (env.reset().cpu().numpy() == orc.reset()).all()
        o, r, d, _ = env.step({"device": torch.from_numpy(dev[k]), "duration": torch.from_numpy(dur[k])})
        oo, orr, od = orc.step(dev[k], dur[k])
        assert (o.cpu().numpy() == oo).all() and (r.cpu().numpy() == orr).all() and (d.cpu().numpy() == od).all(), (D, k)
        if (k + 1) %% 16 == 0:
            assert_state_equal(env, orc, STATE_FIELDS + STAT_FIELDS, where="D=%%d after step %%d" %% (D, k))
    env.check()
    done.append("parity with resets D=%%d" %% D)
os.environ["GW_ROLLOUT_STRICT"] = "1"
N, D, K = 2048, 4, 64
env = gymwipe_amd.VecCounterTrafficEnv(N, num_devices=D, per_env_stats=True); orc = CtOracle(N, D, nthreads=8)
dev, dur = action_stream(4242, K, N, D)
assert (env.reset().cpu().numpy() == orc.reset()).all()
obs, rew, dn = env.rollout(torch.from_numpy(dev), torch.from_numpy(dur))
obs, rew, dn = obs.cpu().numpy(), rew.cpu().numpy(), dn.cpu().numpy()
for k in range(K):
    oo, orr, od = orc.step(dev[k], dur[k])
    assert (obs[k] == oo).all() and (rew[k] == orr).all() and (dn[k] == od).all(), k
assert_state_equal(env, orc, STATE_FIELDS + STAT_FIELDS, where="after the rollout")
done.append("fused rollout")
maps = sorted({l.split()[-1] for l in open("/proc/self/maps") if "gymwipe_amd/lib" in l})
print(json.dumps({"done": done, "loaded": [os.path.basename(m) for m in maps]}))
"""


def test_package_on_the_portable_library_and_plain_ctypes_in_a_fresh_process():
    env = dict(os.environ, GW_NO_XNACKOFF="1", GW_NO_PYFAST="1")
    env.pop("GW_LIB", None)
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["done"] == ["known answer", "parity with resets D=4", "parity with resets D=16", "fused rollout"]
    assert d["loaded"] == ["libgymwipe_amd.so"], d["loaded"]        # neither the xnack- build nor the fast-call shim
