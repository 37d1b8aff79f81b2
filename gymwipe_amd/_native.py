"""
ctypes binding of the C-ABI (include/gymwipe_amd.h).  Loading fails loudly when the
HIP library has not been built: there is no Python/CPU fallback for the compute path.
"""
import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GW_LIB") or os.path.join(_PKG, "lib", "libgymwipe_amd.so")   # GW_LIB: diagnostic builds
CSRC = os.path.join(_PKG, "csrc")

ABI_VERSION = 1
MAX_DEVICES = 32
MAX_RADIOS = MAX_DEVICES + 1
QUEUE_CAP = 100
MAX_NSTATES = 16

OK, EINVAL, ENODEVICE, EHIP, ENOMEM, EUNSUPPORTED, EFIELD = 0, -1, -2, -3, -4, -5, -6
FLAG_CARRY, FLAG_REFEXC, FLAG_TIE, FLAG_BADACT, FLAG_INTERNAL = 1, 2, 4, 8, 16
CFG_PER_ENV_STATS = 1
CFG_EXPLICIT_QUEUE = 2
CFG_NO_COUNTER_TRAFFIC = 4
CFG_PEER_RECEIVE = 8
CFG_FLOAT_DURATION = 16
CFG_PER_ENV_GEOMETRY = 32

EXPORTS = (
    "gw_abi_version", "gw_last_error", "gw_device_count", "gw_config_default", "gw_create",
    "gw_destroy", "gw_reset", "gw_step", "gw_step_fb", "gw_rollout", "gw_set_position", "gw_set_positions", "gw_received", "gw_delivered", "gw_enqueue", "gw_pack_feedback", "gw_unpack_feedback", "gw_get_state",
    "gw_stats_read", "gw_clear_flags", "gw_state_bytes", "gw_snapshot_bytes", "gw_get_snapshot", "gw_set_state", "gw_link_info", "gw_noise_states", "gw_selftest_queue", "gw_selftest_runq",
    "gw_selftest_fastmath",
    "gw_plant_config_default", "gw_plant_create", "gw_plant_destroy", "gw_plant_update", "gw_plant_set_input",
    "gw_plant_state_ptr", "gw_plant_get_state", "gw_plant_feedback", "gw_plant_update_feedback", "gw_now_ptr", "gw_pendulum_step",
    "gw_ctrl_config_default", "gw_ctrl_create", "gw_ctrl_destroy", "gw_ctrl_step", "gw_ctrl_get_state",
    "gw_grid_config_default", "gw_grid_create", "gw_grid_destroy", "gw_grid_run", "gw_grid_get_state", "gw_grid_set_position",
)


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("hip_device", C.c_int32),
        ("num_envs", C.c_int64),
        ("num_devices", C.c_int32),
        ("flags", C.c_int32),
        ("pos", (C.c_double * 2) * MAX_RADIOS),
        ("mult", C.c_int32 * MAX_DEVICES),
        ("dest", C.c_int32 * MAX_DEVICES),
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


class CtrlConfig(C.Structure):
    _fields_ = [
        ("net", Config),
        ("A", C.c_double * 16), ("B", C.c_double * 4), ("x0", C.c_double * 4), ("u0", C.c_double),
        ("ctrl_start_tick", C.c_int32), ("ctrl_period_ticks", C.c_int32),
    ]


class PlantConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("hip_device", C.c_int32),
        ("num_envs", C.c_int64),
        ("A", C.c_double * 16),
        ("B", C.c_double * 4),
        ("dt", C.c_double),
        ("x0", C.c_double * 4),
        ("u0", C.c_double),
    ]


GRID_MAX_DEVICES = 64


class GridConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("hip_device", C.c_int32), ("num_envs", C.c_int64),
        ("num_devices", C.c_int32), ("mobile", C.c_int32),
        ("pos", (C.c_double * 2) * GRID_MAX_DEVICES),
        ("slot", C.c_double), ("frequency", C.c_double), ("bandwidth", C.c_double), ("temperature_c", C.c_double),
        ("bit_rate", C.c_double), ("code_rate", C.c_double), ("max_ber", C.c_double),
        ("tx_power_dbm", C.c_double), ("send_interval", C.c_double),
        ("header_bytes", C.c_int32), ("payload_bytes", C.c_int32),
        ("move_interval", C.c_double), ("move_span", C.c_double), ("seed", C.c_uint64),
    ]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("steps", "transmissions", "delivered", "appended",
                                           "popped", "dropped", "flags_or", "bad_actions")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class NativeError(RuntimeError):
    def __init__(self, code, message):
        RuntimeError.__init__(self, "gymwipe_amd native error %d: %s" % (code, message))
        self.code = code


def build(force=False, quiet=True):
    """Compile the HIP extension in-tree (hipcc --offload-arch=gfx950)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)
            if f.endswith((".hip", ".cpp", ".h"))] + [os.path.join(_PKG, "..", "include", "gymwipe_amd.h")]
    shim = os.path.join(os.path.dirname(LIB_PATH), "_gw_fast.so")            # CPython fast-call shim, same Makefile
    xoff = os.path.join(os.path.dirname(LIB_PATH), "libgymwipe_amd_xnackoff.so")
    stale = (not os.path.exists(LIB_PATH) or not os.path.exists(shim) or not os.path.exists(xoff)
             or os.path.getmtime(xoff) < os.path.getmtime(LIB_PATH)
             or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
             or os.path.getmtime(os.path.join(CSRC, "gw_pyfast.c")) > os.path.getmtime(shim))
    if force or stale:
        cmd = ["make", "-C", CSRC] + (["-B"] if force else [])
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL if quiet else None)
    return LIB_PATH


_lib = None


def _pick_library():
    """The library to load: $GW_LIB if given; else the build for XNACK-off devices (libgymwipe_amd_xnackoff.so, 3.5 % faster
    step kernels) when the GPU says it runs with XNACK off -- a code object compiled for xnack- would not load otherwise --;
    else the build for any XNACK setting.  Same sources, same C-ABI, same results."""
    if os.environ.get("GW_LIB"):
        return LIB_PATH
    xoff = os.path.join(os.path.dirname(LIB_PATH), "libgymwipe_amd_xnackoff.so")
    if os.path.exists(xoff) and not os.environ.get("GW_NO_XNACKOFF"):
        try:
            import torch
            if torch.cuda.is_available():
                names = {torch.cuda.get_device_properties(i).gcnArchName for i in range(torch.cuda.device_count())}
                if names and all("xnack-" in n for n in names):
                    return xoff
        except Exception:
            pass
    return LIB_PATH


def lib():
    """The loaded C-ABI library.  Raises if it is missing -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH) and not os.environ.get("GW_LIB"):
        try:                                   # clean checkout: compile the HIP extension in-tree once
            build(force=True)
        except Exception as exc:               # no hipcc / compile error: fail loudly, never fall back
            raise ImportError(
                "gymwipe_amd: %s is missing and could not be built (%s). Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C gymwipe_amd/csrc` (needs hipcc). "
                "There is no CPU fallback." % (LIB_PATH, exc))
    if not os.path.exists(LIB_PATH):
        raise ImportError("gymwipe_amd: %s is missing. There is no CPU fallback." % LIB_PATH)
    try:
        import torch  # noqa: F401  -- load torch's HIP runtime first so both share one libamdhip64
    except Exception:
        pass
    L = C.CDLL(_pick_library())
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.gw_abi_version.argtypes, L.gw_abi_version.restype = [], C.c_int
    L.gw_last_error.argtypes, L.gw_last_error.restype = [], C.c_char_p
    L.gw_device_count.argtypes, L.gw_device_count.restype = [C.POINTER(C.c_int)], C.c_int
    L.gw_config_default.argtypes, L.gw_config_default.restype = [C.POINTER(Config), i64, i32], C.c_int
    L.gw_create.argtypes, L.gw_create.restype = [C.POINTER(Config), C.POINTER(vp)], C.c_int
    L.gw_destroy.argtypes, L.gw_destroy.restype = [vp], C.c_int
    L.gw_reset.argtypes, L.gw_reset.restype = [vp, vp, vp, vp], C.c_int
    L.gw_step.argtypes, L.gw_step.restype = [vp, vp, vp, vp, vp, vp, vp], C.c_int
    L.gw_step_fb.argtypes, L.gw_step_fb.restype = [vp, vp, vp, vp, vp, vp, vp, vp], C.c_int
    L.gw_rollout.argtypes, L.gw_rollout.restype = [vp, i32, vp, vp, vp, vp, vp, vp], C.c_int
    L.gw_received.argtypes, L.gw_received.restype = [vp, vp, vp], C.c_int
    L.gw_enqueue.argtypes, L.gw_enqueue.restype = [vp, i32, vp, vp], C.c_int
    L.gw_pack_feedback.argtypes, L.gw_pack_feedback.restype = [vp, C.c_int64, vp, vp, vp, vp, i32, vp], C.c_int
    L.gw_unpack_feedback.argtypes, L.gw_unpack_feedback.restype = [vp, C.c_int64, vp, vp, vp, vp, vp], C.c_int
    L.gw_delivered.argtypes, L.gw_delivered.restype = [vp, vp, vp], C.c_int
    L.gw_get_state.argtypes, L.gw_get_state.restype = [vp, C.c_char_p, vp, C.c_size_t], C.c_int
    L.gw_stats_read.argtypes, L.gw_stats_read.restype = [vp, C.POINTER(Stats)], C.c_int
    L.gw_state_bytes.argtypes, L.gw_state_bytes.restype = [vp, C.POINTER(C.c_uint64)], C.c_int
    L.gw_snapshot_bytes.argtypes, L.gw_snapshot_bytes.restype = [vp, C.POINTER(C.c_uint64)], C.c_int
    L.gw_get_snapshot.argtypes, L.gw_get_snapshot.restype = [vp, vp, C.c_uint64], C.c_int
    L.gw_set_state.argtypes, L.gw_set_state.restype = [vp, vp, C.c_uint64], C.c_int
    L.gw_link_info.argtypes = [vp, i32, i32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.gw_link_info.restype = C.c_int
    L.gw_noise_states.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(C.c_double)]
    L.gw_noise_states.restype = C.c_int
    L.gw_selftest_queue.argtypes, L.gw_selftest_queue.restype = [C.c_uint64, i32, i32, i32], C.c_int
    L.gw_set_position.argtypes, L.gw_set_position.restype = [vp, i32, vp, vp, vp, vp], C.c_int
    L.gw_set_positions.argtypes, L.gw_set_positions.restype = [vp, vp, vp, vp], C.c_int
    L.gw_clear_flags.argtypes, L.gw_clear_flags.restype = [vp, vp], C.c_int
    L.gw_selftest_runq.argtypes, L.gw_selftest_runq.restype = [C.c_uint64, i32, i32, i32], C.c_int
    L.gw_selftest_fastmath.argtypes = [C.POINTER(Config), C.POINTER(i32)]
    L.gw_selftest_fastmath.restype = C.c_int
    L.gw_plant_config_default.argtypes, L.gw_plant_config_default.restype = [C.POINTER(PlantConfig), i64], C.c_int
    L.gw_plant_create.argtypes, L.gw_plant_create.restype = [C.POINTER(PlantConfig), C.POINTER(vp)], C.c_int
    L.gw_plant_destroy.argtypes, L.gw_plant_destroy.restype = [vp], C.c_int
    L.gw_plant_update.argtypes, L.gw_plant_update.restype = [vp, vp, i64, vp], C.c_int
    L.gw_plant_set_input.argtypes, L.gw_plant_set_input.restype = [vp, vp, vp, vp], C.c_int
    L.gw_plant_state_ptr.argtypes, L.gw_plant_state_ptr.restype = [vp, C.POINTER(vp)], C.c_int
    L.gw_plant_get_state.argtypes, L.gw_plant_get_state.restype = [vp, C.c_char_p, vp, C.c_size_t], C.c_int
    L.gw_plant_feedback.argtypes, L.gw_plant_feedback.restype = [vp, vp, vp, vp, vp], C.c_int
    L.gw_ctrl_config_default.argtypes, L.gw_ctrl_config_default.restype = [C.POINTER(CtrlConfig), i64], C.c_int
    L.gw_ctrl_create.argtypes, L.gw_ctrl_create.restype = [C.POINTER(CtrlConfig), C.POINTER(vp)], C.c_int
    L.gw_ctrl_destroy.argtypes, L.gw_ctrl_destroy.restype = [vp], C.c_int
    L.gw_ctrl_step.argtypes, L.gw_ctrl_step.restype = [vp, vp, vp, vp, vp, vp, vp], C.c_int
    L.gw_ctrl_get_state.argtypes, L.gw_ctrl_get_state.restype = [vp, C.c_char_p, vp, C.c_size_t], C.c_int
    L.gw_plant_update_feedback.argtypes, L.gw_plant_update_feedback.restype = [vp, vp, C.c_int64, vp, vp, vp, vp], C.c_int
    L.gw_now_ptr.argtypes, L.gw_now_ptr.restype = [vp, C.POINTER(vp), C.POINTER(i64)], C.c_int
    L.gw_pendulum_step.argtypes, L.gw_pendulum_step.restype = [vp, vp, vp, vp, vp, vp, vp, vp], C.c_int
    L.gw_grid_config_default.argtypes, L.gw_grid_config_default.restype = [C.POINTER(GridConfig), i64, i32], C.c_int
    L.gw_grid_create.argtypes, L.gw_grid_create.restype = [C.POINTER(GridConfig), vp, C.POINTER(vp)], C.c_int
    L.gw_grid_destroy.argtypes, L.gw_grid_destroy.restype = [vp], C.c_int
    L.gw_grid_run.argtypes, L.gw_grid_run.restype = [vp, C.c_double, vp], C.c_int
    L.gw_grid_get_state.argtypes, L.gw_grid_get_state.restype = [vp, C.c_char_p, vp, C.c_size_t], C.c_int
    L.gw_grid_set_position.argtypes, L.gw_grid_set_position.restype = [vp, i32, vp, vp, vp], C.c_int
    if L.gw_abi_version() != ABI_VERSION:
        raise ImportError("gymwipe_amd: ABI mismatch (library %d, python %d); rebuild"
                          % (L.gw_abi_version(), ABI_VERSION))
    _lib = L
    return L


_fast = False


def fast():
    """The CPython fast-call shim for the per-step entry points (csrc/gw_pyfast.c), bound to the loaded library's
    gw_step / gw_pendulum_step -- or None when it has not been built (the callers then go through ctypes: same
    library, same kernels, ~1 us more host time per call)."""
    global _fast
    if _fast is False:
        _fast = None
        path = os.path.join(os.path.dirname(LIB_PATH), "_gw_fast.so")
        if os.path.exists(path) and not os.environ.get("GW_NO_PYFAST"):
            try:
                import importlib.machinery
                import importlib.util
                loader = importlib.machinery.ExtensionFileLoader("_gw_fast", path)
                spec = importlib.util.spec_from_loader("_gw_fast", loader)
                mod = importlib.util.module_from_spec(spec)
                loader.exec_module(mod)
                L = lib()
                mod.bind(C.cast(L.gw_step, C.c_void_p).value, C.cast(L.gw_pendulum_step, C.c_void_p).value,
                         C.cast(L.gw_step_fb, C.c_void_p).value)
                _fast = mod
            except Exception:
                _fast = None
    return _fast


def check(rc):
    if rc != OK:
        raise NativeError(rc, (lib().gw_last_error() or b"").decode("utf-8", "replace"))


def default_config(num_envs, num_devices):
    cfg = Config()
    check(lib().gw_config_default(C.byref(cfg), int(num_envs), int(num_devices)))
    return cfg
