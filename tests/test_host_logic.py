"""
CPU tier: host-side logic and the C-ABI surface.  No compute call is made without a GPU: the
library must load, export every symbol include/gymwipe_amd.h declares, refuse loudly to create
an env without a HIP device, and its host-only self-tests must pass.
"""
import ctypes as C
import os
import json
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _have_gpu():
    import torch
    return torch.cuda.is_available()


def test_library_exports_every_declared_symbol(native_lib):
    hdr = open(os.path.join(ROOT, "include", "gymwipe_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(gw_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 15
    from gymwipe_amd import _native
    assert sorted(_native.EXPORTS) == declared, "python binding and header disagree"
    for name in declared:
        assert getattr(native_lib, name) is not None
    assert native_lib.gw_abi_version() == 1


def test_config_default_matches_reference_constants(native_lib):
    """SURVEY.md Appendix E / the class constants of the reference."""
    from gymwipe_amd import _native
    cfg = _native.default_config(65536, 2)
    assert (cfg.pos[0][0], cfg.pos[0][1]) == (0.0, 2.0)          # counter_traffic.py:125
    assert (cfg.pos[1][0], cfg.pos[1][1]) == (0.0, -2.0)         # :126
    assert (cfg.pos[2][0], cfg.pos[2][1]) == (0.0, 0.0)          # :133
    assert list(cfg.mult)[:2] == [1, 3] and list(cfg.dest)[:2] == [1, 0]
    assert cfg.slot == 1e-6 and cfg.frequency == 2.4e9 and cfg.bandwidth == 22e6
    assert cfg.bit_rate == 133.33333e3 and cfg.code_rate == 0.75 and cfg.max_ber == 0.25
    assert cfg.tx_power_dbm == 0.0 and cfg.counter_interval == 0.001 and cfg.counter_bound == 65536
    assert (cfg.mac_header_bytes, cfg.net_header_bytes) == (13, 12)
    assert (cfg.duration_factor, cfg.max_duration, cfg.payload_value) == (1000, 20, 2)
    # D = 4: circle of radius 2 m, multiplicities 1,3,1,3; same layout as the oracle
    from oracle.ct_oracle import default_config as ocfg
    c4, o4 = _native.default_config(8, 4), ocfg(4)
    for i in range(5):
        assert (c4.pos[i][0], c4.pos[i][1]) == (o4.pos[i][0], o4.pos[i][1])
    assert list(c4.mult)[:4] == list(o4.mult)[:4] == [1, 3, 1, 3]
    assert native_lib.gw_config_default(C.byref(cfg), 8, 1) < 0   # D out of range
    assert b"num_devices" in native_lib.gw_last_error()


def test_suffix_encoding_ceil_div_is_exact_for_every_multiplicity():
    """gw_ceil_div (gw_queue.h): ((len + mult - 1) * ceil(65536 / mult)) >> 16 == ceil(len / mult) for every queue length
    0..100 and every multiplicity the handle accepts (1..100) -- and beyond, up to 256."""
    for m in range(1, 257):
        inv16 = (65536 + m - 1) // m
        for ln in range(0, 101):
            assert ((ln + m - 1) * inv16) >> 16 == (ln + m - 1) // m, (m, ln)


@pytest.mark.parametrize("mult,bound", [(1, 65536), (3, 65536), (3, 40), (2, 7), (15, 1), (5, 300), (16, 65536), (37, 65536),
                                        (64, 9), (99, 300), (100, 65536)])
def test_queue_encoding_fuzz_against_explicit_deque(native_lib, mult, bound):
    """gw_queue.h (the code the kernel runs) vs deque(maxlen=100): ticks, resets, pops."""
    for seed in range(3):
        assert native_lib.gw_selftest_queue(seed, 30000, mult, bound) == 0


@pytest.mark.parametrize("mult,bound", [(1, 65536), (3, 65536), (3, 40), (2, 7), (15, 1), (5, 300), (37, 65536), (100, 65536), (64, 9)])
def test_run_length_queue_fuzz_against_explicit_deque(native_lib, mult, bound):
    """gw_runq.h (the generic kernel's queues) vs deque(maxlen=100): ticks, resets, pops, arbitrary enqueued packets."""
    for seed in range(3):
        assert native_lib.gw_selftest_runq(seed, 30000, mult, bound) == 0


def test_fast_paths_validate_for_default_configs(native_lib):
    from gymwipe_amd import _native
    for D, want_states in ((2, 3), (4, 4), (16, 6), (32, 8)):
        cfg = _native.default_config(1024, D)
        mx = C.c_int32()
        mask = native_lib.gw_selftest_fastmath(C.byref(cfg), C.byref(mx))
        assert mask == 31, "D=%d: fast paths %d" % (D, mask)
        assert mx.value == want_states
    cfg = _native.default_config(16, 4)
    cfg.code_rate, cfg.max_ber = 0.5, 0.11                # integer decode shortcut must switch itself off
    assert native_lib.gw_selftest_fastmath(C.byref(cfg), None) & 4 == 0
    cfg = _native.default_config(16, 4)
    cfg.mult[1] = 99                                      # up to the deque's capacity in both queue encodings (rounds 1-2: 15)
    assert native_lib.gw_selftest_fastmath(C.byref(cfg), None) == 31
    cfg.mult[1] = 101
    assert native_lib.gw_selftest_fastmath(C.byref(cfg), None) < 0


@pytest.mark.skipif(_have_gpu(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback_without_a_gpu(native_lib):
    """The product path must fail loudly when there is no HIP device -- never fall back."""
    from gymwipe_amd import _native
    cfg = _native.default_config(16, 2)
    h = C.c_void_p()
    rc = native_lib.gw_create(C.byref(cfg), C.byref(h))
    assert rc == _native.ENODEVICE and not h
    assert b"no CPU fallback" in native_lib.gw_last_error()
    import gymwipe_amd
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        gymwipe_amd.VecCounterTrafficEnv(16)
    with pytest.raises(RuntimeError):
        gymwipe_amd.make("CounterTraffic-v0")


def test_product_path_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under gymwipe_amd/ or include/ may reference it."""
    bad = []
    pat = re.compile(r"(from\s+oracle|import\s+oracle|ct_oracle|des_model|libct_oracle|oracle[/\\.]\w)")
    for base in ("gymwipe_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".h")):
                    for line in open(os.path.join(dirpath, f), errors="replace"):
                        if pat.search(line):
                            bad.append((f, line.strip()))
    assert not bad, bad
    import subprocess
    out = subprocess.run(["ldd", os.path.join(ROOT, "gymwipe_amd", "lib", "libgymwipe_amd.so")],
                         capture_output=True, text=True).stdout
    assert "ct_oracle" not in out


def test_spaces_and_registry_surface():
    import gymwipe_amd
    from gymwipe_amd import spaces
    act = spaces.Dict({"device": spaces.Discrete(2), "duration": spaces.Discrete(20)})   # envs/core.py:39-42
    assert act.contains({"device": 1, "duration": 19})
    assert not act.contains({"device": 2, "duration": 0})
    assert not act.contains({"device": 0, "duration": 20})
    assert not act.contains({"device": 0})
    assert not act.contains({"device": 0.0, "duration": 1})
    assert act.contains({"device": np.int64(1), "duration": np.int32(3)})
    act.seed(1)
    for _ in range(50):
        assert act.contains(act.sample())
    assert spaces.Discrete(131072).n == 2 * 65536                                       # counter_traffic.py:120
    assert set(gymwipe_amd.registry) >= {"CounterTraffic-v0", "VecCounterTraffic-v0", "InvertedPendulum-v0"}   # envs/__init__.py:6-14
    with pytest.raises(KeyError):
        gymwipe_amd.make("NoSuchEnv-v0")
    with pytest.raises(RuntimeError):                  # no GPU here: the envs refuse to run, there is no CPU fallback
        gymwipe_amd.make("InvertedPendulum-v0")
    P = gymwipe_amd.VecInvertedPendulumEnv
    assert (P.SENSOR, P.CONTROLLER, P.SAMPLE_INTERVAL) == (0, 1, 0.001)                # envs/inverted_pendulum.py:79,86-89
    E = gymwipe_amd.VecCounterTrafficEnv
    assert (E.MAX_ASSIGN_DURATION, E.ASSIGNMENT_DURATION_FACTOR) == (20, 1000)          # envs/core.py:25,27
    assert (E.COUNTER_INTERVAL, E.COUNTER_BYTE_LENGTH, E.COUNTER_BOUND) == (0.001, 2, 65536)
    for name in ("onPacketReceived", "onFrequencyBandAssignment", "getReward", "getObservation",
                 "getDone", "getInfo", "getFeedback", "reset"):                        # envs/core.py:59-159
        assert hasattr(gymwipe_amd.Interpreter, name)


def test_shard_range_partitions_the_batch():
    from gymwipe_amd.sharding import shard_range
    for total, world in ((524288, 8), (65536, 1), (10, 4), (7, 8)):
        spans = [shard_range(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        for a, b in zip(spans, spans[1:]):
            assert a[1] == b[0]
        assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def test_bench_algorithmic_byte_model():
    """B(D) = 17 + 2*(12 + 20*D) + 4*(k_app + k_pop)."""
    import bench
    assert bench.algorithmic_bytes(4, 1, 0, 0) == 17 + 2 * (12 + 80)
    assert bench.algorithmic_bytes(2, 10, 440, 10) == 10 * (17 + 2 * 52) + 4 * 450
    assert bench.HBM_PEAK == 8.0e12


def test_oracle_is_clean_under_asan_ubsan():
    """tests/sanitize_cpu.sh: the C oracle under AddressSanitizer + UBSan (CPU only; `--host` also runs the
    library's host side, which needs hipcc and takes longer)."""
    import shutil
    import subprocess
    if not shutil.which("gcc") or not os.path.exists(subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True,
                                                                     text=True).stdout.strip()):
        pytest.skip("no gcc/libasan")
    out = subprocess.run(["bash", os.path.join(ROOT, "tests", "sanitize_cpu.sh")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "sanitizers: clean" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_c_abi_rejects_bad_arguments_without_a_gpu(native_lib):
    """Error convention: every entry returns 0 or a negative GW_E* and leaves a message in
    gw_last_error(); configuration errors are reported before any device is touched."""
    from gymwipe_amd import _native as nat
    L = native_lib
    h = C.c_void_p()
    assert L.gw_create(None, C.byref(h)) == nat.EINVAL and b"NULL" in L.gw_last_error()
    cfg = nat.Config()
    assert L.gw_config_default(C.byref(cfg), 64, 1) == nat.EINVAL          # observation needs senders 0 and 1
    assert L.gw_config_default(C.byref(cfg), 64, nat.MAX_DEVICES + 1) == nat.EINVAL
    for mutate, word in ((lambda c: setattr(c, "num_envs", 0), b"num_envs"),
                         (lambda c: setattr(c, "abi_version", 99), b"abi_version"),
                         (lambda c: c.mult.__setitem__(1, -1), b"mult"),
                         (lambda c: c.dest.__setitem__(0, 7), b"dest"),
                         (lambda c: setattr(c, "slot", 0.0), b"slot"),
                         (lambda c: setattr(c, "flags", nat.CFG_PEER_RECEIVE), b"EXPLICIT_QUEUE"),
                         (lambda c: c.extra_att_db[0].__setitem__(1, 3.0), b"symmetric"),
                         (lambda c: c.mult.__setitem__(0, 101), b"mult")):                       # (<= 100 in both queue encodings)
        cfg = nat.default_config(64, 4)
        mutate(cfg)
        rc = L.gw_create(C.byref(cfg), C.byref(h))
        assert rc in (nat.EINVAL, nat.EUNSUPPORTED), (rc, L.gw_last_error())
        assert word in L.gw_last_error(), L.gw_last_error()
        assert not h.value
    # 32-bit record offsets of the default kernels: e * RB must fit (RB = 48 B at D = 16, 80 B at D = 32, >= 32 B always)
    for D, most in ((4, 0xffffffff // 32), (16, 0xffffffff // 48), (32, 0xffffffff // 80)):
        cfg = nat.default_config(most + 1, D)
        assert L.gw_create(C.byref(cfg), C.byref(h)) == nat.EUNSUPPORTED and b"32-bit record offsets" in L.gw_last_error()
        cfg = nat.default_config(most, D)                                   # the bound itself passes validation (host-only entry)
        assert L.gw_selftest_fastmath(C.byref(cfg), None) >= 0
    # handle-taking entries refuse a NULL handle
    assert L.gw_clear_flags(None, None) == nat.EINVAL
    assert L.gw_step(None, None, None, None, None, None, None) == nat.EINVAL
    assert L.gw_reset(None, None, None, None) == nat.EINVAL
    assert L.gw_rollout(None, 1, None, None, None, None, None, None) == nat.EINVAL
    assert L.gw_enqueue(None, 0, None, None) == nat.EINVAL
    assert L.gw_pack_feedback(None, 4, None, None, None, None, 0, None) == nat.EINVAL
    assert L.gw_get_state(None, b"now", None, 0) == nat.EINVAL
    assert L.gw_destroy(None) == nat.OK                                  # like free(NULL)
    assert L.gw_selftest_queue(1, 10, 0, 10) == nat.EINVAL


def test_hot_kernels_use_no_scratch_memory(tmp_path):
    """Every step / rollout kernel must keep its state in registers: private_segment_fixed_size == 0 for each
    instantiation.  (Twice this build the compiler folded a select between array elements -- or between two arrays -- into
    dynamic addressing of a stack copy, which costs 10-15 % of a launch and shows up nowhere else.)  Cross-compiles the
    device code only; no GPU needed."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    csrc = os.path.join(ROOT, "gymwipe_amd", "csrc")
    # the product build's own flags (scheduling strategy, kernarg preload, ...: they change register allocation)
    flags = subprocess.run(["make", "-s", "-C", csrc, "print-flags"], check=True, capture_output=True, text=True).stdout.split()
    assert "--offload-arch=gfx950" in flags and "-ffp-contract=off" in flags, flags
    xoff = [f + ":xnack-" if f == "--offload-arch=gfx950" else f for f in flags]       # the second library's target
    for src, fl in (("ct_step_sfx.hip", flags), ("ct_rollout_sfx.hip", flags), ("ct_step.hip", flags), ("ct_step_dyn.hip", flags),
                    ("ct_step_sfx.hip", xoff), ("ct_rollout_sfx.hip", xoff)):
        out = tmp_path / (src + ".s")
        subprocess.run([hipcc] + fl + ["-S", "--cuda-device-only", "-o", str(out), "-x", "hip", os.path.join(csrc, src)],
                       check=True, capture_output=True, timeout=900, cwd=csrc)
        text = out.read_text()
        sizes = re.findall(r"^\s+\.private_segment_fixed_size:\s+(\d+)", text, re.M)
        names = re.findall(r"^\s+\.name:\s+(\S+)", text, re.M)
        assert sizes and len(sizes) == len(names), src
        bad = [(n, int(s)) for n, s in zip(names, sizes) if int(s) != 0]
        assert not bad, "%s: kernels with scratch memory: %s" % (src, bad)


def test_fastcall_shim_reaches_the_same_entry_points():
    """csrc/gw_pyfast.c: the CPython shim env.step() uses instead of ctypes calls the library's own gw_step /
    gw_pendulum_step (argument validation answers without a GPU) and rejects malformed calls with a Python error."""
    from gymwipe_amd import _native as nat
    f = nat.fast()
    assert f is not None, "gymwipe_amd/lib/_gw_fast.so missing: make -C gymwipe_amd/csrc"
    L = nat.lib()
    assert f.step(0, 1, 1, 1, 1, 1, 0) == nat.EINVAL == L.gw_step(None, 1, 1, 1, 1, 1, None)
    assert b"env is NULL" in L.gw_last_error()
    assert f.pendulum_step(0, 0, 1, 1, 1, 1, 1, 0) == L.gw_pendulum_step(None, None, 1, 1, 1, 1, 1, None) != nat.OK
    with pytest.raises(TypeError):
        f.step(0, 1, 1)
    with pytest.raises((TypeError, OverflowError)):
        f.step(0, 1, 1, 1, 1, "x", 0)


def test_chunked_gather_one_call_protocol_matches_the_two_call_one():
    """ChunkedFeedbackGather.begin()/advance() (one call per step, what bench.py uses) submits the same chunks, in the same
    order and with the same contents, as slot()/stepped(), including a drain in the middle and a ragged tail."""
    import torch
    from gymwipe_amd.sharding import ChunkedFeedbackGather

    class Work:
        def wait(self):
            pass

    class Dist:
        def __init__(self):
            self.calls = []

        def all_gather_into_tensor(self, out, inp, async_op=True):
            out[:inp.numel()] = inp
            self.calls.append(inp.clone())
            return Work()

    def pack(o, r, d, out):
        out.copy_((o % 251).to(torch.uint8))

    for K in (7, 16, 37):
        a, b = Dist(), Dist()
        two = ChunkedFeedbackGather(5, "cpu", pack, 1, chunk=4, dist_module=a)
        one = ChunkedFeedbackGather(5, "cpu", pack, 1, chunk=4, dist_module=b)
        v = one.begin()
        for k in range(K):
            o, r, d = two.slot()
            o.fill_(k)
            two.stepped()
            assert len(v) == 4 and v[3].dtype == torch.uint8 and v[3].shape == (5,)
            v[0].fill_(k)
            v = one.advance()
            if k == 9:
                two.drain()
                one.drain()
                v = one.begin()
        two.drain()
        one.drain()
        assert len(a.calls) == len(b.calls) and all(torch.equal(x, y) for x, y in zip(a.calls, b.calls)), K


def test_both_library_builds_export_the_c_abi_and_the_loader_defaults_to_the_portable_one():
    """libgymwipe_amd.so (any XNACK setting) and libgymwipe_amd_xnackoff.so (xnack- code objects, picked by the loader only when
    the device reports xnack-) are the same sources: both export every declared symbol.  Without a GPU the loader takes the
    portable build."""
    import ctypes
    from gymwipe_amd import _native as nat
    lib_dir = os.path.dirname(nat.LIB_PATH)
    for name in ("libgymwipe_amd.so", "libgymwipe_amd_xnackoff.so"):
        path = os.path.join(lib_dir, name)
        assert os.path.exists(path), "%s missing: make -C gymwipe_amd/csrc" % path
        L = ctypes.CDLL(path)
        for sym in nat.EXPORTS:
            getattr(L, sym)
        assert L.gw_abi_version() == nat.ABI_VERSION
    import torch
    if not torch.cuda.is_available() and not os.environ.get("GW_LIB"):
        assert nat._pick_library() == nat.LIB_PATH


# ---- bench.py's own rank launcher (plain `python bench.py --gpus N`): spawn / relay / failure logic, no GPU involved ----
def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("gw_bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                     # top level imports neither torch nor the package
    return mod


def test_bench_launcher_starts_n_ranks_and_relays_rank_zero(capfd):
    bench = _bench_module()
    child = ("import os, sys, json; r = int(os.environ['RANK']); "
             "assert os.environ['WORLD_SIZE'] == '3' and os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1'; "
             "int(os.environ['MASTER_PORT']); "
             "print(json.dumps({'rank': r, 'port': os.environ['MASTER_PORT']})); sys.stderr.write('err%d\\n' % r)")
    rc = bench.launch_ranks(3, [sys.executable, "-c", child], timeout=60)
    out, err = capfd.readouterr()
    assert rc == 0
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0])["rank"] == 0          # only rank 0's stdout reaches stdout
    assert '"rank": 1' in err and '"rank": 2' in err                      # the other ranks' stdout went to stderr
    assert "err0" in err and "err2" in err


def test_bench_launcher_fails_the_job_when_one_rank_fails_and_stops_the_rest():
    import time as _time
    bench = _bench_module()
    child = ("import os, sys, time; r = int(os.environ['RANK']);\n"
             "if r == 1: sys.exit(7)\n"
             "time.sleep(600)")
    t0 = _time.monotonic()
    rc = bench.launch_ranks(2, [sys.executable, "-c", child], timeout=120, grace=0.5)
    assert rc == 7 and _time.monotonic() - t0 < 30                        # rank 0 (sleeping) was terminated, not waited for


def test_bench_launcher_times_out():
    bench = _bench_module()
    rc = bench.launch_ranks(2, [sys.executable, "-c", "import time; time.sleep(600)"], timeout=1.0, grace=0.5)
    assert rc == 124


def test_bench_self_launch_happens_before_torch_is_imported():
    """`bench.py --gpus 2` without WORLD_SIZE must reach launch_ranks with neither torch nor the package imported in the
    parent (a parent that had initialised HIP could not safely start GPU children)."""
    code = ("import sys, runpy\n"
            "sys.argv = ['bench.py', '--gpus', '2', '--steps', '1', '--warmup', '0']\n"
            "import importlib.util\n"
            "spec = importlib.util.spec_from_file_location('b', %r); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
            "def fake(n, argv, timeout=None, env=None, grace=10.0):\n"
            "    assert 'torch' not in sys.modules and 'gymwipe_amd' not in sys.modules, 'imported before the launch'\n"
            "    assert n == 2 and argv[0] == sys.executable and argv[1].endswith('bench.py') and argv[2:] == sys.argv[1:]\n"
            "    return 5\n"
            "b.launch_ranks = fake\n"
            "b.main()\n" % os.path.join(ROOT, "bench.py"))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert out.returncode == 5, (out.stdout, out.stderr[-2000:])
