"""The driver's contract with bench.py (one JSON line on stdout; keys, types, roofline and cpu_baseline objects),
checked on a small run -- `-m gpu`."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "128", "--warmup", "64",
                          "--envs", "8192", "--cpu-seconds", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    with open(os.path.join(ROOT, "BASELINE.json")) as fh:
        base = json.load(fh)
    assert d["metric"] in base["metric"] and d["unit"] == "env-steps/s"
    assert d["n_gpus"] == 1 and d["steps"] == 128 and d["warmup"] == 64
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 1e6 and abs(d["value"] - 8192 * 128 / (d["ms_per_step"] * 1e-3 * 128)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    assert r["traffic"] is None or r["traffic"] > 0            # PMC profile on file only for the 65 536-env shapes
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "env-steps/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    for sec in ("fused_rollout", "graph_replay", "steady_state_no_reset"):
        assert "error" not in d[sec], d[sec]
    assert d["repeats"] >= 1 and d["timed_step_indices_after_reset"][0] == 0 and d["us_per_step"]["min"] <= d["us_per_step"]["max"]
    assert r["frac"] <= 1.0 or "frac_above_one_because" in r
    assert r["min_bytes_per_env_step"] == 125 and r["frac_min_bytes"] > 0
    assert c["by_devices"]["4"]["value"] == c["value"] and c["nproc"] >= c["cores"]


@pytest.mark.gpu
def test_bench_config4_prints_a_contract_line_with_one_launch_per_step():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "4", "--steps", "20", "--warmup", "5",
                          "--envs", "4096", "--repeats", "20", "--cpu-seconds", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["unit"] == "env-steps/s" and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["repeats"] == 20
    assert "InvertedPendulum" in d["config"]["workload"] and d["dtype"] == "f64" and d["vs_baseline"] is None
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["launches_per_step"] == 1 and r["kernel"] == "pend_step_kernel"
    assert r["mfma"]["instruction"] == "v_mfma_f64_16x16x4_f64" and 5 < r["mfma"]["plant_substeps_per_env_step"] < 15
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0


@pytest.mark.gpu
def test_bench_multi_gpu_code_path_runs_on_rccl_with_one_rank():
    """GW_BENCH_FORCE_GATHER=1: the N > 1 path of bench.py (process group on RCCL, chunked feedback gather every 16 steps with
    the warm-up flush and the end-of-region drain, the per-step observation gather) with a single rank -- every collective call
    the driver's multi-GPU runs make, on the one GPU this box has."""
    env = dict(os.environ, GW_BENCH_FORCE_GATHER="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                          "--envs", "8192", "--repeats", "30", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                         cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-1000:]                  # ONE line on stdout although RCCL prints a banner to fd 1
    d = json.loads(lines[0])
    assert d["config"]["obs_gather"] == "per-step" and "after EVERY env.step()" in d["config"]["parallelism"]
    g = d["gather_forms"]
    for form, late in (("no_gather", None), ("pipelined_gather", 1), ("chunked_gather", 16)):
        assert "error" not in g[form], g[form]
        assert g[form]["observations_late_by_steps"] == late and g[form]["env_steps_per_s"] > 1e6
    assert g["pipelined_gather"]["bytes_per_rank_per_step"] == 9 * 8192 and g["chunked_gather"]["bytes_per_rank_per_step"] == 8192
    assert d["value"] > 1e6 and d["value"] <= g["no_gather"]["env_steps_per_s"] * 1.5
    r = d["ranks"]
    assert r["world_size"] == 1 and r["backend"] == "nccl" and len(r["devices"]) == 1 and r["devices"][0] and r["rccl_version"]
    assert "without the per-step gather" in d["roofline"]["how"].lower()


@pytest.mark.gpu
def test_plain_bench_gpus_2_starts_its_own_ranks():
    """`python3 bench.py --gpus 2` with NO launcher: the parent starts two fresh rank processes before touching HIP, relays
    rank 0's one line and exits 0.  Two RCCL ranks cannot share the one GPU of this box, so the collectives run on gloo
    (GW_BENCH_BACKEND=gloo: the rehearsal mode) -- launcher, rendezvous, sharded action stream, per-step gather protocol,
    max-over-ranks timing and the `ranks` evidence are the code the driver's N-GPU command runs."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["GW_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                          "--envs", "8192", "--repeats", "10"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-1000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_envs"] == 2 * 8192 and d["config"]["obs_gather"] == "per-step"
    assert "cpu_baseline" not in d and d["scaling"] == "weak" and d["value"] > 1e5
    r = d["ranks"]
    assert r["world_size"] == 2 and r["backend"] == "gloo" and len(r["devices"]) == 2 and "launch_ranks" in r["launcher"]
    assert sorted(e["rank"] for e in r["per_rank"]) == [0, 1] and len({e["pid"] for e in r["per_rank"]}) == 2
    for form in ("no_gather", "pipelined_gather", "chunked_gather"):
        assert "error" not in d["gather_forms"][form], d["gather_forms"][form]
