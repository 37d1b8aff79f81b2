#!/bin/bash
# Run ON THE GPU BOX (through gpurun): every bench line the docs quote, from one box and one build, into gpurun_out/final_<tag>/
# (copy into profiles/<tag>/).  About 6 minutes.  usage: bash tools/final_round.sh r2_final
TAG=${1:-final}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/final_$TAG
mkdir -p $OUT
cd $REPO
B="python3 bench.py"
run() { name=$1; shift; echo "== $name: $@"; "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "FAILED $name"; tail -5 $OUT/$name.err; }; }
run bench_driver_cmd              $B --gpus 1 --steps 20 --warmup 5
run bench_default                 $B
run bench_config3_d16_default     $B --config 3 --cpu-seconds 4
run bench_config3_d16_driver_window $B --config 3 --steps 20 --warmup 5 --no-cpu-baseline
run bench_d2_default              $B --devices 2 --cpu-seconds 4
run bench_d7                      $B --devices 7 --steps 128 --no-cpu-baseline
run bench_d32                     $B --devices 32 --steps 128 --no-cpu-baseline
run bench_config4_default         $B --config 4 --cpu-seconds 4
run bench_config4_driver_window   $B --config 4 --steps 20 --warmup 5 --no-cpu-baseline
run occ_131072                    $B --envs 131072 --steps 256 --no-cpu-baseline
run occ_262144                    $B --envs 262144 --steps 256 --no-cpu-baseline
run occ_524288                    $B --envs 524288 --steps 256 --no-cpu-baseline
run generic_kernel                python3 tools/bench_generic.py
run live_phy                      python3 tools/bench_live_phy.py
run control_loop                  python3 tools/bench_control.py
run grid_phy                      python3 tools/bench_grid.py
run plant_mfma                    python3 tools/bench_plant.py
run host_overhead                 python3 tools/host_overhead.py
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        lines = [l for l in open(f).read().splitlines() if l.strip().startswith("{")]
        d = json.loads(lines[-1])
    except Exception as ex:
        print(os.path.basename(f), "unreadable:", ex); continue
    if "value" in d:
        r = d.get("roofline", {})
        print("%-36s %7.3f G env-steps/s  %6.2f us/step wall  kern %5.2f us  frac %s  rollout %s" % (
            os.path.basename(f), d["value"] / 1e9, d["ms_per_step"] * 1e3, r.get("kernel_avg_us", float("nan")), r.get("frac"),
            ("%.1f G" % (d["fused_rollout"]["env_steps_per_s_this_rank"] / 1e9)) if isinstance(d.get("fused_rollout"), dict) and "env_steps_per_s_this_rank" in d["fused_rollout"] else "-"))
    else:
        print(os.path.basename(f), json.dumps(d)[:300])
PY
