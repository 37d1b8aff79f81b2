#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_driver_cmd.sh: the kernel-stats table, and -- because the driver's command
also launches the step kernel in its warm-up steps and in the secondaries -- the average of the TIMED launches alone
(launches 6..25 after each reset kernel in windows of exactly warm-up + steps launches), which is what bench.py's
roofline.kernel_avg_us measures with HIP events."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
W, K = 5, 20


def find(sub, pattern):
    return sorted(glob.glob(os.path.join(root, sub, "**", pattern), recursive=True))


print("# rocprofv3 summary of the driver's command: python3 bench.py --gpus 1 --steps %d --warmup %d" % (K, W))
for f in find("trace", "*kernel_stats.csv"):
    print("\n## kernel stats (%s)" % os.path.relpath(f, root))
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            if i < 8:
                print(",".join(c[:110] for c in row))
for f in find("trace", "*kernel_trace.csv"):
    rows = []
    with open(f) as fh:
        for row in csv.DictReader(fh):
            rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"]))
    rows.sort()
    rows = [r for r in rows if "(anonymous namespace)::" in r[2]]     # this library's kernels only (no copy / fill blits)
    timed, warm, other, gaps = [], [], [], []
    i = 0
    while i < len(rows):
        if "ct_reset_sfx_kernel" in rows[i][2]:
            j = i + 1
            while j < len(rows) and "ct_step_sfx_kernel" in rows[j][2]:
                j += 1
            seg = rows[i + 1:j]
            if len(seg) == W + K:
                warm += [b - a for a, b, _ in seg[:W]]
                timed += [b - a for a, b, _ in seg[W:]]
                gaps += [seg[q + 1][0] - seg[q][1] for q in range(W, W + K - 1)]
            else:
                other += [b - a for a, b, _ in seg]
            i = j
        else:
            if "ct_step_sfx_kernel" in rows[i][2]:
                other.append(rows[i][1] - rows[i][0])
            i += 1

    def desc(name, d):
        if not d:
            return
        d = sorted(d)
        print("%-44s n=%-6d avg=%8.1f median=%6d min=%6d p90=%6d max=%6d ns"
              % (name, len(d), sum(d) / len(d), d[len(d) // 2], d[0], d[int(len(d) * 0.9)], d[-1]))
    print("\n## ct_step_sfx_kernel launches by role (from the kernel trace)")
    desc("timed launches (steps 5..24 after a reset)", timed)
    desc("warm-up launches (steps 0..4 after a reset)", warm)
    desc("other launches (secondaries)", other)
    desc("gap between consecutive timed launches", gaps)
    if timed:
        per = [0.0] * K
        for q, v in enumerate(timed):
            per[q % K] += v
        nwin = len(timed) // K
        print("avg duration by step index after the reset (us):",
              " ".join("%d:%.2f" % (W + q, per[q] / nwin / 1e3) for q in range(K)))
        print("timed launches avg + avg gap = %.2f us per step on the stream" % ((sum(timed) / len(timed) + (sum(gaps) / max(len(gaps), 1))) / 1e3))
for sub in ("pmc_fetch", "pmc_write"):
    for f in find(sub, "*counter_collection.csv"):
        acc = defaultdict(list)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "ct_step_sfx_kernel" in row["Kernel_Name"]:
                    acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for c, vals in sorted(acc.items()):
            print("\n## %s per ct_step_sfx_kernel launch: n=%d mean=%.1f KB" % (c, len(vals), sum(vals) / len(vals)))
for name in ("bench_trace.json",):
    try:
        with open(os.path.join(root, name)) as fh:
            d = json.loads([l for l in fh if l.strip().startswith("{")][-1])
        r = d["roofline"]
        print("\n## bench.py's own line under the profiler: value %.3g env-steps/s, %.2f us/step wall, kernel_avg_us %.2f (HIP events), "
              "frac %.3f, repeats %d" % (d["value"], d["ms_per_step"] * 1e3, r["kernel_avg_us"], r["frac"], d["repeats"]))
    except Exception as exc:
        print("\n## bench line under the profiler not parsed:", exc)
if len(sys.argv) > 2 and os.path.exists(sys.argv[2]):
    with open(sys.argv[2]) as fh:
        d = json.loads([l for l in fh if l.strip().startswith("{")][-1])
    r = d["roofline"]
    print("## the same command WITHOUT the profiler:      value %.3g env-steps/s, %.2f us/step wall, kernel_avg_us %.2f (HIP events), "
          "frac %.3f, frac_moved %s, repeats %d" % (d["value"], d["ms_per_step"] * 1e3, r["kernel_avg_us"], r["frac"], r["frac_moved"], d["repeats"]))
