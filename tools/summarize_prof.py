#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace stats + PMC passes) into a small text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(sub, pattern):
    return sorted(glob.glob(os.path.join(root, sub, "**", pattern), recursive=True))


print("# rocprofv3 summary for", os.path.basename(root))
for f in find("trace", "*kernel_stats.csv"):
    print("\n## kernel stats (%s)" % os.path.relpath(f, root))
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            if i < 12:
                print(",".join(row))
for f in find("trace", "*kernel_trace.csv"):
    per = defaultdict(list)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            per[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"]),
                                            row.get("VGPR_Count"), row.get("SGPR_Count"),
                                            row.get("Workgroup_Size_X"), row.get("Grid_Size_X")))
    print("\n## per-kernel durations from the trace (ns)")
    for k, v in per.items():
        d = sorted(x[0] for x in v)
        print("%s: n=%d avg=%.0f median=%d min=%d max=%d vgpr=%s sgpr=%s wg=%s grid=%s"
              % (k[:80], len(d), sum(d) / len(d), d[len(d) // 2], d[0], d[-1], v[0][1], v[0][2], v[0][3], v[0][4]))
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_mem"):
    for f in find(sub, "*counter_collection.csv"):
        acc = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("\n## counters (%s) -- mean per dispatch" % sub)
        for k, cs in acc.items():
            if not any(w in k for w in ("step", "rollout", "plant", "grid", "pend")):
                continue
            for c, vals in sorted(cs.items()):
                print("%s  %s: n=%d mean=%.1f" % (k[:60], c, len(vals), sum(vals) / len(vals)))
