#!/usr/bin/env python3
"""Per-shape durations of the BatchNorm kernels from a rocprofv3 kernel trace:  python tools/bn_trace_table.py <kernel_trace.csv>
Groups the dispatches of every bn_* kernel by grid size (= channel count for the resident kernels) and prints launches,
median microseconds and the share of the trace's BatchNorm time."""
import collections
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "bn_" not in n:
        continue
    name = n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    grid = (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r.get("Grid_Size_Y", 1)))
    acc[(name, grid, int(r["Workgroup_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in acc.values())
print(f"{'kernel':34s} {'grid (wg x, y)':>16s} {'threads':>7s} {'n':>5s} {'median us':>10s} {'total ms':>9s} {'share':>6s}")
for (name, grid, nt), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{name:34s} {str(grid):>16s} {nt:7d} {len(v):5d} {statistics.median(v):10.1f} {sum(v) / 1e3:9.3f} {sum(v) / tot:6.1%}")
