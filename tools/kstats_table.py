"""Per-step kernel time table from a rocprofv3 kernel_stats.csv:  python tools/kstats_table.py <csv> <steps> [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per step: {tot / 1e6 / steps:.3f} ms ({steps} steps)")
for r in rows[:top]:
    n = r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    print(f"{n[:72]:72s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms/step  avg {float(r['AverageNs']) / 1e3:8.1f} us")
