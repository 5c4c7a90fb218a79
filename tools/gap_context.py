#!/usr/bin/env python3
"""Kernels around the largest idle gaps of one training step:  python tools/gap_context.py <kernel_trace.csv> [n_gaps]
(the step before the last adam_kernel; for each gap: the last kernels that ended before it and the first that start after)"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:70], r["Queue_Id"]))
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
k = len(adam) - 2
seg = rows[adam[k - 1] + 1: adam[k] + 1]
t0 = seg[0][0]
ends = sorted(seg, key=lambda r: r[1])
gaps = []
cur_end = seg[0][1]
for r in seg[1:]:
    if r[0] > cur_end:
        gaps.append((r[0] - cur_end, cur_end, r[0]))
    cur_end = max(cur_end, r[1])
gaps.sort(reverse=True)
for g, a, b in gaps[:int(sys.argv[2]) if len(sys.argv) > 2 else 4]:
    print(f"--- gap {g / 1e3:.0f} us at {(a - t0) / 1e6:.2f} ms")
    before = [r for r in seg if r[1] <= a][-4:]
    after = [r for r in seg if r[0] >= b][:4]
    for r in before:
        print(f"   before  q{r[3]}  {(r[0] - t0) / 1e6:8.3f} .. {(r[1] - t0) / 1e6:8.3f} ms  {r[2]}")
    for r in after:
        print(f"   after   q{r[3]}  {(r[0] - t0) / 1e6:8.3f} .. {(r[1] - t0) / 1e6:8.3f} ms  {r[2]}")
