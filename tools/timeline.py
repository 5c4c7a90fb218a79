"""Busy / idle analysis of one training step from a rocprofv3 kernel_trace.csv:
    python tools/timeline.py <kernel_trace.csv> [--steps 13]
Splits the trace at the adam_kernel launches (one per step) and reports, for the last steps: wall time, union of
kernel-busy time, time with >= 2 kernels in flight (stream overlap), the largest idle gaps."""
import csv
import sys

rows = []
for f in [a for a in sys.argv[1:] if a.endswith(".csv")]:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
print(f"{len(rows)} kernels, {len(adam)} steps")
for k in range(max(1, len(adam) - 3), len(adam)):
    seg = rows[adam[k - 1] + 1: adam[k] + 1]
    t0, t1 = seg[0][0], max(r[1] for r in seg)
    ev = sorted([(r[0], 1) for r in seg] + [(r[1], -1) for r in seg])
    busy = over = 0
    depth, last = 0, t0
    gaps = []
    for t, d in ev:
        if depth >= 1:
            busy += t - last
        elif t > last:
            gaps.append((t - last, last - t0))
        if depth >= 2:
            over += t - last
        depth += d
        last = t
    gaps.sort(reverse=True)
    print(f"step {k}: wall {(t1 - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, >=2 kernels in flight {over / 1e6:.3f} ms, "
          f"idle {(t1 - t0 - busy) / 1e6:.3f} ms in {len(gaps)} gaps; largest (us @ms): " +
          ", ".join(f"{g / 1e3:.0f}@{o / 1e6:.1f}" for g, o in gaps[:6]))
    queues = {}
    for r in seg:
        queues[r[3]] = queues.get(r[3], 0) + (r[1] - r[0])
    print("   kernel time per queue (ms):", {q: round(v / 1e6, 2) for q, v in queues.items()})

if "--tail" in sys.argv:
    # the last kernels of a step (what Adam waits for) and the first ones of the next: queue, start (us before / after the
    # adam launch), duration
    n = int(sys.argv[sys.argv.index("--tail") + 1])
    k = len(adam) - 2
    a = adam[k]
    t_adam = rows[a][0]
    for r in rows[max(0, a - n): a + n]:
        nm = r[2].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:58]
        print(f"  q{r[3]:>2} {(r[0] - t_adam) / 1e3:9.1f} us  {(r[1] - r[0]) / 1e3:7.1f} us  {nm}")
