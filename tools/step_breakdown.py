"""Kernel time and launches per training step by category, from a `rocprofv3 --kernel-trace --stats` kernel_stats.csv of
`bench.py [--serial]` (profiles/r*_bench_n1[_serial]_kernel_stats.csv).  Steps = Adam launches."""
import csv
import sys


def category(n):
    if "wgrad_reduce" in n or "dy_split" in n or "transpose_add" in n or "channel_amax" in n:
        return "weight-gradient satellites (dY pre-split, slab reduces)"
    if "wgrad" in n:
        return "weight gradients"
    if "conv_igemm" in n or "splitk" in n:
        return "convolution forward + dgrad"
    if "::bn_" in n and "fold" not in n:
        return "BatchNorm"
    if "prep_weights" in n or "amax_kernel" in n:
        return "weight layouts + amax"
    if "adam" in n:
        return "Adam"
    if "at::native" in n or "at::" in n:
        return "torch's own kernels"
    return "other kernels of the library"


def main(path):
    rows = list(csv.DictReader(open(path)))
    steps = sum(int(r["Calls"]) for r in rows if "adam_kernel" in r["Name"])
    cat, tot, launches = {}, 0.0, 0
    for r in rows:
        t, k = float(r["TotalDurationNs"]) / 1e6, int(r["Calls"])
        e = cat.setdefault(category(r["Name"]), [0.0, 0])
        e[0] += t
        e[1] += k
        tot += t
        launches += k
    print(f"{path}: {steps} steps")
    print(f"{'category':58s} {'ms/step':>8s} {'launches/step':>14s}")
    for k, v in sorted(cat.items(), key=lambda x: -x[1][0]):
        print(f"{k:58s} {v[0] / steps:8.3f} {v[1] / steps:14.1f}")
    print(f"{'all kernels':58s} {tot / steps:8.3f} {launches / steps:14.1f}")


if __name__ == "__main__":
    for p in sys.argv[1:]:
        main(p)
