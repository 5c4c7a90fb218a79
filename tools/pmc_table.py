"""Print per-kernel averages of every counter found in rocprofv3 --pmc output directories.

    python tools/pmc_table.py <dir> [<dir> ...] [--match conv_]
"""
import collections
import csv
import glob
import sys


def main():
    dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = "conv_"
    if "--match" in sys.argv:
        match = sys.argv[sys.argv.index("--match") + 1]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for d in dirs:
        for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
                if match not in name:
                    continue
                c = acc[name][r["Counter_Name"]]
                c[0] += 1
                c[1] += float(r["Counter_Value"])
    for k, cs in acc.items():
        print(k)
        for c, (n, v) in sorted(cs.items()):
            print(f"    {c:34s} {v / n:16.1f}   (avg of {n})")


if __name__ == "__main__":
    main()
