"""Summarise tools/mfma_busy_3x3.sh: per (3x3 shape, pass) the MFMA-busy fraction of the MFMA kernel of that pass.
busy = SQ_VALU_MFMA_BUSY_CYCLES (summed over the chip's SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)."""
import collections
import csv
import glob
import sys

O = sys.argv[1]
print(f"{'shape':22s} {'pass':6s} {'kernel':58s} {'launches':>8s} {'MFMA busy':>10s} {'cycles/launch':>14s}")
tot = collections.defaultdict(lambda: [0.0, 0.0])
lines = [l.split() for l in open(O + "/index.txt")]
grouped = {ps for name, ps, _g in lines if name == "aspp_4branches"}       # passes the step runs as one launch for all ASPP branches
for name, ps, geo in lines:
    tag = name.replace("/", "_").replace(".", "_")
    f = glob.glob(f"{O}/{tag}_{ps}/**/*_counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        c = acc[k][r["Counter_Name"]]
        c[0] += 1
        c[1] += float(r["Counter_Value"])
    # the pass's MFMA kernel = the conv kernel with the most MFMA-busy cycles
    best = max((k for k in acc if k.startswith("conv_") and "reduce" not in k), key=lambda k: acc[k]["SQ_VALU_MFMA_BUSY_CYCLES"][1])
    n, busy = acc[best]["SQ_VALU_MFMA_BUSY_CYCLES"]
    _, gui = acc[best]["GRBM_GUI_ACTIVE"]
    simd_cycles = gui / 8.0 * 1024.0
    single = name in ("aspp_d12", "aspp_d24", "aspp_d36") and ps in grouped
    print(f"{name:22s} {ps:6s} {best:58s} {n:8d} {busy / simd_cycles:10.3f} {gui / 8.0 / n:14.0f}"
          + ("   (single launch: not in the step, not in the totals)" if single else ""))
    if single:
        continue
    tot[ps][0] += busy
    tot[ps][1] += simd_cycles
    tot["all"][0] += busy
    tot["all"][1] += simd_cycles
for ps, (b, c) in tot.items():
    print(f"{'3x3 convolutions':22s} {ps:6s} {'(cycle-weighted over the shapes above, one launch each)':58s} {'':8s} {b / c:10.3f}")
