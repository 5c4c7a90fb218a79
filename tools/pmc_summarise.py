#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/<round>_pmc_traffic.json.

    python tools/pmc_summarise.py <calib_fetch_dir> <calib_write_dir> <bench_fetch_dir> <bench_write_dir> <out.json> [<fetch_dir> <write_dir> ...]
(further fetch / write directory pairs - the CAM leg, the loss kernels - are merged in; a kernel seen in several runs keeps the
figures of the run that launched it most)
Counter unit: KiB (rocprofv3 derives FETCH_SIZE = TCC_EA0_RDREQ*64 B / 1024).  The calibration launches have known
byte counts, giving a bytes-per-counted-byte factor for our 4 B/lane and 16 B/lane access patterns."""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    f = glob.glob(d + "/*/*_counter_collection.csv")[0]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        acc[name][0] += 1
        acc[name][1] += float(r["Counter_Value"])
    return {k: {"launches": v[0], "avg_kib": v[1] / v[0]} for k, v in acc.items()}


def main():
    cf, cw, bf, bw, out = sys.argv[1:6]
    extra = sys.argv[6:]
    GiB = 1 << 30
    calf, calw = per_kernel(cf, "FETCH_SIZE"), per_kernel(cw, "WRITE_SIZE")
    factors = {
        "fetch_4B_per_lane": GiB / (calf["copy_planes_kernel"]["avg_kib"] * 1024),
        "write_4B_per_lane": GiB / (calw["copy_planes_kernel"]["avg_kib"] * 1024),
        "fetch_16B_per_lane": GiB / (calf["adam_kernel"]["avg_kib"] * 1024),
        "write_16B_per_lane": 0.75 * GiB / (calw["adam_kernel"]["avg_kib"] * 1024),
    }
    fetch, write = per_kernel(bf, "FETCH_SIZE"), per_kernel(bw, "WRITE_SIZE")
    for i in range(0, len(extra) - 1, 2):
        for dst, src in ((fetch, per_kernel(extra[i], "FETCH_SIZE")), (write, per_kernel(extra[i + 1], "WRITE_SIZE"))):
            for k, v in src.items():
                if k not in dst or v["launches"] > dst[k]["launches"]:
                    dst[k] = v
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith(("conv_", "bn_", "wgrad_", "prep_", "adam", "pairwise", "layercam", "softmax_ce", "bilinear", "dy_split", "multi_amax", "amax_", "maxpool", "gap_")):
            continue
        fk, wk = fetch.get(k, {"launches": 0, "avg_kib": 0.0}), write.get(k, {"launches": 0, "avg_kib": 0.0})
        wide = k.startswith(("adam", "bn_apply", "bn_bwd_apply", "bn_stats"))      # 16 B/lane kernels
        ff = factors["fetch_16B_per_lane" if wide else "fetch_4B_per_lane"]
        wf = factors["write_16B_per_lane" if wide else "write_4B_per_lane"]
        kernels[k] = {"launches": fk["launches"] or wk["launches"],
                      "fetch_bytes_per_launch": fk["avg_kib"] * 1024 * ff, "write_bytes_per_launch": wk["avg_kib"] * 1024 * wf,
                      "raw_fetch_kib": fk["avg_kib"], "raw_write_kib": wk["avg_kib"]}
    import datetime
    json.dump({"unit_note": "bytes = counter(KiB) * 1024 * calibration factor of the access width",
               "collected": datetime.date.today().isoformat(),
               "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/refresh_profiles.sh)",
               "calibration_factors": factors, "kernels": kernels}, open(out, "w"), indent=1)
    print(json.dumps(factors, indent=1))
    for k, v in kernels.items():
        print(f"{k:50s} launches {v['launches']:5d}  fetch {v['fetch_bytes_per_launch'] / 1e6:9.1f} MB  write {v['write_bytes_per_launch'] / 1e6:9.1f} MB")


if __name__ == "__main__":
    main()
