#!/bin/bash
# usage: tools/ktrace_one.sh <tag> <one_conv args...>  -> per-kernel average durations of one tools/one_conv.py run
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_${tag} -- python3 $R/tools/one_conv.py "$@" > /dev/null 2>&1 || exit 1
f=$(ls $R/gpurun_out/kt_${tag}/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    if any(k in n for k in ("conv_", "wgrad", "dy_split", "prep_")):
        print(f"  {n:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
