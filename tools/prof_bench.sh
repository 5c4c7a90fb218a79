#!/bin/bash
# usage: tools/prof_bench.sh <tag> [bench args...]   -> gpurun_out/kt_bench_<tag>/ kernel stats of one bench.py run
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_bench_${tag} -- python3 $R/bench.py --no-cpu-baseline --no-cam --no-roofline --no-full-width "$@" > $R/gpurun_out/kt_bench_${tag}.json 2> $R/gpurun_out/kt_bench_${tag}.err || exit 1
f=$(ls $R/gpurun_out/kt_bench_${tag}/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/kt_bench_${tag}_kernel_stats.csv
head -28 $f | cut -c1-160
