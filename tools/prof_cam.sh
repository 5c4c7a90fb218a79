#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_cam -- python3 $R/bench.py --cam-only > $R/gpurun_out/kt_cam.json 2> $R/gpurun_out/kt_cam.err || exit 1
cp $(ls $R/gpurun_out/kt_cam/*/*kernel_stats.csv | head -1) $R/gpurun_out/kt_cam_kernel_stats.csv
