#!/bin/bash
# copies gpurun_out/refresh/* (tools/refresh_profiles.sh) into profiles/ under this round's prefix
P=${1:-r01}
O=gpurun_out/refresh
cp $O/bench_n1.json profiles/${P}_bench_n1.json
cp $O/bench_n1_kernel_stats.csv profiles/${P}_bench_n1_kernel_stats.csv
cp $O/bench_n1_serial_kernel_stats.csv profiles/${P}_bench_n1_serial_kernel_stats.csv
grep -h "^{\|ms/step" $O/bench_under_rocprof.txt > profiles/${P}_bench_n1_under_rocprof.txt
grep -h "^{\|ms/step" $O/bench_serial_under_rocprof.txt > profiles/${P}_bench_n1_serial_under_rocprof.txt
cp $O/pmc_traffic.json profiles/${P}_pmc_traffic.json
cp $O/pmc_mfma_busy.txt profiles/${P}_pmc_mfma_busy.txt
[ -f $O/roctx_marker_stats.csv ] && cp $O/roctx_marker_stats.csv profiles/${P}_roctx_marker_stats.csv
for f in conv_shapes conv_shapes_bf16x3 conv_accuracy loss_cam_kernels other_configs mfma_busy_3x3 bn_kernels; do grep -v amdgpu.ids $O/$f.txt > profiles/${P}_$f.txt; done
for f in bench_cfg3 bench_cfg4 bench_cfg5 bench_n2_gloo_one_gpu bench_n1_bf16x3 bench_n1_hipgraph bench_n1_rccl_single_rank; do cp $O/$f.json profiles/${P}_$f.json; done
grep -v amdgpu.ids $O/conv_shapes_fp32.txt > profiles/${P}_conv_shapes_fp32_kernels.txt
