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
for f in conv_shapes conv_accuracy loss_cam_kernels other_configs; do grep -v amdgpu.ids $O/$f.txt > profiles/${P}_$f.txt; done
grep -v amdgpu.ids $O/conv_shapes_fp32.txt > profiles/${P}_conv_shapes_fp32_kernels.txt
