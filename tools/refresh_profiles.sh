#!/bin/bash
# Regenerates everything under profiles/ that depends on the kernels (run on the GPU box: gpurun -- tools/refresh_profiles.sh).
# Output lands in gpurun_out/refresh/; copy what is to be judged into profiles/.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
step() { echo "[refresh] $*" >&2; }
# WSDL_REFRESH_PART=A: bench, kernel stats, PMC traffic, roctx markers, MFMA busy of the step;  B: per-shape tables, 3x3 MFMA busy,
# other configs (gpurun calls are limited to 20 minutes: the whole script does not fit one call);  unset: everything
PART=${WSDL_REFRESH_PART:-AB}
if [[ $PART == *A* ]]; then

step "bench (full: roofline + cam + cpu baseline)"
python3 $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err || exit 1

step "rocprofv3 kernel stats, default (overlapped) run"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_default -- python3 $R/bench.py --no-cpu-baseline --no-cam --no-roofline --no-full-width > $O/bench_under_rocprof.txt 2>&1 || exit 1
cp $(ls $O/kt_default/*/*kernel_stats.csv | head -1) $O/bench_n1_kernel_stats.csv

step "rocprofv3 kernel stats, serial run"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_serial -- python3 $R/bench.py --no-cpu-baseline --no-cam --no-roofline --no-full-width --serial > $O/bench_serial_under_rocprof.txt 2>&1 || exit 1
cp $(ls $O/kt_serial/*/*kernel_stats.csv | head -1) $O/bench_n1_serial_kernel_stats.csv

step "PMC calibration + traffic passes"
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/cal_f -- python3 $R/tools/pmc_calibrate.py > /dev/null 2>&1 || exit 1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/cal_w -- python3 $R/tools/pmc_calibrate.py > /dev/null 2>&1 || exit 1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/b_f -- python3 $R/bench.py --no-cpu-baseline --no-cam --no-roofline --no-full-width --steps 3 --warmup 1 > /dev/null 2>&1 || exit 1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/b_w -- python3 $R/bench.py --no-cpu-baseline --no-cam --no-roofline --no-full-width --steps 3 --warmup 1 > /dev/null 2>&1 || exit 1
# the CAM leg (small-grid conv forms, layercam_* kernels) and the loss kernels (pairwise_kernel, softmax_ce_kernel ...)
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/c_f -- python3 $R/bench.py --cam-only > /dev/null 2>&1 || exit 1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/c_w -- python3 $R/bench.py --cam-only > /dev/null 2>&1 || exit 1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/l_f -- python3 $R/tools/loss_cam_bench.py > /dev/null 2>&1 || exit 1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/l_w -- python3 $R/tools/loss_cam_bench.py > /dev/null 2>&1 || exit 1
python3 $R/tools/pmc_summarise.py $O/cal_f $O/cal_w $O/b_f $O/b_w $O/pmc_traffic.json $O/l_f $O/l_w $O/c_f $O/c_w > $O/pmc_traffic.txt 2>&1 || exit 1

step "roctx ranges: marker + kernel trace of two steps and one CAM batch (WSDL_ROCTX=1)"
WSDL_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $O/markers -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-full-width --steps 2 --warmup 1 > $O/markers_bench.txt 2>&1 || exit 1
cp $(ls $O/markers/*/*marker_api_stats.csv 2>/dev/null | head -1) $O/roctx_marker_stats.csv 2>/dev/null || true

step "PMC: MFMA busy cycles of the whole step"
rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $O/b_m -- python3 $R/bench.py --no-cpu-baseline --no-cam --no-roofline --no-full-width --steps 3 --warmup 1 --serial > /dev/null 2>&1 || exit 1
python3 $R/tools/pmc_table.py $O/b_m > $O/pmc_mfma_busy.txt 2>&1

fi
if [[ $PART == *B* ]]; then
step "per-shape conv table (default fp16x2, bf16x3, fp32-MFMA kernels)"
python3 $R/tools/conv_shapes_bench.py > $O/conv_shapes.txt 2>&1 || exit 1
python3 $R/tools/conv_shapes_bench.py --opt conv_arith=0 > $O/conv_shapes_bf16x3.txt 2>&1 || exit 1
python3 $R/tools/conv_shapes_bench.py --opt conv_split=0,wgrad_split=0 > $O/conv_shapes_fp32.txt 2>&1 || exit 1
python3 $R/tools/conv_accuracy.py > $O/conv_accuracy.txt 2>&1 || exit 1

step "MFMA busy of the 3x3 convolutions, per shape"
$R/tools/mfma_busy_3x3.sh > $O/mfma_busy_3x3.txt 2> $O/mfma_busy_3x3.err || exit 1
cd /tmp

step "other BASELINE configs through bench.py, 2-rank rehearsal, bf16x3 A/B, hipGraph A/B"
for c in cfg3 cfg4 cfg5; do python3 $R/bench.py --config $c --steps 10 --warmup 3 > $O/bench_$c.json 2> $O/bench_$c.err || exit 1; done
python3 $R/bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_n2_gloo_one_gpu.json 2> $O/bench_n2.err || exit 1
WSDL_FORCE_DIST=1 python3 $R/bench.py --no-cpu-baseline --no-cam --no-roofline --no-full-width > $O/bench_n1_rccl_single_rank.json 2> /dev/null || exit 1
python3 $R/bench.py --no-cpu-baseline --no-cam --opt conv_arith=0 > $O/bench_n1_bf16x3.json 2> /dev/null || exit 1
python3 $R/bench.py --no-cpu-baseline --no-cam --graph 1 > $O/bench_n1_hipgraph.json 2> /dev/null || exit 1

step "loss / CAM kernels, other configs"
python3 $R/tools/loss_cam_bench.py > $O/loss_cam_kernels.txt 2>&1 || exit 1
python3 $R/tools/configs_bench.py > $O/other_configs.txt 2>&1 || exit 1
python3 $R/tools/bn_bench.py > $O/bn_kernels.txt 2>&1 || exit 1
fi
step done
