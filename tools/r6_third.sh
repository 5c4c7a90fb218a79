#!/bin/bash
# round 6, third GPU pass: quick checks of the new pieces, then A/Bs (no full suite)
mkdir -p gpurun_out/r6
timeout -k 10 500 python -m pytest tests/test_hip_ops.py tests/test_hip_plan.py tests/test_hip_dp.py -m gpu -q -x -k "conv_fwd_dgrad_wgrad or bn_backward_writes or plan or replayed or disagree" > gpurun_out/r6/t_part4.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r6/t_part4.log | cut -c1-250
step() {  # label, env..., bench args after --
  local label="$1"; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-cam --no-roofline --steps 40 $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-40s %8.1f img/s %7.3f ms' % ('$label', d['value'], d['ms_per_step']))"
}
E=weaklysuperviseddl_amd/csrc/exp
{
for r in 1 2; do
  step "in-tree (rowA + pairB)" A=1
  step "oldlds" WSDL_LIB=$E/libwsdl_oldlds.so
  step "rowA only (no pairB)" WSDL_LIB=$E/libwsdl_nopairb.so
  step "pairB only (no rowA)" WSDL_LIB=$E/libwsdl_norowa.so
done
} > gpurun_out/r6/ab_lds2.txt 2>&1; cat gpurun_out/r6/ab_lds2.txt
{
for r in 1 2; do
  step "defer 48MB groups" WSDL_WGRAD_DEFER_MB=48
  step "defer 16MB groups" WSDL_WGRAD_DEFER_MB=16
  step "defer 128MB groups" WSDL_WGRAD_DEFER_MB=128
  step "defer all at end" WSDL_WGRAD_DEFER_MB=100000
  step "no deferral" WSDL_WGRAD_DEFER=0
done
} > gpurun_out/r6/ab_defer2.txt 2>&1; cat gpurun_out/r6/ab_defer2.txt
{
for r in 1 2; do
  step "chan amax on" WSDL_CHAN_AMAX=1
  step "chan amax off" WSDL_CHAN_AMAX=0
  EXTRA="--opt stem_wgrad=0" step "stem wgrad generic" A=1
  EXTRA="--opt wgrad_min_tiles=1" step "wgrad_min_tiles=1" A=1
done
} > gpurun_out/r6/ab_misc.txt 2>&1; cat gpurun_out/r6/ab_misc.txt
python tools/conv_shapes_bench.py --only wgrad --reps 20 > gpurun_out/r6/wgrad_plain.txt 2>&1
python tools/conv_shapes_bench.py --only wgrad --reps 20 --camax x,dy > gpurun_out/r6/wgrad_camax_xdy.txt 2>&1
python tools/conv_shapes_bench.py --only wgrad --reps 20 --camax dy > gpurun_out/r6/wgrad_camax_dy.txt 2>&1
paste <(cut -c1-44 gpurun_out/r6/wgrad_plain.txt) <(cut -c32-44 gpurun_out/r6/wgrad_camax_xdy.txt) <(cut -c32-44 gpurun_out/r6/wgrad_camax_dy.txt) | grep -v amdgpu
