#!/bin/bash
mkdir -p gpurun_out/r6
python tools/stem_wgrad_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/stem_wgrad.txt
timeout -k 10 400 python -m pytest tests/test_hip_ops.py tests/test_hip_plan.py -m gpu -q -x -k "conv_fwd_dgrad_wgrad or bn_backward_writes or plan or channel_scales" > gpurun_out/r6/t_part5.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r6/t_part5.log | cut -c1-250
step() {
  local label="$1"; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-cam --no-roofline --steps 40 $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-40s %8.1f img/s %7.3f ms' % ('$label', d['value'], d['ms_per_step']))"
}
{
for r in 1 2 3; do
  step "default" A=1
  step "chan amax off" WSDL_CHAN_AMAX=0
  EXTRA="--opt wgrad_min_tiles=1" step "wgrad_min_tiles=1" A=1
  EXTRA="--opt stem_wgrad=0" step "stem wgrad generic" A=1
  step "presplit off" WSDL_DY_PRESPLIT=0
done
} > gpurun_out/r6/ab_misc2.txt 2>&1; cat gpurun_out/r6/ab_misc2.txt
python tools/conv_shapes_bench.py --only wgrad --reps 20 --shapes "l3.conv2,l4.0.conv2,l4.conv2,aspp d12,l4.conv3,l3.conv3" > gpurun_out/r6/wgrad_plain2.txt 2>&1
python tools/conv_shapes_bench.py --only wgrad --reps 20 --camax x,dy --shapes "l3.conv2,l4.0.conv2,l4.conv2,aspp d12,l4.conv3,l3.conv3" > gpurun_out/r6/wgrad_camax2.txt 2>&1
paste <(cut -c1-44 gpurun_out/r6/wgrad_plain2.txt) <(cut -c32-44 gpurun_out/r6/wgrad_camax2.txt) | grep -v amdgpu
