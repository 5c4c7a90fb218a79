#!/bin/bash
# round 6, second GPU pass: whole -m gpu suite, then same-box A/Bs
mkdir -p gpurun_out/r6
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r6/t_all3.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r6/t_all3.log | cut -c1-250
if grep -q "failed\|error" <(tail -1 gpurun_out/r6/t_all3.log); then exit 1; fi
timeout -k 10 400 bash tools/ab_lib.sh weaklysuperviseddl_amd/csrc/exp/libwsdl_oldlds.so "l3.conv2 d2,l4.conv2 d4,l4.conv3,l4.0.ds,l2.conv2,l3.conv1,aux" "fwd,dgrad" > gpurun_out/r6/ab_lds.txt 2>&1
cut -c1-110 gpurun_out/r6/ab_lds.txt
python tools/aspp_group_bench.py > gpurun_out/r6/aspp_imgmajor_on.txt 2>&1; tail -8 gpurun_out/r6/aspp_imgmajor_on.txt
python tools/aspp_group_bench.py --opt tile_img_major=0 > gpurun_out/r6/aspp_imgmajor_off.txt 2>&1; tail -8 gpurun_out/r6/aspp_imgmajor_off.txt
ab() {  # name, env assignment A, env assignment B
  for r in 1 2 3; do for e in "$2" "$3"; do
    env $e timeout -k 10 200 python bench.py --no-cpu-baseline --no-cam --no-roofline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s %8.1f img/s %7.3f ms' % ('$e', d['value'], d['ms_per_step']))"
  done; done > gpurun_out/r6/ab_$1.txt 2>&1; cat gpurun_out/r6/ab_$1.txt
}
timeout -k 10 300 bash tools/ab_step.sh - tile_img_major=0 > gpurun_out/r6/ab_imgmajor_step.txt 2>&1; cat gpurun_out/r6/ab_imgmajor_step.txt
ab defer WSDL_WGRAD_DEFER=1 WSDL_WGRAD_DEFER=0
ab presplit WSDL_DY_PRESPLIT=1 WSDL_DY_PRESPLIT=0
ab chanamax WSDL_CHAN_AMAX=1 WSDL_CHAN_AMAX=0
