#!/bin/bash
# round 6, first GPU pass: whole -m gpu suite, then same-box A/Bs of (a) the LDS store mappings (b) image-major tile order (c) deferred slab reductions
mkdir -p gpurun_out/r6
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r6/t_all1.log 2>&1
rc=$?
tail -4 gpurun_out/r6/t_all1.log | cut -c1-300
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 400 bash tools/ab_lib.sh weaklysuperviseddl_amd/csrc/exp/libwsdl_oldlds.so "l3.conv2 d2,l4.conv2 d4,l4.conv3,l4.0.ds,l2.conv2,l3.conv1,aux" "fwd,dgrad" > gpurun_out/r6/ab_lds.txt 2>&1
cut -c1-110 gpurun_out/r6/ab_lds.txt
python tools/aspp_group_bench.py > gpurun_out/r6/aspp_imgmajor_on.txt 2>&1; tail -8 gpurun_out/r6/aspp_imgmajor_on.txt
python tools/aspp_group_bench.py --opt tile_img_major=0 > gpurun_out/r6/aspp_imgmajor_off.txt 2>&1; tail -8 gpurun_out/r6/aspp_imgmajor_off.txt
timeout -k 10 300 bash tools/ab_step.sh - tile_img_major=0 > gpurun_out/r6/ab_imgmajor_step.txt 2>&1; cat gpurun_out/r6/ab_imgmajor_step.txt
for r in 1 2 3; do for d in 1 0; do
  WSDL_WGRAD_DEFER=$d timeout -k 10 200 python bench.py --no-cpu-baseline --no-cam --no-roofline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('defer=$d %8.1f img/s %7.3f ms' % (d['value'], d['ms_per_step']))"
done; done > gpurun_out/r6/ab_defer_step.txt 2>&1; cat gpurun_out/r6/ab_defer_step.txt
