#!/bin/bash
mkdir -p gpurun_out/r6
for f in 0 1 2; do echo "### bn_bwd_form=$f"; python tools/bn_bench.py --opt bn_bwd_form=$f 2>&1 | grep -v amdgpu | tail -6; done | tee gpurun_out/r6/bn_bwd_form.txt
for r in 1 2; do for f in 0 1 2; do
timeout -k 10 200 python bench.py --no-cpu-baseline --no-cam --no-roofline --steps 40 --opt bn_bwd_form=$f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bn_bwd_form=$f %8.1f img/s %7.3f ms' % (d['value'], d['ms_per_step']))"
done; done | tee -a gpurun_out/r6/bn_bwd_form.txt
