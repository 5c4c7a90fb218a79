#!/bin/bash
mkdir -p gpurun_out/r6
timeout -k 10 600 python -m pytest tests/test_hip_plan.py tests/test_hip_dp.py tests/test_hip_ops.py -m gpu -q -x -k "deferred or batchnorm or bn_ or plan" > gpurun_out/r6/t_part7.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r6/t_part7.log | cut -c1-250
python tools/bn_bench.py > gpurun_out/r6/bn_bench.txt 2>&1; grep -v amdgpu gpurun_out/r6/bn_bench.txt | tail -12
for r in 1 2; do
timeout -k 10 200 python bench.py --no-cpu-baseline --no-cam --no-roofline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default %8.1f img/s %7.3f ms' % (d['value'], d['ms_per_step']))"
done
