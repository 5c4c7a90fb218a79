#!/bin/bash
mkdir -p gpurun_out/r6
python tools/stem_wgrad_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/stem_wgrad3.txt
python tools/aspp_group_bench.py 2>&1 | grep -E "^forward|^input" | tee gpurun_out/r6/aspp_default.txt
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r6/t_all4.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r6/t_all4.log | cut -c1-250
for r in 1 2 3; do
timeout -k 10 200 python bench.py --no-cpu-baseline --no-cam --no-roofline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default %8.1f img/s %7.3f ms' % (d['value'], d['ms_per_step']))"
done | tee gpurun_out/r6/bench_default.txt
