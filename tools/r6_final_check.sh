#!/bin/bash
# what the driver runs at round end, in one call: the -m gpu suite, smoke(), the default bench line
mkdir -p gpurun_out/r6
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r6/t_final.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r6/t_final.log | cut -c1-250
python __graft_entry__.py smoke 2>&1 | tail -2 | cut -c1-300
python bench.py > gpurun_out/r6/bench_final.json 2> gpurun_out/r6/bench_final.err; python -c "
import json; d=json.load(open('gpurun_out/r6/bench_final.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('mfma_busy_3x3'), d['cpu_baseline']['value'], (d.get('range') or {}).get('exceeded'))"
