#!/bin/bash
# Step-level A/B of environment switches on one box (each twice, interleaved).  usage: tools/ab_env.sh "A=1" "WSDL_X=0" ...
for round in 1 2; do
for e in "$@"; do
    env $e timeout -k 10 200 python bench.py --no-cpu-baseline --no-cam --no-roofline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-36s %8.1f img/s %7.3f ms' % ('$e', d['value'], d['ms_per_step']))" || exit 1
done
done
