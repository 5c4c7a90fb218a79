#!/bin/bash
# Step-level A/B of library option sets on one box: every set twice, interleaved.  Kernel-level sweeps and the step do not
# always agree (profiles/r02_notes.md).   usage: tools/ab_step.sh opts1 opts2 ...     ("-" = defaults)
for round in 1 2; do
for o in "$@"; do
    oo=""; if [ "$o" != "-" ]; then oo="--opt $o"; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-cam --no-roofline --steps 40 $oo 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-44s %8.1f img/s %7.3f ms' % ('$o', d['value'], d['ms_per_step']))" || exit 1
done
done
