#!/bin/bash
# A/B of library options on the dominant conv shapes, one process per option set (same box, back to back, twice)
# usage: tools/ab_conv.sh "<shapes>" "<passes>" opt1 opt2 ...     ("-" = no options)
shapes="$1"; passes="$2"; shift 2
for round in 1 2; do
for o in "$@"; do
    oo=""; if [ "$o" != "-" ]; then oo="--opt $o"; fi
    echo "### round $round opts: $o"
    python tools/conv_shapes_bench.py --shapes "$shapes" --only "$passes" --reps 20 $oo | grep -v "^shape"
done
done
