#!/bin/bash
# A/B of two builds of the library on one box: the in-tree build against a variant (tools/build_variant.sh), per kernel on the
# dominant shapes and on the training step, each twice, interleaved.   usage: tools/ab_lib.sh <variant .so> "<shapes>" "<passes>"
var="$1"; shapes="$2"; passes="$3"
for round in 1 2; do
  for lib in "" "$var"; do
    echo "### round $round lib: ${lib:-in-tree}"
    WSDL_LIB=$lib python tools/conv_shapes_bench.py --shapes "$shapes" --only "$passes" --reps 20 | grep -v "^shape"
  done
done
for round in 1 2 3; do
  for lib in "" "$var"; do
    WSDL_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-cam --no-roofline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step %-40s %8.1f img/s %7.3f ms' % ('${lib:-in-tree}', d['value'], d['ms_per_step']))" || exit 1
  done
done
