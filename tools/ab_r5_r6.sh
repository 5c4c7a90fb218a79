#!/bin/bash
# Round 5's tree against this tree on ONE box.  Prepare it first (here, before gpurun snapshots the repository; _r5tree/ is git-ignored):
#     mkdir _r5tree && git archive d6e83c1 | tar -x -C _r5tree && (cd _r5tree && python -m weaklysuperviseddl_amd._build)
# then: the training step (cfg2) four times
# each, interleaved, then cfg3 / cfg5 / CAM once each.   usage: tools/ab_r5_r6.sh  (on the GPU box, from the repository root)
mkdir -p gpurun_out/r6
run() {  # label, dir, bench args
  local label="$1" dir="$2"; shift 2
  (cd $dir && timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline "$@" 2>/dev/null) | python -c "
import sys, json
d = json.loads(sys.stdin.read())
cam = d.get('cam') or {}
print('%-10s %-28s %8.1f img/s %7.3f ms   cam %s ms/img' % ('$label', ' '.join('$*'.split()), d['value'], d['ms_per_step'], cam.get('ms_per_img')))"
}
{
for r in 1 2 3 4; do
  run r5 _r5tree --no-cam --steps 40
  run r6 . --no-cam --steps 40
done
for c in cfg3 cfg5; do
  run r5 _r5tree --no-cam --config $c --steps 10 --warmup 3
  run r6 . --no-cam --config $c --steps 10 --warmup 3
done
run r5 _r5tree --steps 10
run r6 . --steps 10
} 2>&1 | tee gpurun_out/r6/ab_r5_r6.txt
