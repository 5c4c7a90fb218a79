python -m pytest tests/test_hip_ops.py -q -x -k "conv_fwd_dgrad_wgrad or amax or scales" 2>&1 | tail -2
python -m pytest tests/test_hip_models.py -q -x -k "layercam or classifier or cam_batches or pseudo" 2>&1 | tail -2
for o in "" "ksplit_target=512" "ksplit_target=512,ksplit_max=16" "ksplit_target=768,ksplit_max=16,ksplit_min_chunks=3" "ksplit_target=512,ksplit_min_chunks=2,ksplit_max=12" "ksplit_target=256" ; do
  echo "### opts: $o"
  if [ -z "$o" ]; then python bench.py --cam-only --no-roofline 2>/dev/null; else python bench.py --cam-only --no-roofline --opt "$o" 2>/dev/null; fi | python -c "import json,sys; d=json.loads(sys.stdin.read())['cam']; print(d['ms_per_img'], d['ms_per_img_pipelined'])"
done
