for i in 1 2; do
python bench.py --cam-only --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read())['cam']; print('cam-only', d['ms_per_img'], d['ms_per_img_pipelined'])"
python bench.py --cam-only --no-roofline --opt ksplit_target=320 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read())['cam']; print('cam-only ks320', d['ms_per_img'], d['ms_per_img_pipelined'])"
python bench.py --steps 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('full', d['value'], d['cam']['ms_per_img'], d['cam']['ms_per_img_pipelined'])"
python bench.py --steps 10 --no-cpu-baseline --no-roofline --opt ksplit_target=320 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('full ks320', d['value'], d['cam']['ms_per_img'], d['cam']['ms_per_img_pipelined'])"
done
