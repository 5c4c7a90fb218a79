python -m pytest tests/test_hip_ops.py -q -k "pairwise or affinit or cache or golden" 2>&1 | tail -3
python -m pytest tests/test_hip_fullsize.py -q -k "pairwise or cfg3 or cfg5" 2>&1 | tail -3
python -m pytest tests/test_hip_models.py -q -k "refine or ncut" 2>&1 | tail -3
python - <<PY
import torch, sys
sys.path.insert(0,'.')
import bench
d=torch.device("cuda:0")
print("ncut cfg3 fwd+bwd:", bench.ncut_bench(d)["us_fwd_bwd"], "us")
print("ncut cfg5 fwd+bwd:", bench.ncut_bench(d,8,512,512)["us_fwd_bwd"], "us")
PY
