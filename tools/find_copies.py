"""Where do the ~39 device-to-device copies per training step come from?  torch.profiler over two steps, aten::copy_ /
aten::clone / aten::contiguous grouped by the python frames that issued them."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
model, opt, step, eager = bench.build_workload("cfg2", 16, 256, dev, 0, graph=False)
for _ in range(4):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    for _ in range(2):
        step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::fill_", "aten::zero_", "aten::add_", "aten::add"):
        st = [s for s in (ev.stack or []) if "weaklysuperviseddl_amd" in s or "bench.py" in s]
        key = (ev.name, tuple(str(x) for x in (ev.input_shapes or [])[:2]), " <- ".join(s.split("/")[-1] for s in st[:3]))
        cnt[key] += 1
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print(v / 2, k)
