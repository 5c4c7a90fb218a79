"""What does the host pay for one replay of the training step's launch plan, and where?  (round 5)
usage: python tools/plan_probe.py [B] [S]"""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weaklysuperviseddl_amd import ops, plan
from weaklysuperviseddl_amd._lib import lib, check
from weaklysuperviseddl_amd.TraditionalModel import build_segmentation_model, train_step
from weaklysuperviseddl_amd.TraditionalModel.SegmentationModel import make_optimizer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_segmentation_model().to(dev).train()
opt = make_optimizer(model, lr=1e-4)
img = torch.randn(B, 3, S, S, device=dev)
masks = (torch.rand(B, S, S, device=dev) > 0.5).long()
for _ in range(4):
    train_step(model, opt, img, masks)
torch.cuda.synchronize()
st = next(iter(opt._wsdl_planned.values()))
print("disabled:", st.disabled, "stats:", st.plan.stats)
names = ["kernel", "memset", "stream_wait", "event_record", "event_wait", "mark"]
for label, idle in (("GPU idle at the start", True), ("GPU busy (back to back)", False)):
    us, n = (C.c_double * 6)(), (C.c_longlong * 6)()
    tot = [0.0] * 6
    wall = 0.0
    reps = 5
    for _ in range(reps):
        if idle:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        check(lib().wsdl_plan_replay_timed(st.plan.handle, us, n))
        wall += time.perf_counter() - t0
        for k in range(6):
            tot[k] += us[k]
    torch.cuda.synchronize()
    print(f"{label}: replay {wall / reps * 1e3:.2f} ms;", ", ".join(f"{names[k]} {n[k]} x {tot[k] / reps / max(n[k], 1):.2f} us" for k in range(5)))
# untimed replays, back to back
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    st.plan.replay()
issue = (time.perf_counter() - t0) / 10 * 1e3
torch.cuda.synchronize()
print(f"plain replay: host {issue:.2f} ms/step, step {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms")
