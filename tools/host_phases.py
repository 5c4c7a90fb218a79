"""Host time of the phases of the training step (no device synchronisation inside the step): where the issue time goes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from weaklysuperviseddl_amd import ops
dev = torch.device("cuda", 0)
model, opt, step, eager = bench.build_workload("cfg2", 16, 256, dev, 0, graph=False)
img, masks = bench.synthetic_batch(16, 256, 256, dev, 1)
for _ in range(5):
    step()
torch.cuda.synchronize()
N = 10
acc = [0.0] * 5
for _ in range(N):
    t0 = time.perf_counter()
    m = torch.clamp(masks, max=1)
    out = model(img)["out"]
    t1 = time.perf_counter()
    loss = ops.cross_entropy(out, m.long())
    t2 = time.perf_counter()
    opt.zero_grad()
    t3 = time.perf_counter()
    loss.backward()
    t4 = time.perf_counter()
    opt.step()
    t5 = time.perf_counter()
    for i, (a, b) in enumerate(((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5))):
        acc[i] += b - a
torch.cuda.synchronize()
print("host ms/step: forward %.2f  loss %.2f  zero_grad %.2f  backward %.2f  optimizer+relayout %.2f  total %.2f" %
      tuple([a / N * 1e3 for a in acc] + [sum(acc) / N * 1e3]))
# the autograd engine's own share: python backward functions are timed through a hook on ops
import cProfile, pstats, io, threading
