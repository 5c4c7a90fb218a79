"""Is the CAM leg (8 x 224 x 224: ~160 small launches, host-bound in eager mode) faster as a hipGraph replay?
    python tools/cam_graph_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from weaklysuperviseddl_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
gen, imgs, cls = bench.cam_setup(dev)
for _ in range(3):
    ref = gen.generate_batch(imgs, 1.0, cls, thresh=0.3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    gen.generate_batch(imgs, 1.0, cls, thresh=0.3)
torch.cuda.synchronize()
print("eager  %.3f ms/batch" % ((time.perf_counter() - t0) / 10 * 1e3))
s_imgs, s_cls = imgs.clone(), cls.clone()
ops.reset_amax_pool(dev)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = gen.generate_batch(s_imgs, 1.0, s_cls, thresh=0.3)
ops.reset_amax_pool(dev)
g.replay()
torch.cuda.synchronize()
print("replay equals eager:", [bool(torch.equal(a, b)) for a, b in zip(out, ref)])
t0 = time.perf_counter()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
print("replay %.3f ms/batch" % ((time.perf_counter() - t0) / 10 * 1e3))
