#!/usr/bin/env python3
"""Forward / backward / optimiser wall time of one training step (B=16, 256x256), device-synchronised phases."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from weaklysuperviseddl_amd import ops  # noqa: E402
from weaklysuperviseddl_amd.TraditionalModel import build_segmentation_model  # noqa: E402
from weaklysuperviseddl_amd.TraditionalModel.SegmentationModel import make_optimizer  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_segmentation_model().to(dev).train()
opt = make_optimizer(model)
img, masks = bench.synthetic_batch(16, 256, 256, dev, 1)


def phase(fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    return r, (time.perf_counter() - t) * 1e3


tf = tb = to = 0.0
for it in range(8):
    out, f = phase(lambda: ops.cross_entropy(model(img)["out"], masks))
    opt.zero_grad()
    _, b = phase(lambda: out.backward())
    _, o = phase(opt.step)
    if it >= 3:
        tf, tb, to = tf + f, tb + b, to + o
print(f"forward+CE {tf / 5:.2f} ms   backward {tb / 5:.2f} ms   adam(+prefetch) {to / 5:.2f} ms")
for overlap in (False, True):
    ops.OVERLAP_WGRAD[0] = overlap
    tb = 0.0
    for it in range(6):
        out = ops.cross_entropy(model(img)["out"], masks)
        opt.zero_grad()
        _, b = phase(lambda: out.backward())
        opt.step()
        if it >= 2:
            tb += b
    print(f"backward with wgrad overlap={overlap}: {tb / 4:.2f} ms")
