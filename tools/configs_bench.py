#!/usr/bin/env python3
"""One-GPU timing of the BASELINE.json configs that are parity-test cases rather than bench lines:
cfg3 (B=32, 256x256, CE + 0.1*NCut on the logits) and the per-GPU share of cfg5 (B=8, 512x512, CE + 0.1*NCut +
0.1*Boundary), plus the batched refinement.  Prints ms/step and img/s."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from weaklysuperviseddl_amd import ops  # noqa: E402
from weaklysuperviseddl_amd.TraditionalModel import (build_segmentation_model, train_step, LocalNormalizedCutLoss,  # noqa: E402
                                                     ConstrainToBoundaryLossSingle, refine_pseudo_masks_batched)
from weaklysuperviseddl_amd.TraditionalModel.SegmentationModel import make_optimizer  # noqa: E402

dev = torch.device("cuda:0")


def run(B, S, extra, steps=12, warm=4):
    torch.manual_seed(0)
    model = build_segmentation_model().to(dev).train()
    opt = make_optimizer(model)
    img, masks = bench.synthetic_batch(B, S, S, dev, 1)
    for _ in range(warm):
        train_step(model, opt, img, masks, extra)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        loss = train_step(model, opt, img, masks, extra)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    return dt * 1e3, B / dt, loss.item()


ncut = LocalNormalizedCutLoss(0.1, 5)
bnd = ConstrainToBoundaryLossSingle(0.1, 5, 5)
print("cfg2  B=16 256^2 CE                      : %.2f ms/step  %.1f img/s  loss %.4f" % run(16, 256, None))
print("cfg3  B=32 256^2 CE + 0.1 NCut           : %.2f ms/step  %.1f img/s  loss %.4f" %
      run(32, 256, lambda o, i: 0.1 * ncut(o, i)))
print("cfg5* B=8  512^2 CE + 0.1 NCut + 0.1 Bnd : %.2f ms/step  %.1f img/s  loss %.4f" %
      run(8, 512, lambda o, i: 0.1 * ncut(o, i) + 0.1 * bnd(ops.softmax_channels(o), i).mean()))

model = build_segmentation_model().to(dev).eval()
N = 64
img, masks = bench.synthetic_batch(N, 256, 256, dev, 3)
masks255 = masks * 255
refine_pseudo_masks_batched(model, img[:8], masks255[:8], num_steps=2)
refine_pseudo_masks_batched(model, img, masks255, threshold=0.3, lr=1e-4, num_steps=10)     # allocator warm-up at N
torch.cuda.synchronize()
t = time.perf_counter()
out = refine_pseudo_masks_batched(model, img, masks255, threshold=0.3, lr=1e-4, num_steps=10)
torch.cuda.synchronize()
dt = time.perf_counter() - t
print("refine_pseudo_masks_batched N=64 256^2, 10 steps (+1 DeepLab eval forward): %.1f ms total, %.2f ms/img" % (dt * 1e3, dt * 1e3 / N))
